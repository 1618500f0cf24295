#!/usr/bin/env python3
"""Mean of every counter of one kernel over the launches of rocprofv3 --pmc passes (developer tool).

    python tools/pmc_summary.py <kernel substring> <out.json> <counter_collection.csv> [more csv ...]
"""
import collections
import csv
import json
import sys


def main():
    needle, out = sys.argv[1:3]
    acc = collections.defaultdict(list)
    dur = []
    for path in sys.argv[3:]:
        seen = set()
        for r in csv.DictReader(open(path)):
            if needle not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res = {"kernel": needle, "launches": len(dur), "avg_us_under_pmc": sum(dur) / max(len(dur), 1),
           "counters_mean_per_launch": {k: sum(v) / len(v) for k, v in sorted(acc.items())}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
