#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into HBM bytes per launch of one kernel.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <out.json>

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B stores.
"""
import collections
import csv
import json
import sys


def mean_by_kernel(path, counter, needle):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)


def main():
    fetch_csv, write_csv, needle, out = sys.argv[1:5]
    f, nf = mean_by_kernel(fetch_csv, "FETCH_SIZE", needle)
    w, nw = mean_by_kernel(write_csv, "WRITE_SIZE", needle)
    res = {"kernel": needle, "launches_fetch_pass": nf, "launches_write_pass": nw,
           "fetch_size_kib_raw": f, "write_size_kib_raw": w,
           "fetch_bytes_corrected": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
           "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
           "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); Infinity-Cache hits are included in FETCH_SIZE"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
