#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection CSV (developer tool).

    python tools/pmc_kernel.py <counter_collection.csv> <kernel name substring>
"""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} launches {len(v):5d}  mean {sum(v) / len(v):16.1f}")
