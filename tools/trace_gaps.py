#!/usr/bin/env python3
"""Per-kernel durations and the idle gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV (developer tool).

    python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n]
"""
import csv
import sys
import collections

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))[skip:]
busy = sum(e - s for s, e, _ in ev)
span = max(e for _, e, _ in ev) - ev[0][0]
# union of busy intervals (kernels of two streams overlap)
cur_s, cur_e, union = ev[0][0], ev[0][1], 0
for s, e, _ in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"kernels {len(ev)}  span {span / 1e3:.1f} us  sum of durations {busy / 1e3:.1f} us  GPU busy (union) {union / 1e3:.1f} us  idle {100 * (1 - union / span):.1f} %")
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    gaps[n1[:60]].append((s1 - e0) / 1e3)
print("gap before kernel (us, mean over launches; negative = overlap with the previous kernel):")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:60s} n={len(v):5d} mean {sum(v) / len(v):7.2f}  total {sum(v):9.1f}")
