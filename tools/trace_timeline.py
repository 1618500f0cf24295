#!/usr/bin/env python3
"""One training step's kernel timeline from a rocprofv3 --kernel-trace CSV (developer tool): every kernel between two
consecutive optimizer kernels with its start offset, duration, queue and the gap to the latest end seen so far.

    python tools/trace_timeline.py <kernel_trace.csv> [which_step_from_the_end=3] [marker=adamw_ema]
"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marker = sys.argv[3] if len(sys.argv) > 3 else "adamw_ema"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
marks = [i for i, e in enumerate(ev) if marker in e[2]]
a, b = marks[-back - 1], marks[-back]
t0 = ev[a][1]
print(f"step between optimizer kernels {a} and {b}: {b - a - 1} kernels, span {(ev[b][1] - ev[a][1]) / 1e3:.1f} us (end to end)")
last_end = t0
qs = {}
for s, e, n, q in ev[a + 1:b + 1]:
    qi = qs.setdefault(q, len(qs))
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  q{qi}  gap {(s - last_end) / 1e3:7.1f}  {short}")
    last_end = max(last_end, e)
