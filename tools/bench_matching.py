"""Exact unique row matching at a table-sized shape (developer tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clavaddpm_amd import matching
for (na, nb, d) in [(20000, 25000, 8), (100000, 120000, 8)]:
    rng = np.random.RandomState(0)
    A, B = rng.standard_normal((na, d)), rng.standard_normal((nb, d))
    matching.match_tables(A[:1000], B[:2000])
    t0 = time.perf_counter()
    idx, dist = matching.match_tables(A, B, unique_matching=True)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx2, _ = matching.match_tables(A, B, unique_matching=False)
    dt2 = time.perf_counter() - t0
    print(f"na={na} nb={nb} d={d}: unique {dt:.2f} s ({na/dt:.0f} rows/s), non-unique {dt2:.2f} s; mean sq dist {np.mean(dist):.4f}", flush=True)
