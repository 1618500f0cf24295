#!/usr/bin/env python3
"""Developer tool: time the bench workload's guided chain (20000 rows, n_num 2, 50 clusters, real layer widths) for a few
hundred reverse steps, with the library variant named by CLAVA_LIB_VARIANT (tools/_variant.py).  Prints us per reverse step.

    python3 tools/bench_chain.py [steps=300] [reps=3] [rows=20000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: F401,E402
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    dev = "cuda:0"
    torch.cuda.set_device(0)
    diff, net, clf, cond_fn = bench.build_workload(dev, 2, 50, 1000, int(os.environ.get("FIT_STEPS", "600")))
    ys = torch.randint(0, 50, (B, 1), generator=torch.Generator().manual_seed(1234)).to(dev)
    out = []
    for r in range(reps + 1):
        x, idx = diff._initial_state(B, None, 20240 + r, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keep = diff._run_chain(x, idx, cond_fn, ys, None, 20240 + r, 0, n_steps=steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r:
            out.append(1e6 * dt / steps)
    print(f"variant={os.environ.get('CLAVA_LIB_VARIANT', '-')} us/step: " + " ".join(f"{v:.1f}" for v in out) +
          f"  | min {min(out):.1f}  rows/s at T=1000: {B / (min(out) * 1e-3):.0f}  absmax {float(x.abs().max()):.3f}", flush=True)


if __name__ == "__main__":
    main()
