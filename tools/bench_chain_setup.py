"""Per-call set-up cost of a fused chain (weight packing, calibration pass, time-embedding tables) vs the chain itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
diff, net, clf, cond_fn = bench.build_workload("cuda:0", 2, 50, 1000, 0)
ys = torch.randint(0, 50, (20000, 1)).cuda()
for n in (1, 1, 1, 20, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, idx = diff._initial_state(len(ys), None, 5, 0)
    keep = diff._run_chain(x, idx, cond_fn, ys, None, 5, 0, n_steps=n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n_steps={n}: {dt*1e3:.1f} ms")
