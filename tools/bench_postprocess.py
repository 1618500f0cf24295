"""Post-sampling decoding timing (developer tool): quantile inverse-transform and round_columns on a 10 M x 8 table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clavaddpm_amd import postprocess as pp
from oracle import postprocess as opp
N, d, nq = 10_000_000, 8, 1000
rng = np.random.RandomState(0)
real = rng.standard_normal((30000, d)) * 10
real[:, 1] = np.round(real[:, 1] / 4)
refs = np.linspace(0, 1, nq)
quant = np.maximum.accumulate(np.nanpercentile(real, refs * 100, axis=0))
x = torch.randn(N, d, device="cuda", dtype=torch.float64)
q, r = torch.from_numpy(quant).cuda(), torch.from_numpy(refs).cuda()
out = torch.empty_like(x)
for _ in range(2): pp.quantile_inverse(x, q, r, out=out)
a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
a.record()
for _ in range(5): pp.quantile_inverse(x, q, r, out=out)
b.record()
for _ in range(5): pp.round_columns(real, out, [1])
c.record()
torch.cuda.synchronize()
t1, t2 = a.elapsed_time(b) / 5, b.elapsed_time(c) / 5
print(f"quantile_inverse {N}x{d}: {t1:.3f} ms  {16.0*N*d/t1/1e6:.0f} GB/s   round_columns(1 col, {len(np.unique(real[:,1]))} values): {t2:.3f} ms")
xs = x[:200000].cpu().numpy()
t0 = time.perf_counter(); opp.quantile_inverse(xs, quant, refs); dt = time.perf_counter() - t0
print(f"cpu (numpy/scipy restatement) 200000x{d}: {dt*1e3:.1f} ms -> GPU/CPU per row {dt/200000 / (t1*1e-3/N):.0f}x")
