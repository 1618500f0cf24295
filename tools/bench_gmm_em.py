"""GMM M-step and whole-EM-iteration timing (developer tool): torch events around the C-ABI calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clavaddpm_amd.clustering import gmm_estep, gmm_mstep, unpack_mstep
for (N, d, k) in [(1_000_000, 32, 50), (1_000_000, 32, 16), (1_000_000, 64, 50), (1_000_000, 8, 50), (4_000_000, 32, 50)]:
    rng = np.random.RandomState(0)
    f = lambda a: torch.from_numpy(a).cuda()
    means, pc, w = f(rng.standard_normal((k, d))), f(rng.uniform(0.3, 3.0, (k, d))), f(rng.dirichlet(np.ones(k)))
    X = torch.randn(N, d, device="cuda", dtype=torch.float64) * 2
    lpn, resp, lab = gmm_estep(X, w, means, pc)
    for _ in range(3): gmm_mstep(X, resp, lpn)
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    a.record()
    for _ in range(10): out = gmm_mstep(X, resp, lpn)
    b.record()
    for _ in range(10):
        lpn, resp, lab = gmm_estep(X, w, means, pc)
        out = gmm_mstep(X, resp, lpn)
        st, w2, m2, cv2, pc2 = unpack_mstep(out, k, d)
    c.record()
    torch.cuda.synchronize()
    tm, te = a.elapsed_time(b) / 10, b.elapsed_time(c) / 10
    chunks = (d + 31) // 32
    print(f"N={N} d={d} k={k}: M-step {tm*1e3:.0f} us ({4.0*N*d*k/tm/1e9:.1f} TF f64 unpadded, {8.0*N*(d+k*chunks+1)/tm/1e6:.0f} GB/s)"
          f"   E+M iteration {te*1e3:.0f} us", flush=True)
