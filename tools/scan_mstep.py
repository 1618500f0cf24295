import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from clavaddpm_amd.clustering import gmm_estep, gmm_mstep
N, d = 1_000_000, 32
X = torch.randn(N, d, device="cuda", dtype=torch.float64) * 2
for k in (16, 32, 48, 50, 64):
    lr = torch.log_softmax(torch.randn(N, k, device="cuda", dtype=torch.float64) * 3, dim=1)
    lpn = torch.randn(N, device="cuda", dtype=torch.float64)
    for _ in range(3): gmm_mstep(X, lr, lpn)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(10): gmm_mstep(X, lr, lpn)
    b.record(); torch.cuda.synchronize()
    print(k, a.elapsed_time(b) / 10 * 1e3, "us", flush=True)
