"""M sweep of the main GEMM shape (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clavaddpm_amd import ops

def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

N = K = 1024
for M in [4096, 8192, 16384, 19968, 20000, 20096, 20480, 24576, 32768, 16384+4096]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    ms = timeit(lambda: ops.linear(x, w, b, act="relu", out=out))
    tiles = ((M + 127) // 128) * 8
    print(f"M={M:6d} tiles={tiles:5d} per-CU={tiles/256:5.2f}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  us/tile-round={ms*1e3/(-(-tiles//256)):6.1f}")
