"""Micro-benchmark of the fp32 MFMA GEMM at the denoiser / classifier layer shapes (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clavaddpm_amd import ops

def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

shapes = [(20000, 1024, 1024), (20000, 512, 128), (20000, 1024, 512), (20000, 512, 1024), (20000, 128, 128),
          (20000, 256, 128), (20000, 512, 256), (4096, 1024, 1024), (20000, 15, 512), (20000, 128, 15), (20480, 1024, 1024)]
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    ms = timeit(lambda: ops.linear(x, w, b, act="relu", out=out))
    ms_t = timeit(lambda: torch.relu_(torch.addmm(b, x, w.T)))
    print(f"fwd  M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF   (torch/hipBLASLt {ms_t*1e3:8.1f} us {2*M*N*K/ms_t/1e9:7.1f} TF)")
for M, N, K in [(20000, 1024, 1024), (20000, 512, 1024), (4096, 1024, 1024)]:
    dy = torch.randn(M, N, device="cuda"); w = torch.randn(N, K, device="cuda"); a = torch.randn(M, K, device="cuda")
    out = torch.empty(M, K, device="cuda")
    ms = timeit(lambda: ops.linear_dgrad(dy, w, act_out=a, slope=0.0, out=out))
    print(f"dgrad M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF")
for M, N, K, S in [(4096, 1024, 1024, 4), (4096, 1024, 1024, 8), (4096, 512, 1024, 8)]:
    dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    out = torch.empty(S, N, K, device="cuda")
    ms = timeit(lambda: ops.linear_wgrad(dy, x, splits=S, out=out))
    print(f"wgrad M={M} N={N} K={K} S={S}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF")
