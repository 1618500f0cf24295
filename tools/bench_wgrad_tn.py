"""Weight-gradient product: batch-major kernel vs transposes + NT kernel (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa
import torch
from clavaddpm_amd import ops

def timeit(fn, n=50, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

M = int(os.environ.get("M", "4096"))
g = torch.Generator(device="cuda").manual_seed(0)
SHAPES = [(1024, 1024), (1024, 512), (512, 1024), (512, 128), (128, 8397)][:int(os.environ.get("NSHAPES", "5"))]
for (N, K) in SHAPES:
    dy = ops.split_pack(torch.randn(M, N, device="cuda", generator=g) * 0.01, 12)
    x = ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 4)
    S = int(os.environ.get("S", "8"))
    slabs = torch.zeros((S, N * K), device="cuda")
    dyt, xt = ops.split_transpose(dy), ops.split_transpose(x)
    t_tn = timeit(lambda: ops.linear_wgrad_f16x3_tn(dy, x, S, slabs, N * K))
    t_nt = timeit(lambda: ops.linear_wgrad_f16x3(dyt, xt, S, slabs, N * K))
    t_tr = timeit(lambda: (ops.split_transpose(dy, out=dyt), ops.split_transpose(x, out=xt)))
    fl = 2.0 * M * N * K
    print(f"N_out={N:5d} K_in={K:5d}: tn {t_tn:7.1f} us ({fl/t_tn/1e6:6.1f} TF)   nt {t_nt:7.1f} us ({fl/t_nt/1e6:6.1f} TF) + transposes {t_tr:6.1f} us", flush=True)
