"""Hidden-block weight gradient on the split-f16 kernels: time vs number of batch slabs (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clavaddpm_amd import _lib, ops
from bench_f16x3 import timeit
g = torch.Generator(device="cuda").manual_seed(0)
M = 4096
for (N, K) in [(1024, 1024), (512, 1024), (1024, 512)]:
    dyt = ops.split_transpose(ops.split_pack(torch.randn(M, N, device="cuda", generator=g), 0))
    xt = ops.split_transpose(ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 0))
    slabs = torch.empty(8, N * K, device="cuda")
    for sp in (1, 2, 4, 8):
        line = []
        for c in (1, 5, 6, 7, -1):
            ms = timeit(lambda: ops.linear_wgrad_f16x3(dyt, xt, sp, slabs, N * K, tile_config=c))
            line.append(f"cfg{c}: {ms*1e3:6.1f}us {2*M*N*K/ms/1e9:5.0f}TF")
        print(f"N={N} K={K} splits={sp} | " + " | ".join(line), flush=True)
    dy, x = torch.randn(M, N, device="cuda", generator=g), torch.randn(M, K, device="cuda", generator=g)
    ms = timeit(lambda: ops.linear_wgrad(dy, x, splits=8, out=slabs.view(8, N, K)))
    print(f"    float32 kernel, 8 slabs: {ms*1e3:6.1f}us {2*M*N*K/ms/1e9:5.0f}TF")
