"""Does a power-of-two row pitch of split operands cost bandwidth (HBM channel aliasing)?  (developer tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clavaddpm_amd import _lib, ops
from bench_f16x3 import timeit
g = torch.Generator(device="cuda").manual_seed(0)

def padded(rows, cols, pad):
    pitch = ops.split_pitch(cols) + pad
    return ops.SplitTensor(torch.zeros(rows * pitch, device="cuda", dtype=torch.uint8), rows, cols, 0, pitch)

for M in (20000, 4096):
    for (N, K) in [(1024, 1024), (512, 1024), (1024, 512)]:
        xf = torch.randn(M, K, device="cuda", generator=g)
        wf = torch.randn(N, K, device="cuda", generator=g) * 0.05
        for pad in (0, 128, 256, 512):
            x = ops.split_pack(xf, 0, out=padded(M, K, pad))
            w = ops.split_pack(wf, 12, out=padded(N, K, pad))
            out = padded(M, N, pad)
            ms = timeit(lambda: ops.linear_f16x3(x, w, None, act="relu", out=out))
            print(f"fwd M={M} N={N} K={K} pad={pad:3d}: {ms*1e3:6.1f}us {2*M*N*K/ms/1e9:5.0f}TF", flush=True)
M = 4096
for (N, K) in [(1024, 1024), (512, 1024)]:
    dys, xs = ops.split_pack(torch.randn(M, N, device="cuda", generator=g), 0), ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 0)
    slabs = torch.empty(8, N * K, device="cuda")
    for pad in (0, 128, 256, 512):
        dyt = ops.split_transpose(dys, out=padded(N, M, pad))
        xt = ops.split_transpose(xs, out=padded(K, M, pad))
        for sp in (4, 8):
            ms = timeit(lambda: ops.linear_wgrad_f16x3(dyt, xt, sp, slabs, N * K))
            print(f"wgrad N={N} K={K} pad={pad:3d} splits={sp}: {ms*1e3:6.1f}us {2*M*N*K/ms/1e9:5.0f}TF", flush=True)
