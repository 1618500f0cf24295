"""Split-f16 products at the wide end-layer shapes of the scale-sweep table (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clavaddpm_amd import _lib, ops
from bench_f16x3 import timeit
L = _lib.lib(); ncfg = L.clava_gemm_f16x3_config_count_()
g = torch.Generator(device="cuda").manual_seed(0)
M, D, H = 4096, 8397, 512
for (N, K, tag) in [(H, D, "proj fwd"), (D, H, "head fwd"), (H, D, "head dgrad")]:
    x = ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 0)
    w = ops.split_pack(torch.randn(N, K, device="cuda", generator=g) * 0.05, 12)
    out = torch.empty(M, N, device="cuda")
    line = []
    for c in list(range(ncfg)) + [-1]:
        ms = timeit(lambda: ops.linear_f16x3(x, w, None, out_f32=True, tile_config=c, out=out))
        line.append(f"cfg{c}: {ms*1e3:6.1f}us {2*M*N*K/ms/1e9:5.0f}TF")
    print(f"{tag:10s} M={M} N={N} K={K} | " + " | ".join(line), flush=True)
for (N, K, tag) in [(H, D, "proj wgrad"), (D, H, "head wgrad"), (1024, 1024, "hidden wgrad")]:
    dyt = ops.split_transpose(ops.split_pack(torch.randn(M, N, device="cuda", generator=g), 0))
    xt = ops.split_transpose(ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 0))
    slabs = torch.empty(8, N * K, device="cuda")
    line = []
    for c in list(range(ncfg)) + [-1]:
        ms = timeit(lambda: ops.linear_wgrad_f16x3(dyt, xt, 8, slabs, N * K, tile_config=c))
        line.append(f"cfg{c}: {ms*1e3:6.1f}us {2*M*N*K/ms/1e9:5.0f}TF")
    print(f"{tag:10s} M={M} N={N} K={K} | " + " | ".join(line), flush=True)
    xs = ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 0)
    buf = ops.split_empty(K, M, 0, "cuda")
    ms = timeit(lambda: ops.split_transpose(xs, out=buf))
    print(f"    transpose [{M} x {K}] {ms*1e3:6.1f}us {16.0*M*K/ms/1e6:6.0f} GB/s")
    xf = torch.randn(M, K, device="cuda", generator=g)
    ms = timeit(lambda: ops.split_pack(xf, 0, out=xs))
    print(f"    pack      [{M} x {K}] {ms*1e3:6.1f}us {8.0*M*K/ms/1e6:6.0f} GB/s")
