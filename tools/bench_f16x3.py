"""Correctness + micro-benchmark of the split-f16 GEMM per tile configuration (developer tool)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: F401,E402
import torch

from clavaddpm_amd import _lib, ops


def timeit(fn, n=20, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    L = _lib.lib()
    ncfg = L.clava_gemm_f16x3_config_count_()
    g = torch.Generator(device="cuda").manual_seed(0)
    # ---- correctness on ragged shapes, every configuration
    for (M, N, K) in [(333, 70, 45), (1000, 300, 96), (513, 1024, 128), (700, 50, 128), (321, 2, 512)]:
        x = torch.randn(M, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) * 0.1
        b = torch.randn(N, device="cuda", generator=g)
        ref = (x.double() @ w.double().T + b.double()).clamp_min(0)
        ovf = ops.overflow_flag("cuda")
        xs, wsp = ops.split_pack(x, 3, ovf), ops.split_pack(w, 12, ovf)
        rt = float((ops.split_unpack(xs) - x).abs().max() / x.abs().max())
        for c in range(-1, ncfg):
            y = ops.linear_f16x3(xs, wsp, b, act="relu", out_shift=5, overflow=ovf, tile_config=c)
            yf = ops.linear_f16x3(xs, wsp, b, act="relu", out_f32=True, overflow=ovf, tile_config=c)
            e1 = float((ops.split_unpack(y).double() - ref).abs().max() / ref.abs().max())
            e2 = float((yf.double() - ref).abs().max() / ref.abs().max())
            print(f"M={M} N={N} K={K} cfg={c}: roundtrip {rt:.1e} split-out err {e1:.2e} f32-out err {e2:.2e} ovf={int(ovf.item())}")
    # fp32 kernel for comparison
    x = torch.randn(1000, 1024, device="cuda", generator=g)
    w = torch.randn(1024, 1024, device="cuda", generator=g) * 0.03
    ref = x.double() @ w.double().T
    e32 = float((ops.linear(x, w).double() - ref).abs().max() / ref.abs().max())
    e16 = float((ops.linear_f16x3(ops.split_pack(x, 0), ops.split_pack(w, 12), out_f32=True).double() - ref).abs().max() / ref.abs().max())
    print(f"K=1024 max-rel error: fp32 MFMA kernel {e32:.2e}, split-f16 {e16:.2e}")
    # ---- speed at the layer shapes of the sampling loop
    M = int(os.environ.get("M", "20000"))
    for (N, K) in [(1024, 1024), (512, 1024), (1024, 512), (512, 512), (256, 512), (512, 256), (256, 128), (128, 256), (128, 128),
                   (512, 32), (128, 32), (50, 128), (2, 512), (2, 128)]:
        x = ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 0)
        w = ops.split_pack(torch.randn(N, K, device="cuda", generator=g) * 0.05, 12)
        b = torch.randn(N, device="cuda", generator=g)
        out = ops.split_empty(M, N, 0, "cuda")
        line = []
        for c in range(ncfg):
            try:
                ms = timeit(lambda: ops.linear_f16x3(x, w, b, act="relu", tile_config=c, out=out))
            except _lib.ClavaHipError as e:
                line.append(f"cfg{c}: n/a")
                continue
            line.append(f"cfg{c}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.1f}TF")
        ms = timeit(lambda: ops.linear_f16x3(x, w, b, act="relu", tile_config=-1, out=out))
        print(f"N={N:5d} K={K:5d} | " + " | ".join(line) + f" || auto {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.1f}TF", flush=True)


if __name__ == "__main__":
    main()
