"""Accuracy of the split-f16 GEMM as a function of the plane scale (developer tool): does a small scale (large
headroom, low plane in the float16 subnormals) cost accuracy?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clavaddpm_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 2048, 512, 1024
for spread in (0.0, 2.0, 4.0):
    x = torch.randn(M, K, device="cuda", generator=g) * torch.exp(spread * torch.randn(M, K, device="cuda", generator=g))
    x = x.clamp_min(0)     # ReLU-like
    w = torch.randn(N, K, device="cuda", generator=g) * 0.03
    ref = x.double() @ w.double().T
    mx = float(x.abs().max())
    e32 = float((ops.linear(x, w).double() - ref).abs().max() / ref.abs().max())
    import math
    for target in (14, 10, 7, 5, 3, 0, -4):
        s = target - math.ceil(math.log2(mx))
        xs = ops.split_pack(x, s)
        rt = float((ops.split_unpack(xs).double() - x.double()).abs().max() / mx)
        y = ops.linear_f16x3(xs, ops.split_pack(w, 16), out_f32=True)
        e = (y.double() - ref).abs()
        print(f"spread {spread} target 2^{target}: roundtrip abs/max {rt:.2e}  gemm max-err/max {float(e.max()/ref.abs().max()):.2e} "
              f"rms-err/rms {float(e.pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()):.2e} (fp32 kernel max-err/max {e32:.2e})")
