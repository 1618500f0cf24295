"""Debug: float32 MFMA GEMM family at the scale-sweep end-layer shapes (odd 8397-wide operands) vs float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clavaddpm_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
r = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
for M in (256, 4096):
    for D in (8397, 8396, 8400, 301):
        x = torch.randn(M, D, device=dev, generator=g)
        w0 = torch.randn(128, D, device=dev, generator=g) * 0.01
        wh = torch.randn(D, 512, device=dev, generator=g) * 0.05
        h = torch.randn(M, 512, device=dev, generator=g).clamp_min(0)
        dy = torch.randn(M, D, device=dev, generator=g) * 1e-3
        g0 = torch.randn(M, 128, device=dev, generator=g)
        print(M, D,
              "fwd proj %.1e" % r(ops.linear(x, w0), x.double() @ w0.double().T),
              "fwd head %.1e" % r(ops.linear(h, wh), h.double() @ wh.double().T),
              "dgrad head %.1e" % r(ops.linear_dgrad(dy, wh, act_out=h, slope=0.0), (dy.double() @ wh.double()) * (h > 0)),
              "wgrad head %.1e" % r(ops.linear_wgrad(dy, h, splits=8).double().sum(0), dy.double().T @ h.double()),
              "wgrad proj %.1e" % r(ops.linear_wgrad(g0, x, splits=8).double().sum(0), g0.double().T @ x.double()),
              "dgrad proj %.1e" % r(ops.linear_dgrad(g0, w0), g0.double() @ w0.double()))
