"""Multinomial loss / q-sample kernels: time per (row, feature) block against the class count (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clavaddpm_amd import training as tr
import ctypes
from clavaddpm_amd._lib import lib, check, ptr, stream_ptr
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion

def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

B, dev = 4096, "cuda"
for m, k in [(50, 2), (50, 8), (50, 32), (50, 64), (50, 128), (50, 256), (20, 1000), (0, 0)]:
    K = np.array([k] * m) if m else np.array([int(v) for v in np.exp(np.random.RandomState(0).uniform(np.log(2), np.log(1000), 50)).round().clip(2, 1000)])
    mm, sk = len(K), int(K.sum())
    net = MLPDiffusion(1 + sk, 0, "none", {"d_layers": [64, 64], "dropout": 0.0}).to(dev)
    d = GaussianMultinomialDiffusion(K, 1, net, num_timesteps=1000, device=torch.device(dev)).to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.cat([torch.randn(B, 1, device=dev, generator=g)] + [torch.randint(0, int(kk), (B, 1), device=dev, generator=g).float() for kk in K], 1)
    word = torch.zeros(1, device=dev, dtype=torch.int32)
    t, noise, x_in, idx0, idxt = tr.draw_noised_input(d, x, 1, word, dense=False)
    out = torch.randn(B, 1 + sk, device=dev, generator=g)
    t_q = timeit(lambda: tr.draw_noised_input(d, x, 1, word, dense=False))
    t_qd = timeit(lambda: tr.draw_noised_input(d, x, 1, word, dense=True))
    t_l = timeit(lambda: tr.loss_and_grad(d, out, noise, idx0, idxt, t))
    nb = B * mm
    terms = torch.empty((B, mm), device=dev); d_out = torch.empty_like(out); sched = d._c_schedule()
    def kernel_only():
        check(lib().clava_multinomial_loss_f32(ctypes.byref(sched), B, mm, sk, ptr(d._offsets_i32), ptr(d._log_k), ptr(out[:, 1:]), out.stride(0),
                                               ptr(idx0), ptr(idxt), ptr(t), ctypes.c_float(1.0 / 1000), ptr(terms), ptr(d_out[:, 1:]), d_out.stride(0),
                                               stream_ptr()), "mloss")
    t_k = timeit(kernel_only, n=50)
    print(f"m={mm:3d} K={k if m else 'sweep':>5} sumK={sk:6d}: q-sample idx-only {t_q:7.1f} us, dense {t_qd:7.1f} us; loss {t_l:7.1f} us, kernel alone {t_k:7.1f} us = {1e3*t_k/nb:6.2f} ns/block, {1e3*t_k/(B*sk):6.3f} ns/class", flush=True)
