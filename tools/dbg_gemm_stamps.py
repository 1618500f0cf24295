"""Diagnostic: in-kernel clock and barrier-wait share of the split-f16 GEMM main loop (needs the `stamps` variant:
    python tools/build_variant.py stamps gemm_f16x3.hip -DCLAVA_GEMM_STAMPS
    CLAVA_LIB_VARIANT=stamps python tools/dbg_gemm_stamps.py [N K cfg M])."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: F401,E402
import numpy as np
import torch

from clavaddpm_amd import _lib, ops

N, K, cfg = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 1024, 0)))
M = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
g = torch.Generator(device="cuda").manual_seed(0)
x = ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 0)
w = ops.split_pack(torch.randn(N, K, device="cuda", generator=g) * 0.05, 12)
b = torch.randn(N, device="cuda", generator=g)
out = ops.split_empty(M, N, 0, "cuda")
import time
t0 = time.time()
while time.time() - t0 < 2.0:                      # >= 2 s of back-to-back launches: the clock the chip holds under this load
    for _ in range(200):
        ops.linear_f16x3(x, w, b, act="relu", tile_config=cfg, out=out)
    torch.cuda.synchronize()
L = _lib.lib()
n = 4096 * 8 * 8
buf = (ctypes.c_ulonglong * n)()
assert L.clava_debug_gemm_stamps_(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8, 8).astype(np.float64)
a = a[a[:, 0, 3] > 0]
cyc, bar, ticks, nkb = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
clk = np.median(cyc / ticks) * 0.1
print(f"workgroups {len(a)}  k blocks {int(nkb[0, 0])}  in-kernel clock {clk:.3f} GHz")
print(f"loop cycles per k block: median {np.median(cyc / nkb):.0f}   barrier-wait cycles per k block: median {np.median(bar / nkb):.0f} "
      f"(waves 0-3 {np.median(bar[:, :4] / nkb[:, :4]):.0f}, waves 4-7 {np.median(bar[:, 4:] / nkb[:, 4:]):.0f})")
print("per wave barrier share:", np.round(np.median(bar / cyc, axis=0), 3))
t_start, t_loop0, t_loop1, t_end = a[..., 4], a[..., 5], a[..., 6], a[..., 7]
base = t_start.min()
us = lambda v: (v - base) * 0.01
print(f"timeline of the LAST launch (us from the first wave's start; median / max over waves): kernel start {np.median(us(t_start)):.2f}/{us(t_start).max():.2f}  "
      f"loop start {np.median(us(t_loop0)):.2f}/{us(t_loop0).max():.2f}  loop end {np.median(us(t_loop1)):.2f}/{us(t_loop1).max():.2f}  "
      f"stores retired {np.median(us(t_end)):.2f}/{us(t_end).max():.2f}")
print(f"per wave: prologue {np.median(us(t_loop0) - us(t_start)):.2f} us, loop {np.median(us(t_loop1) - us(t_loop0)):.2f} us, epilogue {np.median(us(t_end) - us(t_loop1)):.2f} us")
