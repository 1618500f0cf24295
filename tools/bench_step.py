"""Fused step kernels at the scale-sweep shape (100 numeric + 50 categorical columns, K_f log-uniform in
[2, 1000], seed 0): achieved HBM GB/s of clava_gauss_step_f32 and clava_multinomial_step_f32 (reverse mode, in-kernel
noise, next denoiser input written) against the algorithmic bytes of DESIGN.md section 4.  Developer tool; bench.py's
`step_kernels` leg calls `measure`."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("CLAVA_LIB_VARIANT"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _variant  # noqa: F401  (developer A/B builds)
import numpy as np
import torch


def m5_classes(m=50, lo=2, hi=1000, seed=0):
    rng = np.random.RandomState(seed)
    return np.exp(rng.uniform(np.log(lo), np.log(hi), size=m)).astype(np.int64).clip(lo, hi)


def measure(B=4096, n_num=100, K=None, T=1000, reps=20, device="cuda", only=None):
    from clavaddpm_amd._lib import c_int, c_u64, check, lib, ptr, stream_ptr
    from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion
    K = m5_classes() if K is None else np.asarray(K)
    m, sk = len(K), int(K.sum())
    D = n_num + sk
    diff = GaussianMultinomialDiffusion(K, n_num, None, num_timesteps=T, device=torch.device(device)).to(device)
    g = torch.Generator(device="cpu").manual_seed(0)
    model_out = torch.randn(B, D, generator=g).to(device)
    x = torch.randn(B, n_num, generator=g, dtype=torch.float64).to(device)
    grad = torch.randn(B, n_num, generator=g).to(device)
    idx = torch.from_numpy(np.stack([np.random.RandomState(1 + f).randint(0, K[f], B) for f in range(m)], 1).astype(np.int32)).to(device)
    idx_out = torch.empty_like(idx)
    x_in = torch.empty(B, D, device=device)
    sched = diff._c_schedule()
    L = lib()

    def gauss(step):
        check(L.clava_gauss_step_f32(ctypes.byref(sched), c_int(B), c_int(n_num), ptr(model_out), c_int(D), ptr(x), ptr(grad),
                                     c_int(n_num), None, ptr(x_in), c_int(D), None, None, c_int(T // 2), c_int(0), c_u64(7),
                                     c_u64(0), ctypes.c_uint(step), stream_ptr()), "clava_gauss_step_f32")

    def multi(step):
        check(L.clava_multinomial_step_f32(ctypes.byref(sched), c_int(0), c_int(B), c_int(m), c_int(sk), ptr(diff._offsets_i32),
                                           ptr(diff._log_k), ptr(model_out[:, n_num:]), c_int(D), ptr(idx), ptr(idx_out), None,
                                           c_int(0), ptr(x_in[:, n_num:]), c_int(D), None, c_int(0), None, c_int(T // 2), c_u64(7),
                                           c_u64(0), ctypes.c_uint(step), stream_ptr()), "clava_multinomial_step_f32")

    def multi_idx(step):      # as the chain runs it since round 2: the first layer gathers by index, no log-one-hot image is written
        check(L.clava_multinomial_step_f32(ctypes.byref(sched), c_int(0), c_int(B), c_int(m), c_int(sk), ptr(diff._offsets_i32),
                                           ptr(diff._log_k), ptr(model_out[:, n_num:]), c_int(D), ptr(idx), ptr(idx_out), None,
                                           c_int(0), None, c_int(0), None, c_int(0), None, c_int(T // 2), c_u64(7),
                                           c_u64(0), ctypes.c_uint(step), stream_ptr()), "clava_multinomial_step_f32")

    out = {"rows": B, "n_num": n_num, "m": m, "sum_k": sk}
    for name, fn, nbytes in (("gauss_step", gauss, B * n_num * 28.0), ("multinomial_step", multi, B * (8.0 * sk + 8.0 * m)),
                             ("multinomial_step_indices_only", multi_idx, B * (4.0 * sk + 8.0 * m))):
        if only is not None and name != only:
            continue
        for s in range(3):
            fn(s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for s in range(reps):
            fn(s)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        out[name] = {"us": round(us, 2), "algorithmic_bytes": nbytes, "gbps": round(nbytes / us / 1e3, 1),
                     "frac_of_hbm_peak": round(nbytes / us / 1e3 / 8000.0, 4)}
    return out


if __name__ == "__main__":
    import json
    if "--profile" in sys.argv:           # one shape only, for rocprofv3 runs
        print(json.dumps(measure(B=20000, reps=10, only=os.environ.get("STEP_ONLY"))))
        sys.exit(0)
    for B in (4096, 20000):
        print(json.dumps(measure(B=B)))
    print(json.dumps(measure(B=20000, K=[3, 5, 2, 7, 4, 10, 2, 6])))
    if "--sweep" in sys.argv:
        for k in (8, 32, 64, 256, 1024):
            r = measure(B=20000, K=[k] * (8192 // k))
            print(k, r["multinomial_step"])
