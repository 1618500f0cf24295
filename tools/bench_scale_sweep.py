"""Scale-sweep configuration (BASELINE.json configs[4]): 100 numeric + 50 categorical columns
(K_f log-uniform in [2, 1000], seed 0), batch 4096.  Times the denoiser training step and an unguided reverse chain and
prints the library's per-kernel-family breakdown.  Developer tool; bench.py's `scale_sweep` leg calls `measure`."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

DL = [512, 1024, 1024, 1024, 1024, 512]


def classes(m=50, seed=0):
    rng = np.random.RandomState(seed)
    return np.array([int(v) for v in np.exp(rng.uniform(np.log(2), np.log(1000), m)).round().clip(2, 1000)])


def _families(L, n_iter):
    L.clava_prof_family_name.restype = ctypes.c_char_p
    nf = L.clava_prof_family_count()
    la, ms, fl, by = (ctypes.c_int * nf)(), (ctypes.c_double * nf)(), (ctypes.c_double * nf)(), (ctypes.c_double * nf)()
    L.clava_prof_collect(la, ms, fl, by)
    out = {}
    for i in range(nf):
        if la[i]:
            out[L.clava_prof_family_name(i).decode()] = {"launches_per_iter": round(la[i] / n_iter, 1), "avg_us": round(1e3 * ms[i] / la[i], 1),
                                                         "us_per_iter": round(1e3 * ms[i] / n_iter, 1),
                                                         "gbps": round(by[i] / ms[i] / 1e6, 1) if by[i] else 0.0,
                                                         "tflops": round(fl[i] / ms[i] / 1e9, 1) if fl[i] else 0.0}
    return out


def measure(batch=4096, n_num=100, train_steps=60, chain_steps=100, T=1000, breakdown=True, dev="cuda"):
    from clavaddpm_amd import _lib
    from clavaddpm_amd.scripts.train import Trainer
    from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
    L = _lib.lib()
    K = classes()
    sk = int(K.sum())
    torch.manual_seed(0)
    net = MLPDiffusion(n_num + sk, 0, "none", {"d_layers": list(DL), "dropout": 0.0}).to(dev).train()
    diff = GaussianMultinomialDiffusion(K, n_num, net, num_timesteps=T, device=torch.device(dev)).to(dev).train()
    rng = np.random.RandomState(1)
    data = torch.from_numpy(np.concatenate([rng.standard_normal((batch, n_num)), np.stack([rng.randint(0, k, batch) for k in K], 1)], 1)).float()
    y = torch.zeros(batch, dtype=torch.long)
    res = {"rows": batch, "n_num": n_num, "m": len(K), "sum_k": sk, "d_in": n_num + sk}
    tr = Trainer(diff, None, lr=6e-4, weight_decay=1e-5, steps=10 ** 9, device=torch.device(dev))
    data = data.to(dev)
    for _ in range(5):
        tr._run_step(data, {"y": y})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(train_steps):
        tr._run_step(data, {"y": y})
    t_enq = (time.perf_counter() - t0) / train_steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / train_steps
    flops = batch * 6.0 * (640 * (n_num + sk) + 4292608)
    res["train"] = {"steps_per_s": round(1 / dt, 1), "ms_per_step": round(dt * 1e3, 3), "tflops": round(flops / dt / 1e12, 1),
                    "host_enqueue_ms_per_step": round(t_enq * 1e3, 3)}
    if breakdown:
        L.clava_prof_enable(ctypes.c_int(8192))
        for _ in range(10):
            tr._run_step(data, {"y": y})
        torch.cuda.synchronize()
        L.clava_prof_disable()
        res["train"]["families"] = _families(L, 10)
    diff.eval()
    net.eval()
    # reverse chain: per-step GPU time from the library's per-launch events (one stream, so the family times add up);
    # a few steps only - an untrained denoiser diverges over a long chain, which no fitted one does
    diff.sample(batch, torch.ones(1), seed=3, n_steps=4)
    torch.cuda.synchronize()
    # the chain's one-off launches (initial categorical draw, embedding tables) must not be charged to a step: profile a
    # 10-step and a 5-step chain and take the difference (per-step marginal cost; a chain runs T = 1000 or 2000 steps)
    def profiled(n):
        L.clava_prof_enable(ctypes.c_int(8192))
        diff.sample(batch, torch.ones(1), seed=3, n_steps=n)
        torch.cuda.synchronize()
        L.clava_prof_disable()
        return _families(L, 1)
    n_hi, n_lo = 10, 5
    f_hi, f_lo = profiled(n_hi), profiled(n_lo)
    fam = {}
    for k, v in f_hi.items():
        lo = f_lo.get(k, {"launches_per_iter": 0.0, "us_per_iter": 0.0})
        n_l = (v["launches_per_iter"] - lo["launches_per_iter"]) / (n_hi - n_lo)
        us = (v["us_per_iter"] - lo["us_per_iter"]) / (n_hi - n_lo)
        if n_l > 0:
            fam[k] = {"launches_per_iter": round(n_l, 2), "avg_us": round(us / n_l, 1), "us_per_iter": round(us, 1),
                      "gbps": v["gbps"], "tflops": v["tflops"]}
    per_step = sum(v["us_per_iter"] for v in fam.values()) * 1e-6
    res["sample"] = {"ms_per_reverse_step": round(per_step * 1e3, 3), "rows_per_s_at_T": round(batch / (per_step * T), 1),
                     "timesteps": T, "note": f"kernel time of a {n_hi}-step minus a {n_lo}-step unguided chain, per step (one-off launches of a chain excluded), scaled to T"}
    if breakdown:
        res["sample"]["families"] = fam
    return res


if __name__ == "__main__":
    import json
    r = measure()
    fam_t, fam_s = r["train"].pop("families", {}), r["sample"].pop("families", {})
    print(json.dumps(r))
    for name, fam in (("train", fam_t), ("sample", fam_s)):
        print(name)
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["us_per_iter"]):
            print(f"    {k:28s} {v}")
