"""Category flip rate of the HIP categorical step against exact (float64) arithmetic on the same inputs.

Gumbel arg-max draws are exact functions of (weights, x_t, idx_t, uniforms); any float32-level pipeline - the reference's
included - resolves a near-tie of the top two scores by its own rounding.  This measures how often: every reverse step
starts from the float64 trajectory's state (teacher forcing, so one flip does not cascade), the HIP path runs ONE step
(`sample(..., t_start=t, n_steps=1)`) on injected uniforms, and its indices are compared with the float64 evaluation of
`p_sample` (`tab_ddpm/gaussian_multinomial_diffsuion.py:414-472, 506-516`, exact per-block normaliser).
The float64 evaluation is plain torch on `f64_device` ("cpu" in the test, the GPU in bench.py's scale_sweep leg).
Self-contained on purpose: bench.py must not import `oracle/` outside its CPU-baseline legs."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

LOG_TINY = float(torch.log(torch.tensor(1e-30, dtype=torch.float32)))


def _log_onehot(idx, K, off, dtype):
    x = torch.full((idx.shape[0], int(off[-1])), LOG_TINY, dtype=dtype, device=idx.device)
    x.scatter_(1, idx.long() + torch.as_tensor(off[:-1], device=idx.device)[None, :], 0.0)
    return x


def _lae(a, b):
    m = torch.maximum(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def _forward64(p, x_in, t, dim_t=128):
    """MLPDiffusion.forward (`tab_ddpm/modules.py:451-460`) in float64 on the float32-rounded input."""
    half = dim_t // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float64, device=x_in.device) / half).float().double()
    a = t[:, None].double() * freqs[None]
    temb = torch.cat([torch.cos(a), torch.sin(a)], dim=-1)
    e = F.linear(F.silu(F.linear(temb, p["time_embed.0.weight"], p["time_embed.0.bias"])), p["time_embed.2.weight"], p["time_embed.2.bias"])
    h = F.linear(x_in, p["proj.weight"], p["proj.bias"]) + e
    i = 0
    while f"mlp.blocks.{i}.linear.weight" in p:
        h = F.relu(F.linear(h, p[f"mlp.blocks.{i}.linear.weight"], p[f"mlp.blocks.{i}.linear.bias"]))
        i += 1
    return F.linear(h, p["mlp.head.weight"], p["mlp.head.bias"])


def _step64(diff, p, x, idx, t, noise, u, K, off, n_num):
    """One reverse step in float64: returns (x_next float64, idx_next, margin = top score - runner-up per draw)."""
    dev = x.device
    B = x.shape[0]
    tt = torch.full((B,), t, dtype=torch.long, device=dev)
    x_in = torch.cat([x.float().double(), _log_onehot(idx, K, off, torch.float64)], 1)
    out = _forward64(p, x_in, tt)
    eps, logits = out[:, :n_num], out[:, n_num:]
    g = lambda name: getattr(diff, name).to(dev).double()[t]
    # Gaussian part (`:200-237, 354-383`): fixed-large variance, eps parametrisation
    x0 = g("sqrt_recip_alphas_cumprod") * x - g("sqrt_recipm1_alphas_cumprod") * eps
    mean = g("posterior_mean_coef1") * x0 + g("posterior_mean_coef2") * x
    var = diff._model_variance.to(dev).double()[t]           # [posterior_variance[1], beta_1 .. beta_{T-1}] (`:209-211`)
    x_next = mean + (0.0 if t == 0 else 1.0) * torch.sqrt(var) * noise.to(dev).double()
    # categorical part
    la, l1a = g("log_alpha"), g("log_1_min_alpha")
    tm1 = max(t - 1, 0)
    lca = diff.log_cumprod_alpha.to(dev).double()[tm1]
    l1ca = diff.log_1_min_cumprod_alpha.to(dev).double()[tm1]
    log_xt = _log_onehot(idx, K, off, torch.float64)
    logk = torch.cat([torch.full((int(k),), math.log(int(k)), dtype=torch.float64) for k in K]).to(dev)[None, :]
    idx_next, margins = [], []
    ud = u.to(dev).double()
    gum = -torch.log(-torch.log(ud + 1e-30) + 1e-30)
    for f in range(len(K)):
        sl = slice(int(off[f]), int(off[f + 1]))
        lp0 = F.log_softmax(logits[:, sl], dim=1)
        ev = lp0 if t == 0 else _lae(lp0 + lca, l1ca - logk[:, sl])
        un = ev + _lae(log_xt[:, sl] + la, l1a - logk[:, sl])
        sc = gum[:, sl] + (un - torch.logsumexp(un, dim=1, keepdim=True))
        top = sc.topk(2, dim=1) if sc.shape[1] > 1 else None
        idx_next.append(sc.argmax(dim=1, keepdim=True))
        margins.append((top.values[:, 0] - top.values[:, 1]) if top is not None else torch.full((B,), float("inf"), device=dev, dtype=torch.float64))
    return x_next, torch.cat(idx_next, 1), torch.stack(margins, 1)


def measure(batch=4096, n_num=100, steps=4, T=1000, f64_device="cpu", modes=("f16x3", "f32"), K=None, seed=0, hidden=None):
    """{mode: {draws, flips, rate, ...}} over `steps` teacher-forced reverse steps t = T-1 .. T-steps of the scale-sweep table."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    from weights import mlp_shapes, seeded_state
    from clavaddpm_amd import split_nets
    from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
    import bench_scale_sweep
    K = bench_scale_sweep.classes() if K is None else np.asarray(K)
    hidden = list(bench_scale_sweep.DL) if hidden is None else list(hidden)
    sumK = int(K.sum())
    off = np.concatenate([[0], np.cumsum(K)]).astype(np.int64)
    state = seeded_state(mlp_shapes(n_num + sumK, hidden), 31 + seed, scale=0.5)
    net = MLPDiffusion(n_num + sumK, 0, "none", {"d_layers": list(hidden), "dropout": 0.0})
    net.load_state_dict(state)
    d = GaussianMultinomialDiffusion(K, n_num, net.to("cuda").eval(), num_timesteps=T, device=torch.device("cuda")).to("cuda").eval()
    p64 = {k: v.double().to(f64_device) for k, v in state.items()}
    g = torch.Generator().manual_seed(5 + seed)
    x = torch.randn(batch, n_num, generator=g).double().to(f64_device)
    idx = torch.stack([torch.randint(0, int(k), (batch,), generator=g) for k in K], 1).to(f64_device)
    res = {m: {"draws": 0, "flips": 0, "flips_with_margin_above_1e-4": 0, "numeric_max_rel_dev": 0.0} for m in modes}
    prev = split_nets.get_gemm_mode()
    try:
        for s in range(steps):
            t = T - 1 - s
            noise = torch.randn(batch, n_num, generator=g).double()
            u = torch.rand(batch, sumK, generator=g)
            x_next, idx_next, margin = _step64(d, p64, x, idx, t, noise, u, K, off, n_num)
            nz = {"z0": x.cpu(), "u_init": u, "noise": noise[None], "u": u[None], "y": torch.zeros(batch, dtype=torch.long)}
            for m in modes:
                split_nets.set_gemm_mode(m)
                got, _ = d.sample(batch, torch.ones(1), noise=nz, t_start=t, n_steps=1, idx_start=idx.cpu())
                gi = got[:, n_num:].long()
                diff_ = gi != idx_next.cpu()
                r = res[m]
                r["draws"] += diff_.numel()
                r["flips"] += int(diff_.sum())
                r["flips_with_margin_above_1e-4"] += int((diff_ & (margin.cpu() > 1e-4)).sum())
                ref = x_next.cpu()
                r["numeric_max_rel_dev"] = max(r["numeric_max_rel_dev"], float((got[:, :n_num] - ref).abs().max() / ref.abs().max()))
            x, idx = x_next, idx_next
    finally:
        split_nets.set_gemm_mode(prev)
    for r in res.values():
        r["rate"] = r["flips"] / max(r["draws"], 1)
    return res


if __name__ == "__main__":
    import json
    print(json.dumps(measure(f64_device=os.environ.get("F64_DEVICE", "cuda"))))
