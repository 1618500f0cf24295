"""GPU parity at the shapes of the other BASELINE.json configurations (they are parity cases, not bench lines):
  * configs[0] california household -> individual: D_in = 15 numeric columns, 25 clusters, guided;
  * configs[4] scale sweep: 100 numeric + 50 categorical columns (K_f log-uniform in [2, 1000]), batch 4096.
Each case compares the HIP path with the CPU oracle on the same injected noise at a size the oracle finishes in
seconds, and both layer-kernel families with each other at the full batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import relerr
from weights import classifier_shapes, mlp_shapes, seeded_state

from clavaddpm_amd import split_nets
from clavaddpm_amd.pipeline_utils import Classifier, make_cond_fn
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
from oracle import chain, guidance, schedule

DEV = "cuda"
DL = [512, 1024, 1024, 1024, 1024, 512]
CL = [128, 256, 512, 1024, 512, 256, 128]


def _scale_sweep_K():
    rng = np.random.RandomState(0)
    return [int(v) for v in np.exp(rng.uniform(np.log(2), np.log(1000), 50)).round().clip(2, 1000)]


def _mixed_noise(b, n_num, sumK, steps, seed):
    g = torch.Generator().manual_seed(seed)
    return {"z0": torch.randn(b, n_num, generator=g), "u_init": torch.rand(b, sumK, generator=g),
            "noise": torch.randn(steps, b, n_num, generator=g).double(), "u": torch.rand(steps, b, sumK, generator=g),
            "y": torch.zeros(b, dtype=torch.long)}


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_scale_sweep_mixed_chain_vs_oracle(mode):
    K = _scale_sweep_K()
    n_num, sumK, T, b, steps = 100, sum(K), 1000, 192, 3
    state = seeded_state(mlp_shapes(n_num + sumK, DL), 31, scale=0.5)
    net = MLPDiffusion(n_num + sumK, 0, "none", {"d_layers": list(DL), "dropout": 0.0})
    net.load_state_dict(state)
    d = GaussianMultinomialDiffusion(np.array(K), n_num, net.to(DEV).eval(), num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
    nz = _mixed_noise(b, n_num, sumK, steps, 5)
    ref = chain.sample(schedule.tables("cosine", T), T, state, n_num, K, nz, steps=steps)
    split_nets.set_gemm_mode(mode)
    n0 = split_nets.fallback_count
    try:
        x, _ = d.sample(b, torch.ones(1), noise=nz, n_steps=steps)
    finally:
        split_nets.set_gemm_mode("f16x3")
    assert split_nets.fallback_count == n0
    assert torch.equal(x[:, n_num:], ref[:, n_num:])              # 192 x 50 x 3 category draws: exact
    assert relerr(x[:, :n_num], ref[:, :n_num]) < 1e-4


def test_scale_sweep_full_batch_both_kernel_families_agree():
    K = _scale_sweep_K()
    n_num, sumK, T, b = 100, sum(K), 1000, 4096
    net = MLPDiffusion(n_num + sumK, 0, "none", {"d_layers": list(DL), "dropout": 0.0})
    net.load_state_dict(seeded_state(mlp_shapes(n_num + sumK, DL), 31, scale=0.5))
    d = GaussianMultinomialDiffusion(np.array(K), n_num, net.to(DEV).eval(), num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
    out = {}
    for mode in ("f16x3", "f32"):
        split_nets.set_gemm_mode(mode)
        out[mode], _ = d.sample(b, torch.ones(1), seed=3, n_steps=4)
    split_nets.set_gemm_mode("f16x3")
    a, r = out["f16x3"], out["f32"]
    assert torch.isfinite(r).all()
    assert relerr(a[:, :n_num], r[:, :n_num]) < 1e-4
    cats = a[:, n_num:]
    assert all(int(cats[:, f].max()) < K[f] and int(cats[:, f].min()) >= 0 for f in range(len(K)))
    # Gumbel arg-max near-ties may resolve differently between two correct float32-level products: allow 1e-4 of the draws
    assert float((a[:, n_num:] != r[:, n_num:]).double().mean()) < 1e-4


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_california_guided_chain_vs_oracle(mode):
    n_num, C, T, b, steps = 15, 25, 2000, 256, 4
    ps = seeded_state(mlp_shapes(n_num, DL), 41, scale=0.5)
    cs = seeded_state(classifier_shapes(n_num, C, CL), 42)
    net = MLPDiffusion(n_num, 0, "none", {"d_layers": list(DL), "dropout": 0.0})
    net.load_state_dict(ps)
    clf = Classifier(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=CL)
    clf.load_state_dict(cs)
    d = GaussianMultinomialDiffusion(np.array([0]), n_num, net.to(DEV).eval(), num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
    g = torch.Generator().manual_seed(9)
    ys = torch.randint(0, C, (b, 1), generator=g)
    nz = {"z0": torch.randn(b, n_num, generator=g), "noise": torch.randn(steps, b, n_num, generator=g).double()}
    tb = schedule.tables("cosine", T)
    ref = chain.conditional_sample(tb, T, ps, n_num, lambda x, t: guidance.cond_grad(cs, x, t, ys, 1.0), nz, steps=steps)
    split_nets.set_gemm_mode(mode)
    try:
        x, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=make_cond_fn(clf.to(DEV).eval(), 1.0), noise=nz, n_steps=steps)
    finally:
        split_nets.set_gemm_mode("f16x3")
    assert relerr(x, ref) < 1e-4
