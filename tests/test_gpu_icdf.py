"""The inverse-CDF sampler of the reverse step's categorical draws (production path, in-kernel noise) against the EXACT
posterior p(x_{t-1} | x_t, x0_hat) of the reference (`q_posterior(predict_start(model_out), log_x_t, t)`,
tab_ddpm/gaussian_multinomial_diffsuion.py:414-455), evaluated in float64 by the oracle.  The Gumbel arg-max path (the
reference's own construction, `:506-516`) is held to the same bars in the same test: both must draw from that distribution.
Bit parity on injected uniforms stays with the Gumbel path (tests/test_gpu_parity.py)."""
import ctypes

import numpy as np
import pytest
import torch
from scipy.stats import chi2

pytestmark = pytest.mark.gpu

from clavaddpm_amd._lib import lib
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
from oracle import multinomial as omn, schedule

DEV = "cuda"


@pytest.fixture(params=["icdf", "gumbel"])
def sampler(request):
    prev = lib().clava_categorical_sampler_set(ctypes.c_int(2 if request.param == "icdf" else 0))
    yield request.param
    lib().clava_categorical_sampler_set(ctypes.c_int(prev))


def _exact_posterior(tb, logits_row, idx_t, t, K):
    """float64 probabilities per feature from the oracle's exact-normaliser p_pred on one row."""
    lo = torch.from_numpy(logits_row[None, :]).float()
    log_xt = omn.index_to_log_onehot(torch.tensor(idx_t)[None, :], K)
    lp = omn.p_pred_exact_norm(tb, lo, log_xt, torch.tensor([t]), K)[0].double().exp().numpy()
    off = omn.offsets(K)
    return [lp[off[f]:off[f + 1]] / lp[off[f]:off[f + 1]].sum() for f in range(len(K))]


@pytest.mark.parametrize("t", [0, 1, 25, 49])
def test_reverse_step_draws_follow_the_exact_posterior(sampler, t):
    """2^18 rows with the same logits / x_t per (t, feature): class frequencies within 6 sigma of the exact posterior for
    every class, chi-square over the classes with at least 8 expected draws; features of every lane-group width, every
    classes-per-lane variant of the one-block-per-wavefront form (64, 100, 200, 300, 1000 classes = 1, 2, 4, 8, 16 per lane),
    a 1500-class feature (streamed in 1024-class chunks) and peaked as well as flat logits."""
    K, T, B = [2, 5, 13, 40, 64, 100, 200, 300, 1000, 1500], 50, 1 << 18
    rng = np.random.RandomState(11 + t)
    sumK = sum(K)
    logits_row = (rng.standard_normal(sumK) * np.repeat([0.5, 2.0, 1.0, 3.0, 1.0, 2.0, 0.7, 1.5, 2.5, 2.0], K)).astype(np.float32)
    idx_t = [1, 3, 12, 0, 63, 99, 130, 150, 999, 1100]          # (hot classes in the last lane, the ragged slot, a middle slot)
    net = MLPDiffusion(sumK, 0, "none", {"d_layers": [32, 32], "dropout": 0.0}).to(DEV)
    d = GaussianMultinomialDiffusion(np.array(K), 0, net, num_timesteps=T, device=torch.device(DEV)).to(DEV)
    tb = schedule.tables("cosine", T)
    want = _exact_posterior(tb, logits_row, idx_t, t, K)
    logits = torch.from_numpy(logits_row).to(DEV)[None, :].expand(B, sumK).contiguous()
    idx = torch.tensor(idx_t, dtype=torch.int32, device=DEV)[None, :].expand(B, len(K)).contiguous()
    tt = torch.full((B,), t, dtype=torch.long, device=DEV)
    out, _ = d._categorical_step(0, logits, idx, tt, None, False, seed=1234 + t, step_index=3)
    out = out.cpu().numpy()
    for f, k in enumerate(K):
        assert out[:, f].min() >= 0 and out[:, f].max() < k
        freq = np.bincount(out[:, f], minlength=k) / B
        p = want[f]
        sigma = np.sqrt(p * (1 - p) / B)
        # ~4 200 class frequencies per case, 8 cases per sampler: at 5 sigma one of them is expected beyond the bound in
        # every tenth run of this file (seen when a change of the generator's key moved the draws); 6 sigma: one in 2 500
        worst = np.max(np.abs(freq - p) - 6.0 * sigma)
        assert worst < 2e-6, (sampler, t, k, worst)
        big = p * B >= 8
        chi = float((((freq[big] - p[big]) * B) ** 2 / (p[big] * B)).sum())
        dof = int(big.sum()) - 1
        if dof >= 2:
            # (the exact chi-square quantile: its right tail is much heavier than a normal's at a dozen degrees of freedom)
            assert chi < chi2.isf(1e-6, dof), (sampler, t, k, chi, dof)
    # one t for the whole launch (what a chain passes): the per-feature constants then come from the workgroup's LDS table
    # instead of being evaluated per block - the same function on the same arguments, so exactly the same draws
    from clavaddpm_amd._lib import c_int, check, ptr, stream_ptr
    one_t = torch.empty((B, len(K)), device=DEV, dtype=torch.int32)
    sched = d._c_schedule()
    check(lib().clava_multinomial_step_f32(ctypes.byref(sched), c_int(0), c_int(B), c_int(len(K)), c_int(sumK), ptr(d._offsets_i32),
                                           ptr(d._log_k), ptr(logits), c_int(sumK), ptr(idx), ptr(one_t), None, c_int(sumK), None, c_int(0),
                                           None, c_int(0), None, c_int(t), ctypes.c_ulonglong(1234 + t), ctypes.c_ulonglong(0),
                                           ctypes.c_uint(3), stream_ptr()), "clava_multinomial_step_f32")
    assert np.array_equal(one_t.cpu().numpy(), out)
    # a pure function of (seed, step, global row, feature): the same call repeats, another step index draws afresh
    again, _ = d._categorical_step(0, logits, idx, tt, None, False, seed=1234 + t, step_index=3)
    other, _ = d._categorical_step(0, logits, idx, tt, None, False, seed=1234 + t, step_index=4)
    assert np.array_equal(again.cpu().numpy(), out)
    assert float((other.cpu().numpy() != out)[:, 7:].mean()) > 0.002       # (a peaked posterior repeats most draws)


def test_icdf_draws_do_not_depend_on_how_rows_are_cut(sampler):
    """Row-offset keyed draws: rows [a, b) of one launch equal a launch of those rows alone with row_offset = a."""
    from clavaddpm_amd._lib import c_int, check, ptr, stream_ptr
    K, T, B = [3, 17, 70, 400], 20, 3000
    g = torch.Generator(device=DEV).manual_seed(3)
    net = MLPDiffusion(sum(K), 0, "none", {"d_layers": [32, 32], "dropout": 0.0}).to(DEV)
    d = GaussianMultinomialDiffusion(np.array(K), 0, net, num_timesteps=T, device=torch.device(DEV)).to(DEV)
    logits = torch.randn(B, sum(K), device=DEV, generator=g) * 2
    idx = torch.stack([torch.randint(0, k, (B,), device=DEV, generator=g) for k in K], 1).to(torch.int32).contiguous()
    sched = d._c_schedule()

    def run(a, b, row_offset):
        out = torch.empty((b - a, len(K)), device=DEV, dtype=torch.int32)
        lg = logits[a:b].contiguous()
        check(lib().clava_multinomial_step_f32(ctypes.byref(sched), c_int(0), c_int(b - a), c_int(len(K)), c_int(sum(K)), ptr(d._offsets_i32),
                                               ptr(d._log_k), ptr(lg), c_int(sum(K)), ptr(idx[a:b].contiguous()), ptr(out), None, c_int(sum(K)),
                                               None, c_int(0), None, c_int(0), None, c_int(7), ctypes.c_ulonglong(99),
                                               ctypes.c_ulonglong(row_offset), ctypes.c_uint(5), stream_ptr()), "clava_multinomial_step_f32")
        return out
    whole = run(0, B, 0)
    assert torch.equal(whole[1000:2200], run(1000, 2200, 1000))
    assert int(whole.min()) >= 0 and all(int(whole[:, f].max()) < k for f, k in enumerate(K))


def test_whole_chain_with_wide_categorical_features_is_invariant_to_row_cuts():
    """An unguided chain on a mixed table with 70- and 300-class features and in-kernel draws (the production path: the
    inverse-CDF sampler, four wide blocks per Philox call keyed by the global row): rows [a, b) of one call equal a call of
    those rows alone at row_offset = a, bit for bit; categories in range, reproducible for a seed."""
    K, n_num, T = [3, 70, 300, 9, 129], 2, 20
    torch.manual_seed(0)
    net = MLPDiffusion(n_num + sum(K), 0, "none", {"d_layers": [128, 128], "dropout": 0.0}).to(DEV).eval()
    d = GaussianMultinomialDiffusion(np.array(K), n_num, net, num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
    full, _ = d.sample(300, torch.ones(1), seed=21, noise=None)
    again, _ = d.sample(300, torch.ones(1), seed=21)
    assert torch.equal(full, again) and torch.isfinite(full).all()
    cats = full[:, n_num:]
    assert torch.equal(cats, cats.round()) and bool((cats >= 0).all()) and all(int(cats[:, f].max()) < K[f] for f in range(len(K)))
    assert len(set(cats[:, 2].long().tolist())) > 20                    # the 300-class feature does not collapse
    part_a, _ = d.sample(100, torch.ones(1), seed=21, row_offset=0)
    part_b, _ = d.sample(200, torch.ones(1), seed=21, row_offset=100)
    assert torch.equal(torch.cat([part_a, part_b]), full)
    other, _ = d.sample(300, torch.ones(1), seed=22)
    assert not torch.equal(other[:, n_num:], cats)
