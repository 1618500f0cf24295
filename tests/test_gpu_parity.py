"""GPU parity of the HIP path against the golden vectors captured from the reference, through the C ABI.

Tolerances (north_star): category indices and cluster labels exact; numeric columns within 1e-4 relative
(float32 GEMMs differ from the CPU sgemm by summation order only, ~1e-7 per layer).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden, relerr, sub
from weights import classifier_shapes, mlp_shapes, seeded_state

from clavaddpm_amd.pipeline_utils import Classifier, make_cond_fn
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion, timestep_embedding

DEV = "cuda"


@pytest.fixture(params=["f16x3", "f32"], autouse=True)
def gemm_mode(request):
    """Every test of this file runs on both layer-kernel families of the sampling loop: split-f16 on the f16 matrix
    cores (the default) and the float32-input MFMA kernels; the tolerances are the same."""
    from clavaddpm_amd import split_nets
    before = split_nets.get_gemm_mode()
    split_nets.set_gemm_mode(request.param)
    n0 = split_nets.fallback_count
    yield request.param
    split_nets.set_gemm_mode(before)
    assert split_nets.fallback_count == n0, "the split-f16 chain fell back to float32 on a golden fixture"


def make_mlp(d_in, d_layers, state):
    net = MLPDiffusion(d_in, 0, "none", {"d_layers": list(d_layers), "dropout": 0.0})
    net.load_state_dict(state)
    return net.to(DEV).eval()


def make_diffusion(K, n_num, net, T, sched="cosine"):
    return GaussianMultinomialDiffusion(np.array(K), n_num, net, num_timesteps=T, scheduler=sched, device=torch.device(DEV)).to(DEV).eval()


def test_g2_mlp_forward():
    g = golden("g2_mlp.npz")
    temb = timestep_embedding(g["t"].to(DEV), 128).cpu()
    # the reference's float32 exp() for the frequencies is 1-ulp accurate and CPU-dependent; a 1-ulp frequency
    # difference moves cos/sin(t * f) by up to t * 2^-24 ~ 6e-5 at t ~ 1000 (see modules._freqs)
    assert float((temb - g["temb"]).abs().max()) < 1e-4
    with torch.no_grad():
        out = make_mlp(14, [64, 64], sub(g, "p:"))(g["x"].to(DEV), g["t"].to(DEV)).cpu()
        assert relerr(out, g["out"]) < 1e-5
        big = make_mlp(15, [512, 1024, 1024, 1024, 1024, 512], seeded_state(mlp_shapes(15, [512, 1024, 1024, 1024, 1024, 512]), 22))
        assert relerr(big(g["big_x"].to(DEV), g["big_t"].to(DEV)).cpu(), g["big_out"]) < 1e-5


def test_g3_gaussian_step():
    g = golden("g3_gauss_step.npz")
    d = make_diffusion([0], 6, None, 1000)
    t = g["t"].to(DEV)
    for tag in ("f64", "f32"):
        for gtag in ("plain", "guided"):
            grad = g[f"{tag}_grad"].to(DEV)
            cf = (lambda x, t_, **kw: grad) if gtag == "guided" else None
            r = d.gaussian_p_sample(g[f"{tag}_eps"].to(DEV), g[f"{tag}_x"].to(DEV), t, model_kwargs={}, cond_fn=cf,
                                    noise=g[f"{tag}_{gtag}_noise"])
            assert r["sample"].dtype == torch.float64
            assert relerr(r["sample"].cpu(), g[f"{tag}_{gtag}_sample"]) < 1e-12, (tag, gtag)
            assert relerr(r["pred_xstart"].cpu(), g[f"{tag}_{gtag}_pred_xstart"]) < 1e-12, (tag, gtag)
    xt = d.gaussian_q_sample(g["q_x0"].to(DEV), t, noise=g["q_noise"].to(DEV)).cpu()
    assert torch.equal(xt, g["q_xt"])


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_g4_multinomial_step(tag):
    g = golden("g4_multinomial_step.npz")
    K = g[f"{tag}_K"].tolist()
    d = make_diffusion(K, 0, None, 1000)
    t = g[f"{tag}_t"].to(DEV)
    new, lp = d.p_sample_index(g[f"{tag}_logits"].to(DEV), g[f"{tag}_idx_t"].to(DEV), t, g[f"{tag}_u"], return_log_prob=True)
    assert torch.equal(new.cpu().long(), g[f"{tag}_idx_new"])                    # sampled category indices: exact
    # log p(x_{t-1} | x_t): the reference normaliser is a difference of two row-wide float32 logcumsumexp values
    # (tab_ddpm/utils.py:151-166) and cancels catastrophically on wide rows (up to 4e-2 off on this 1371-class
    # fixture, platform-dependent); the kernel reduces each block directly.  So: against the golden values where the
    # cancellation is benign (10 classes), and against the oracle with an exact float64 normaliser everywhere.
    if tag == "small":
        assert float((lp.cpu() - g[f"{tag}_log_model_prob"]).abs().max()) < 5e-5
    from oracle import multinomial as mn, schedule
    tb = schedule.tables("cosine", 1000)
    exact = mn.p_pred_exact_norm(tb, g[f"{tag}_logits"], mn.index_to_log_onehot(g[f"{tag}_idx_t"], K), g[f"{tag}_t"], K)
    assert float((lp.cpu().double() - exact).abs().max()) < 2e-5
    q = d.q_sample_index(g[f"{tag}_idx_0"].to(DEV), t, g[f"{tag}_uq"])
    assert torch.equal(q.cpu().long(), g[f"{tag}_idx_q"])


def test_g5_chain_mixed():
    g = golden("g5_chain_mixed.npz")
    T, K, n_num = int(g["T"]), g["K"].tolist(), int(g["n_num"])
    net = make_mlp(n_num + sum(K), [64, 64], sub(g, "p:"))
    d = make_diffusion(K, n_num, net, T)
    x, out = d.sample(64, torch.ones(1), noise=sub(g, "nz:"))
    ref = g["sample"]
    assert x.dtype == torch.float64 and x.shape == ref.shape and not x.is_cuda
    assert torch.equal(x[:, n_num:], ref[:, n_num:])                              # category indices: exact
    assert relerr(x[:, :n_num], ref[:, :n_num]) < 1e-4
    assert torch.equal(out["y"].cpu(), g["nz:y"])


def _classifier(g, prefix, d_in, C, hidden):
    clf = Classifier(d_in=d_in, d_out=C, dim_t=128, hidden_sizes=hidden)
    clf.load_state_dict(sub(g, prefix) if prefix else seeded_state(classifier_shapes(d_in, C, hidden), 71))
    return clf.to(DEV).eval()


def test_g5_chain_conditional():
    g = golden("g5_chain_cond.npz")
    T, n_num, C = int(g["T"]), int(g["n_num"]), int(g["C"])
    net = make_mlp(n_num, [64, 64], sub(g, "p:"))
    d = make_diffusion([0], n_num, net, T)
    clf = _classifier(g, "c:", n_num, C, g["hidden"].tolist())
    ys = g["ys"]
    x, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=make_cond_fn(clf, 1.0), noise=sub(g, "nz:"))
    assert relerr(x, g["sample"]) < 1e-4
    # the same chain with the guidance supplied as an opaque Python callback (the reference's closure contract)
    fused = make_cond_fn(clf, 1.0)
    x2, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys.to(DEV)}, cond_fn=lambda x_, t_, y=None: fused(x_, t_, y=y),
                                 noise=sub(g, "nz:"))
    assert relerr(x2, g["sample"]) < 1e-4


@pytest.mark.parametrize("tag", ["small", "real"])
def test_g7_classifier_and_cond_grad(tag):
    g = golden("g7_classifier.npz")
    hidden, C = g[f"{tag}_hidden"].tolist(), int(g[f"{tag}_C"])
    x, t, y = g[f"{tag}_x"].to(DEV), g[f"{tag}_t"].to(DEV), g[f"{tag}_y"].to(DEV)
    clf = _classifier(g, "small_p:" if tag == "small" else None, x.shape[1], C, hidden)
    with torch.no_grad():
        assert relerr(clf(x.float(), t).cpu(), g[f"{tag}_logits"]) < 1e-5
    grad = make_cond_fn(clf, 1.5)(x, t, y=y)
    assert grad.dtype == torch.float32
    assert relerr(grad.cpu(), g[f"{tag}_grad"]) < 1e-4


def test_sharded_sampling_is_invariant_to_the_number_of_ranks():
    """Counter-based noise keyed by the global row: 1 rank and 3 simulated ranks produce the same table, bit for bit."""
    from clavaddpm_amd import parallel
    g = golden("g5_chain_cond.npz")
    T, n_num, C = int(g["T"]), int(g["n_num"]), int(g["C"])
    net = make_mlp(n_num, [64, 64], sub(g, "p:"))
    d = make_diffusion([0], n_num, net, T)
    clf = _classifier(g, "c:", n_num, C, g["hidden"].tolist())
    cf = make_cond_fn(clf, 1.0)
    ys = torch.randint(0, C, (301,), generator=torch.Generator().manual_seed(1))
    full, _ = parallel.conditional_sample_rows(d, ys, cf, sample_batch_size=128, seed=77, rank=0, world=1, gather=False)
    parts = [parallel.conditional_sample_rows(d, ys, cf, sample_batch_size=50, seed=77, rank=r, world=3, gather=False)[0] for r in range(3)]
    assert full.shape == (301, n_num) and torch.isfinite(full).all()
    assert torch.equal(torch.cat(parts), full)
    other, _ = parallel.conditional_sample_rows(d, ys, cf, sample_batch_size=128, seed=78, rank=0, world=1, gather=False)
    assert not torch.equal(other, full)


def test_sample_all_unconditional_mixed_properties():
    """sample_all: shapes, dtypes, category ranges, NaN policy and truncation to num_samples (reference :1072-1099)."""
    K, n_num = [3, 5, 2], 4
    g = golden("g5_chain_mixed.npz")
    net = make_mlp(n_num + sum(K), [64, 64], sub(g, "p:"))
    d = make_diffusion(K, n_num, net, 50)
    x, y = d.sample_all(150, 64, torch.ones(1), seed=5)
    assert x.shape == (150, n_num + len(K)) and x.dtype == torch.float64 and y.shape == (150,)
    cats = x[:, n_num:]
    assert torch.equal(cats, cats.round()) and bool((cats >= 0).all())
    assert all(int(cats[:, f].max()) < K[f] for f in range(len(K)))
    # every class of the 2-class feature shows up among 150 rows (the chain does not collapse)
    assert set(cats[:, 2].long().tolist()) == {0, 1}
    x2, _ = d.sample_all(150, 64, torch.ones(1), seed=5)
    assert torch.equal(x, x2)                                           # same seed, same table


@pytest.mark.parametrize("K", [[2, 8, 5], [16, 9, 3, 12], [32, 17, 2], [33, 64, 7], [256, 65, 130], [257, 1000, 4], [1025, 3, 1500],
                               [2, 3] * 550],
                         ids=["g8", "g16", "g32", "g64", "regs256", "stream", "beyond1024", "m1100_unstaged"])
def test_multinomial_kernel_every_group_width_and_block_size(K):
    """Every code path of multinomial_step_kernel (8/16/32/64-lane groups, register-resident blocks up to 256 classes,
    streamed blocks, blocks beyond 1024 classes) x (reverse with and without log p, q_sample, t = 0 rows) against the
    oracle on injected uniforms: indices exact, log p within 2e-5 of the exact-normaliser oracle."""
    from oracle import multinomial as mn, schedule
    T, B = 1000, 96
    sumK = sum(K)
    d = make_diffusion(K, 0, None, T)
    tb = schedule.tables("cosine", T)
    g = torch.Generator().manual_seed(sumK)
    logits = torch.randn(B, sumK, generator=g) * 3.0
    idx_t = torch.stack([torch.randint(0, k, (B,), generator=g) for k in K], 1)
    t = torch.randint(1, T, (B,), generator=g)
    t[:7] = 0                                                  # the decoder rows (t == 0: log_EV = log_x0 branch)
    t[7:12] = T - 1
    u = torch.rand(B, sumK, generator=g)
    ref_idx, ref_lp = mn.p_sample(tb, logits, idx_t, t, u, K)
    new, lp = d.p_sample_index(logits.to(DEV), idx_t.to(DEV), t.to(DEV), u, return_log_prob=True)
    new2, _ = d.p_sample_index(logits.to(DEV), idx_t.to(DEV), t.to(DEV), u, return_log_prob=False)
    exact = mn.p_pred_exact_norm(tb, logits, mn.index_to_log_onehot(idx_t, K), t, K)
    assert float((lp.cpu().double() - exact).abs().max()) < 2e-5
    # the arg-max without the normaliser (production path) and with it agree, and both equal the oracle's draw wherever the
    # oracle's two best scores are further apart than float32 rounding of the scores
    gum = -torch.log(-torch.log(u + 1e-30) + 1e-30)
    sc = (gum.double() + exact)
    off = mn.offsets(K)
    safe = torch.ones(B, len(K), dtype=torch.bool)
    for f in range(len(K)):
        top2 = sc[:, off[f]:off[f + 1]].topk(2, dim=1).values
        safe[:, f] = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert torch.equal(sc[:, off[f]:off[f + 1]].argmax(1)[safe[:, f]], new.cpu().long()[:, f][safe[:, f]])
    assert float(safe.double().mean()) > 0.99
    assert torch.equal(new.cpu()[safe], new2.cpu()[safe])
    # the float32 restatement of the reference: its row-wide logcumsumexp normaliser (tab_ddpm/utils.py:151-166) degenerates
    # on very wide rows (1100 blocks: -inf at t = 0, so log p = +inf and every draw is class 0); compare where it is finite
    fin = torch.stack([torch.isfinite(ref_lp[:, off[f]:off[f + 1]]).all(1) for f in range(len(K))], 1)
    assert float(fin.double().mean()) > 0.9
    assert torch.equal(new.cpu().long()[safe & fin], ref_idx[safe & fin])
    uq = torch.rand(B, sumK, generator=g)
    q = d.q_sample_index(idx_t.to(DEV), t.to(DEV), uq)
    ref_q = mn.q_sample(tb, idx_t, t, uq, K)
    assert torch.equal(q.cpu().long(), ref_q)
    # in-kernel noise: indices in range, and the same (seed, row) draws whatever the batch is cut into
    a, _ = d._categorical_step(0, logits.to(DEV), idx_t.to(DEV), t.to(DEV), None, False, seed=11, step_index=3)
    assert all(int(a[:, f].min()) >= 0 and int(a[:, f].max()) < K[f] for f in range(len(K)))


def _reference_style_cond_fn(classifier, classifier_scale):
    """The guidance closure as the reference pipeline writes it (`pipeline_utils.py:853-864`): plain torch autograd
    through `classifier(x_in, t)` - no clavaddpm_amd helper - so it only works if `Classifier.forward` is differentiable
    w.r.t. its input."""
    import torch.nn.functional as F

    def cond_fn(x, t, y=None, remove_first_col=False):
        assert y is not None
        with torch.enable_grad():
            x_in = (x[:, 1:] if remove_first_col else x).detach().requires_grad_(True).float()
            log_probs = F.log_softmax(classifier(x_in, t), dim=-1)
            selected = log_probs[range(len(log_probs)), y.view(-1)]
            return torch.autograd.grad(selected.sum(), x_in)[0] * classifier_scale
    return cond_fn


@pytest.mark.parametrize("tag", ["small", "real"])
def test_g7_reference_closure_runs_unchanged_on_the_hip_classifier(tag):
    g = golden("g7_classifier.npz")
    hidden, C = g[f"{tag}_hidden"].tolist(), int(g[f"{tag}_C"])
    x, t, y = g[f"{tag}_x"].to(DEV), g[f"{tag}_t"].to(DEV), g[f"{tag}_y"].to(DEV)
    clf = _classifier(g, "small_p:" if tag == "small" else None, x.shape[1], C, hidden)
    grad = _reference_style_cond_fn(clf, 1.5)(x, t, y=y)
    assert grad.shape == x.shape and grad.dtype == torch.float32
    assert relerr(grad.cpu(), g[f"{tag}_grad"]) < 1e-4
    fused = make_cond_fn(clf, 1.5)(x, t, y=y)
    assert relerr(grad, fused) < 1e-5
    # float64 chain state in (as `condition_mean` hands it over, :315-328): the closure's own .float() is the only cast
    grad64 = _reference_style_cond_fn(clf, 1.5)(x.double(), t, y=y)
    assert torch.equal(grad64, grad)
    with torch.no_grad():                                        # and without grad mode nothing is recorded
        assert not clf(x.float(), t).requires_grad


def test_g5_chain_with_the_reference_closure():
    g = golden("g5_chain_cond.npz")
    T, n_num, C = int(g["T"]), int(g["n_num"]), int(g["C"])
    d = make_diffusion([0], n_num, make_mlp(n_num, [64, 64], sub(g, "p:")), T)
    clf = _classifier(g, "c:", n_num, C, g["hidden"].tolist())
    ys = g["ys"]
    x, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys.to(DEV)}, cond_fn=_reference_style_cond_fn(clf, 1.0), noise=sub(g, "nz:"))
    assert relerr(x, g["sample"]) < 1e-4


@pytest.mark.parametrize("m,kf,b", [(70, 5, 258), (70, 12, 258), (70, 40, 258), (200, 3, 31), (130, 64, 17)])
def test_categorical_draws_cover_every_feature_beyond_64_features(m, kf, b):
    """Regression (round 2): with more than 64 features the kernel's feature ordering came out of LDS atomics issued by
    two wavefronts, workgroups disagreed about it, and some (row, feature) blocks were never drawn (the index buffer kept
    whatever it held).  The ordering is now a stable ballot sort.  Initial draw (:1033-1035) from injected uniforms against
    the Gumbel arg-max of `log_sample_categorical` (:506-516) in torch, every entry written, two runs identical."""
    import ctypes
    from clavaddpm_amd._lib import c_int, c_u64, check, lib, ptr, stream_ptr
    K = [kf] * m
    sumK = sum(K)
    net = MLPDiffusion(sumK, 0, "none", {"d_layers": [64, 64], "dropout": 0.0})
    d = GaussianMultinomialDiffusion(np.array(K), 0, net.to(DEV).eval(), num_timesteps=50, device=torch.device(DEV)).to(DEV).eval()
    g = torch.Generator().manual_seed(m + kf)
    u = torch.rand(b, sumK, generator=g)
    want = (-torch.log(-torch.log(u + 1e-30) + 1e-30)).view(b, m, kf).argmax(-1)
    sched = d._c_schedule()
    runs = []
    for _ in range(2):
        idx = torch.full((b, m), -7, device=DEV, dtype=torch.int32)
        ud = u.to(DEV).contiguous()
        check(lib().clava_multinomial_step_f32(ctypes.byref(sched), c_int(1), c_int(b), c_int(m), c_int(sumK), ptr(d._offsets_i32),
                                               ptr(d._log_k), None, c_int(0), None, ptr(idx), ptr(ud), c_int(sumK), None, c_int(0), None,
                                               c_int(0), None, c_int(0), c_u64(3), c_u64(0), ctypes.c_uint(0), stream_ptr()), "init")
        runs.append(idx.cpu().long())
    assert int((runs[0] == -7).sum()) == 0
    assert torch.equal(runs[0], want) and torch.equal(runs[0], runs[1])
