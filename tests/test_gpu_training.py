"""GPU parity of the training path (mixed_loss, backward, AdamW + EMA, Trainer loop) against the reference's golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden, relerr, sub

from clavaddpm_amd.scripts.train import Trainer
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion

DEV = "cuda"


@pytest.fixture(params=["f16x3", "f32"], autouse=True)
def gemm_mode(request):
    """Every test runs with the Trainer's hidden-block products on both layer-kernel families (split-f16 on the f16
    matrix cores after a float32 calibration step / float32-input MFMA throughout); same tolerances."""
    from clavaddpm_amd import split_nets
    before = split_nets.get_gemm_mode()
    split_nets.set_gemm_mode(request.param)
    yield request.param
    split_nets.set_gemm_mode(before)


def _setup(K, n_num, state, T=1000):
    d_in = n_num + (sum(K) if K[0] else 0)
    net = MLPDiffusion(d_in, 0, "none", {"d_layers": [64, 64], "dropout": 0.0})
    net.load_state_dict(state)
    net = net.to(DEV).train()
    d = GaussianMultinomialDiffusion(np.array(K), n_num, net, num_timesteps=T, device=torch.device(DEV)).to(DEV).train()
    return net, d


def _close_trajectory(a, b, name):
    """Weights after a few Adam steps: the bulk agrees to 1e-4 of the tensor's scale; the few elements whose gradient is
    rounding noise (see test_g8_trainer_trajectory) may differ by up to the accumulated step size."""
    d = (a.double() - b.double()).abs().flatten() / b.abs().max().double()
    assert float(torch.quantile(d, 0.99)) < 1e-4, (name, float(torch.quantile(d, 0.99)))
    assert float(d.max()) < 5e-3, (name, float(d.max()))


@pytest.mark.parametrize("tag", ["mixed", "gauss"])
def test_g6_mixed_loss_and_grads(tag):
    g = golden("g6_loss.npz")
    K, n_num = g[f"{tag}_K"].tolist(), int(g[f"{tag}_n_num"])
    net, d = _setup(K, n_num, sub(g, f"{tag}_p:"))
    draws = {"t": g[f"{tag}_t"], "noise": g[f"{tag}_noise"], "u": g.get(f"{tag}_u")}
    x = g[f"{tag}_x"].to(DEV)
    lm, lg = d.mixed_loss(x, {"y": torch.zeros(len(x), dtype=torch.long, device=DEV)}, draws=draws)
    assert lm.dim() == 0 and lg.dim() == 0
    # losses within 1e-4 relative (north_star); the multinomial term carries the reference's normaliser noise (see
    # test_g4_multinomial_step) which is ~1e-6 relative on this 10-class fixture
    assert abs(float(lg.detach()) - float(g[f"{tag}_loss_gauss"])) <= 1e-4 * abs(float(g[f"{tag}_loss_gauss"]))
    if K[0]:
        assert abs(float(lm.detach()) - float(g[f"{tag}_loss_multi"])) <= 1e-4 * abs(float(g[f"{tag}_loss_multi"]))
    else:
        assert float(lm.detach()) == 0.0
    (lm + lg).backward()
    # Gradients.  The reference normalises the categorical posteriors by subtracting two row-wide float32 logcumsumexp
    # values (tab_ddpm/utils.py:151-166); on the CPU, swapping only that normaliser for an exact one moves these very
    # gradients by 2.2e-4 relative, so the golden values carry that much noise of their own.  Hence: tight against the
    # oracle with the exact normaliser (the arithmetic the kernel implements), loose against the golden values.
    from oracle import loss as oloss, schedule
    tb = schedule.tables("cosine", 1000)
    p_ref = {k: v.clone().requires_grad_(True) for k, v in sub(g, f"{tag}_p:").items()}
    rm, rg = oloss.mixed_loss(tb, 1000, p_ref, g[f"{tag}_x"], n_num, K, g[f"{tag}_t"], g[f"{tag}_noise"], g.get(f"{tag}_u"),
                              exact_norm=True)
    (rm + rg).backward()
    for name, p in net.named_parameters():
        assert relerr(p.grad.cpu(), p_ref[name].grad) < 2e-5, name
        assert relerr(p.grad.cpu(), g[f"{tag}_g:{name}"]) < (5e-4 if K[0] else 1e-4), name


def test_g8_trainer_trajectory():
    g = golden("g8_trainer.npz")
    T, K, n_num, steps = int(g["T"]), g["K"].tolist(), int(g["n_num"]), int(g["steps"])
    net, d = _setup(K, n_num, sub(g, "p0:"), T)
    tr = Trainer(d, iter(()), lr=float(g["lr"]), weight_decay=float(g["wd"]), steps=steps, device=torch.device(DEV))
    lh = g["loss_history"].numpy()
    for i in range(steps):
        draws = {"t": g["t"][i], "noise": g["noise"][i], "u": g["u"][i]}
        lm, lg = tr._run_step(g["x"][i], {"y": torch.zeros(32, dtype=torch.long)}, draws=draws)
        tr._anneal_lr(i)
        assert abs(np.around(float(lm), 4) - lh[i, 1]) <= 1e-4 * max(1.0, abs(lh[i, 1])) + 1.01e-4
        assert abs(np.around(float(lg), 4) - lh[i, 2]) <= 1.01e-4
    # Weights after 5 AdamW steps.  Adam divides each gradient element by its own running magnitude, so an element whose
    # gradient is ~1e-6 of the tensor's largest gets a full lr-sized update whose SIGN-level content is rounding noise
    # (1e-7-of-max differences between any two float32 implementations, the reference on CPU vs on GPU included).  The
    # trajectory is therefore only comparable to ~lr * steps / |w| ~ 1e-3; the optimizer arithmetic itself is pinned to
    # 1e-6 on identical gradients in test_adamw_ema_kernel_matches_torch_adamw.
    from oracle import schedule, trainer as otr
    tb = schedule.tables("cosine", T)
    p_ref = {k: v.clone() for k, v in sub(g, "p0:").items()}
    dr = [{"t": g["t"][i], "noise": g["noise"][i], "u": g["u"][i]} for i in range(steps)]
    _, ema_ref = otr.run_loop(tb, T, p_ref, list(g["x"]), dr, n_num, K, float(g["lr"]), float(g["wd"]), steps, exact_norm=True)
    for name, p in net.named_parameters():
        _close_trajectory(p.detach().cpu(), p_ref[name].detach(), name)
        _close_trajectory(p.detach().cpu(), g["p1:" + name], name)
    for name, p in tr.ema_model.named_parameters():
        assert relerr(p.detach().cpu(), ema_ref[name]) < 1e-5, name
        assert relerr(p.detach().cpu(), g["ema:" + name]) < 1e-5, name
    tr.finalize()
    import pickle
    blob = pickle.dumps(d)                                   # the pipeline pickles the whole diffusion object
    assert len(blob) < 4 * sum(p.numel() for p in net.parameters()) * 2
    d2 = pickle.loads(blob)
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), d2._denoise_fn.named_parameters()):
        assert n1 == n2 and torch.equal(p1.cpu(), p2.cpu())


def test_adamw_ema_kernel_matches_torch_adamw():
    """clava_adamw_ema_f32 (slab sum + AdamW + EMA) against torch.optim.AdamW / update_ema on identical gradients."""
    import ctypes
    from clavaddpm_amd._lib import c_float, c_int, check, lib, ptr, stream_ptr
    torch.manual_seed(3)
    n, S = 10007, 4
    w0 = torch.randn(n)
    w_ref = w0.clone().requires_grad_(True)
    ema_ref = w0.clone()
    opt = torch.optim.AdamW([w_ref], lr=6e-4, weight_decay=1e-5)
    w, ema = w0.clone().to(DEV), w0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 7):
        slabs = torch.randn(S, n) * (10.0 ** torch.randint(-6, 1, (n,)).float())
        lr = 6e-4 * (1 - (step - 1) / 10)
        for pg in opt.param_groups:
            pg["lr"] = lr
        w_ref.grad = slabs.sum(0) * 0.5
        opt.step()
        ema_ref.mul_(0.999).add_(w_ref.detach(), alpha=1 - 0.999)
        sl = slabs.to(DEV).contiguous()
        check(lib().clava_adamw_ema_f32(ptr(w), ptr(ema), ptr(m), ptr(v), ptr(sl), c_int(S), ctypes.c_longlong(n), c_float(0.5),
                                        ctypes.c_longlong(n), ctypes.c_double(lr), ctypes.c_double(0.9), ctypes.c_double(0.999),
                                        ctypes.c_double(1e-8), ctypes.c_double(1e-5), ctypes.c_longlong(step), ctypes.c_double(0.999),
                                        stream_ptr()), "clava_adamw_ema_f32")
    assert relerr(w.cpu(), w_ref.detach()) < 2e-6
    assert relerr(ema.cpu(), ema_ref) < 1e-6


def test_reference_loop_with_torch_adamw_matches_fused_trainer():
    """The reference's own loop body (zero_grad, mixed_loss, backward, torch AdamW, update_ema) on the compat path gives
    the same trajectory as the fused Trainer."""
    from clavaddpm_amd.scripts.utils_train import update_ema
    from copy import deepcopy
    g = golden("g8_trainer.npz")
    T, K, n_num, steps = int(g["T"]), g["K"].tolist(), int(g["n_num"]), int(g["steps"])
    net, d = _setup(K, n_num, sub(g, "p0:"), T)
    ema = deepcopy(net)
    opt = torch.optim.AdamW(d.parameters(), lr=float(g["lr"]), weight_decay=float(g["wd"]))
    for i in range(steps):
        draws = {"t": g["t"][i], "noise": g["noise"][i], "u": g["u"][i]}
        opt.zero_grad()
        lm, lg = d.mixed_loss(g["x"][i].to(DEV), {"y": torch.zeros(32, dtype=torch.long, device=DEV)}, draws=draws)
        (lm + lg).backward()
        opt.step()
        for pg in opt.param_groups:
            pg["lr"] = float(g["lr"]) * (1 - i / steps)
        update_ema(ema.parameters(), net.parameters())
    for name, p in net.named_parameters():
        _close_trajectory(p.detach().cpu(), g["p1:" + name], name)         # see test_g8_trainer_trajectory
    for name, p in ema.named_parameters():
        assert relerr(p.detach().cpu(), g["ema:" + name]) < 1e-5, name


def test_training_reduces_loss_on_a_learnable_target():
    """Size-independent property: a few hundred fused steps on data with structure lower the Gaussian loss."""
    torch.manual_seed(0)
    n_num = 6
    net = MLPDiffusion(n_num, 0, "none", {"d_layers": [128, 128], "dropout": 0.0}).to(DEV).train()
    d = GaussianMultinomialDiffusion(np.array([0]), n_num, net, num_timesteps=100, device=torch.device(DEV)).to(DEV).train()
    base = torch.randn(4096, 2)
    data = torch.cat([base, base * 0.5, base[:, :1] - base[:, 1:], base[:, :1] + base[:, 1:]], dim=1)

    def it():
        while True:
            yield data[torch.randint(0, len(data), (1024,))], torch.zeros(1024, dtype=torch.long)

    tr = Trainer(d, it(), lr=2e-3, weight_decay=0.0, steps=300, device=torch.device(DEV))
    tr.log_every = 50
    tr.run_loop()
    hist = tr.loss_history["gloss"].to_numpy()
    assert hist[-1] < 0.6 * hist[0], hist


def test_split_overflow_reruns_the_window_on_the_float32_kernels():
    """A raised split-f16 overflow flag no longer aborts training: `run_loop` restores the snapshot of
    the window's start and re-runs the window's batches on the float32 kernels.  The flag is raised by hand in the middle
    of the second window; the run must finish with the same number of logged rows, count one recovery, and end within
    float32-vs-split rounding (amplified by AdamW: see test_g8) of a run that never overflowed - both runs see the same
    batches and the same draws, because the snapshot carries the RNG states."""
    from clavaddpm_amd import split_nets

    def run(poke_at):
        torch.manual_seed(0)
        n_num = 6
        net = MLPDiffusion(n_num, 0, "none", {"d_layers": [128, 256, 128], "dropout": 0.0}).to(DEV).train()
        d = GaussianMultinomialDiffusion(np.array([0]), n_num, net, num_timesteps=100, device=torch.device(DEV)).to(DEV).train()
        g = torch.Generator().manual_seed(5)
        data = torch.randn(40, 512, n_num, generator=g)
        state = {"i": 0}

        def it():
            while True:
                i = state["i"]
                state["i"] += 1
                if i == poke_at:
                    tr._split.flag.fill_(1)            # as if a kernel of this window had seen a value beyond 65504
                yield data[i % 40], torch.zeros(512, dtype=torch.long)

        tr = Trainer(d, it(), lr=1e-3, weight_decay=0.0, steps=30, device=torch.device(DEV))
        tr.split_period = 8
        tr.log_every = 5
        tr.run_loop()
        return tr, torch.cat([p.detach().flatten().cpu() for p in net.parameters()]), state["i"]

    prev = split_nets.get_gemm_mode()
    split_nets.set_gemm_mode("f16x3")
    try:
        tr_a, w_a, drawn_a = run(poke_at=13)           # window 2 = steps 9..17 (step 0 and 9 calibrate): detected at step 18
        tr_b, w_b, drawn_b = run(poke_at=-1)
    finally:
        split_nets.set_gemm_mode(prev)
    assert tr_a.overflow_recoveries == 1 and tr_b.overflow_recoveries == 0
    assert drawn_a == drawn_b == 30                     # re-run batches come from the snapshot, not from the iterator
    assert list(tr_a.loss_history["step"]) == list(tr_b.loss_history["step"]) == [5, 10, 15, 20, 25, 30]
    assert np.allclose(tr_a.loss_history["gloss"].to_numpy(), tr_b.loss_history["gloss"].to_numpy(), atol=2e-3)
    assert torch.isfinite(w_a).all()
    assert float((w_a - w_b).abs().max()) < 5e-3 * float(w_b.abs().max())


def test_mixed_table_step_replays_as_a_graph_with_fresh_categorical_draws():
    """tables with categorical columns take the captured-step path too.  The categorical generator's
    key reads a device step word (clava_multinomial_step_counter_f32), so two replays on the SAME batch noise the
    categories differently, the word advances once per replay, and training still lowers both losses."""
    from clavaddpm_amd import split_nets
    prev = split_nets.get_gemm_mode()
    split_nets.set_gemm_mode("f16x3")
    try:
        torch.manual_seed(0)
        K, n_num, B = [4, 7, 3], 3, 1024
        net = MLPDiffusion(n_num + sum(K), 0, "none", {"d_layers": [128, 256, 128], "dropout": 0.0}).to(DEV).train()
        d = GaussianMultinomialDiffusion(np.array(K), n_num, net, num_timesteps=100, device=torch.device(DEV)).to(DEV).train()
        g = torch.Generator().manual_seed(1)
        z = torch.randn(8192, 2, generator=g)
        cats = torch.stack([(z[:, 0] > 0).long() * 3, (z[:, 1] * 2).clamp(-3, 3).round().long() + 3, (z[:, 0] + z[:, 1] > 0).long() * 2], 1)
        table = torch.cat([z, z[:, :1] - z[:, 1:], cats.float()], 1)

        def it():
            while True:
                yield table[torch.randint(0, len(table), (B,), generator=g)], torch.zeros(B, dtype=torch.long)

        tr = Trainer(d, it(), lr=2e-3, weight_decay=0.0, steps=240, device=torch.device(DEV))
        tr.log_every = 40
        tr.run_loop()
        gr = tr._graph
        assert gr is not None and int(tr._step_word.item()) > 200        # nearly every step was a replay
        hist = tr.loss_history
        # (240 steps from scratch: the first logged means sit near 1.1 / 1.3, the last near 0.88 / 0.83, whatever the draws)
        assert hist["mloss"].iloc[-1] < 0.9 * hist["mloss"].iloc[0] and hist["gloss"].iloc[-1] < 0.9 * hist["gloss"].iloc[0], hist
        # two replays on one batch: the log-one-hot input columns (noised categories) must differ in a fair share of rows
        x = table[:B]
        before = int(tr._step_word.item())
        tr._flatten()
        tr._run_step(x, {"y": torch.zeros(B, dtype=torch.long)})
        a = tr._graph["keep"][2][:, n_num:].clone()
        tr._run_step(x, {"y": torch.zeros(B, dtype=torch.long)})
        b = tr._graph["keep"][2][:, n_num:].clone()
        assert int(tr._step_word.item()) == before + 2
        changed = float((a != b).any(dim=1).float().mean())
        assert changed > 0.2, changed
    finally:
        split_nets.set_gemm_mode(prev)


def test_device_draws_of_the_captured_step():
    """clava_train_draw_qsample_f32: t uniform on 0..T-1, noise standard normal, x_t = gaussian_q_sample(x0, t, noise)
    (:172-180) exactly from the stored draws; the draws are a pure function of (seed, step word): the same word repeats
    them, the next word gives fresh ones.  Distribution checks at 4-sigma bars for 2^18 rows."""
    from clavaddpm_amd import training as tr
    torch.manual_seed(0)
    B, n, T = 1 << 18, 3, 1000
    net = MLPDiffusion(n, 0, "none", {"d_layers": [64, 64], "dropout": 0.0}).to(DEV)
    d = GaussianMultinomialDiffusion(np.array([0]), n, net, num_timesteps=T, device=torch.device(DEV)).to(DEV)
    x = torch.randn(B, n + 2, device=DEV)[:, :n]                # row stride 5: a view, as a batch with trailing columns
    word = torch.zeros(1, device=DEV, dtype=torch.int32)
    t0, z0, xin0, _, _ = tr.draw_noised_input(d, x, 1234, word)
    t0b, z0b, xin0b, _, _ = tr.draw_noised_input(d, x, 1234, word)
    word.add_(1)
    t1, z1, xin1, _, _ = tr.draw_noised_input(d, x, 1234, word)
    t2, z2, _, _, _ = tr.draw_noised_input(d, x, 99, word)
    assert torch.equal(t0, t0b) and torch.equal(z0, z0b) and torch.equal(xin0, xin0b)
    for ta, za in ((t1, z1), (t2, z2)):
        assert float((ta != t0).float().mean()) > 0.99 and float((za != z0).float().mean()) > 0.99
    assert int(t0.min()) == 0 and int(t0.max()) == T - 1
    tf = t0.double()
    assert abs(float(tf.mean()) - (T - 1) / 2) < 4 * (T / 12 ** 0.5) / B ** 0.5
    counts = torch.bincount(t0, minlength=T).double()
    assert float(((counts - B / T) ** 2 / (B / T)).sum()) < T + 4 * (2 * T) ** 0.5      # chi-square, 999 degrees of freedom
    zz = z0.double().flatten()
    N = zz.numel()
    assert abs(float(zz.mean())) < 4 / N ** 0.5 and abs(float(zz.var()) - 1) < 4 * (2 / N) ** 0.5
    assert abs(float((zz ** 3).mean())) < 4 * (15 / N) ** 0.5 and abs(float((zz ** 4).mean()) - 3) < 4 * (96 / N) ** 0.5
    assert abs(float((z0[:, 0] * z0[:, 1]).double().mean())) < 4 / B ** 0.5             # columns independent
    assert abs(float((z0[:, 0].double() * (tf - tf.mean())).mean())) < 4 * (T / 12 ** 0.5) / B ** 0.5
    want = d.sqrt_alphas_cumprod.float()[t0][:, None] * x + d.sqrt_one_minus_alphas_cumprod.float()[t0][:, None] * z0
    assert float((xin0 - want).abs().max()) <= 1e-6 * float(want.abs().max())


@pytest.mark.parametrize("form", ["closed", "gumbel"])
def test_categorical_q_sample_of_the_training_draws_has_the_reference_distribution(form, monkeypatch):
    """clava_train_draw_mixed_f32 draws x_t ~ q(x_t | x_0) of the categorical columns in closed form (clean class with
    probability e^lh / (e^lh + (K-1) e^lc), else uniform over the others); the reference perturbs q_pred's K
    log-probabilities with Gumbel noise and takes the arg-max (:403-412, :494-516).  Both forms must show the exact
    probabilities cumprod_alpha_t + (1 - cumprod_alpha_t) / K (4.5-sigma bars, 2^18 rows, every t of an 8-step schedule
    and every feature), uniform targets among the other classes, the clean classes as idx0, a log-one-hot image that is
    exactly index_to_log_onehot(idxt) (utils.py:127-134), and draws that are a pure function of (seed, step word)."""
    from clavaddpm_amd import training as tr
    monkeypatch.setattr(tr, "QSAMPLE_CLOSED", form == "closed")
    torch.manual_seed(0)
    K, n, T, B = [2, 5, 13, 40, 300], 1, 8, 1 << 18
    net = MLPDiffusion(n + sum(K), 0, "none", {"d_layers": [32, 32], "dropout": 0.0}).to(DEV)
    d = GaussianMultinomialDiffusion(np.array(K), n, net, num_timesteps=T, device=torch.device(DEV)).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.cat([torch.randn(B, n, device=DEV, generator=g)] +
                  [torch.randint(0, k, (B, 1), device=DEV, generator=g).float() for k in K], 1)
    word = torch.zeros(1, device=DEV, dtype=torch.int32)
    t, _, x_in, idx0, idxt = tr.draw_noised_input(d, x, 77, word, dense=True)
    t_b, _, x_in_b, idx0_b, idxt_b = tr.draw_noised_input(d, x, 77, word, dense=False)
    assert torch.equal(idxt, idxt_b) and torch.equal(t, t_b) and x_in_b.shape[1] == n
    assert torch.equal(idx0.long(), x[:, n:].long())
    off = np.concatenate([[0], np.cumsum(K)])
    image = torch.full((B, sum(K)), float(np.float32(np.log(1e-30))), device=DEV)
    image.scatter_(1, idxt.long() + torch.tensor(off[:-1], device=DEV)[None, :], 0.0)
    assert torch.equal(x_in[:, n:], image)
    word.add_(1)
    _, _, _, _, idxt_next = tr.draw_noised_input(d, x, 77, word, dense=False)
    assert float((idxt_next != idxt).any(dim=1).float().mean()) > 0.5
    ca = d.log_cumprod_alpha.double().exp().cpu().numpy()
    stay = (idxt == idx0).cpu().numpy()
    tt = t.cpu().numpy()
    for f, k in enumerate(K):
        for tv in range(T):
            rows = tt == tv
            nrow = int(rows.sum())
            p = ca[tv] + (1 - ca[tv]) / k
            assert abs(stay[rows, f].mean() - p) < 4.5 * (p * (1 - p) / nrow) ** 0.5 + 1e-6, (form, k, tv)
        if k > 2:
            moved = ~stay[:, f]
            shift = ((idxt[:, f] - idx0[:, f]) % k).cpu().numpy()[moved]          # 1 .. k-1, uniform
            counts = np.bincount(shift, minlength=k)[1:].astype(np.float64)
            e = moved.sum() / (k - 1)
            assert ((counts - e) ** 2 / e).sum() < (k - 2) + 4.5 * (2 * (k - 2)) ** 0.5, (form, k)


def test_closed_form_q_sample_edge_cases():
    """A one-class feature can only stay where it is; a clean class outside its feature (corrupt input) has no clean class
    to favour and draws uniformly inside the feature -- never an index outside it (the gather-form first layer and the
    class-sum weight gradient index with it)."""
    from clavaddpm_amd import training as tr
    torch.manual_seed(0)
    K, B = [1, 6], 1 << 14
    net = MLPDiffusion(sum(K), 0, "none", {"d_layers": [32, 32], "dropout": 0.0}).to(DEV)
    d = GaussianMultinomialDiffusion(np.array(K), 0, net, num_timesteps=10, device=torch.device(DEV)).to(DEV)
    x = torch.zeros(B, 2, device=DEV)
    x[:, 1] = 9.0                                           # outside 0..5
    word = torch.zeros(1, device=DEV, dtype=torch.int32)
    t, noise, x_in, idx0, idxt = tr.draw_noised_input(d, x, 3, word, dense=False)
    assert noise is None and int(idxt[:, 0].abs().max()) == 0
    assert int(idxt[:, 1].min()) == 0 and int(idxt[:, 1].max()) == 5
    counts = torch.bincount(idxt[:, 1].long(), minlength=6).double()
    assert float(((counts - B / 6) ** 2 / (B / 6)).sum()) < 5 + 4.5 * 10 ** 0.5


def test_numeric_table_graph_replays_draw_fresh_timesteps_and_noise():
    """The captured step of a numeric-only table: every replay advances the device step word and so draws new t and noise."""
    from clavaddpm_amd import split_nets
    prev = split_nets.get_gemm_mode()
    split_nets.set_gemm_mode("f16x3")
    try:
        torch.manual_seed(0)
        B = 512
        net = MLPDiffusion(2, 0, "none", {"d_layers": [128, 256, 128], "dropout": 0.0}).to(DEV).train()
        d = GaussianMultinomialDiffusion(np.array([0]), 2, net, num_timesteps=1000, device=torch.device(DEV)).to(DEV).train()
        tr = Trainer(d, None, lr=1e-3, weight_decay=0.0, steps=10 ** 6, device=torch.device(DEV))
        x = torch.randn(B, 2)
        y = {"y": torch.zeros(B, dtype=torch.long)}
        for _ in range(4):
            tr._run_step(x, y)
        assert tr._graph is not None
        before = int(tr._step_word.item())
        tr._run_step(x, y)
        ta, za = tr._graph["keep"][3].clone(), tr._graph["keep"][4].clone()
        tr._run_step(x, y)
        tb, zb = tr._graph["keep"][3].clone(), tr._graph["keep"][4].clone()
        assert int(tr._step_word.item()) == before + 2
        assert float((ta != tb).float().mean()) > 0.95 and float((za != zb).float().mean()) > 0.99
    finally:
        split_nets.set_gemm_mode(prev)


def test_captured_steps_follow_the_eager_trajectory():
    """Graph replays (split-f16 step and the captured float32 calibration step, inputs through the pinned mailbox, optimizer
    inside the graph) against the same Trainer run eagerly: the draws are keyed by (seed, step word), the kernels are the
    same, so the losses of all 40 steps (a calibration every 6) agree to 1e-5 and the final weights to 2e-5."""
    import copy
    from clavaddpm_amd import split_nets
    prev = split_nets.get_gemm_mode()
    split_nets.set_gemm_mode("f16x3")
    try:
        torch.manual_seed(3)
        B = 512
        net0 = MLPDiffusion(2, 0, "none", {"d_layers": [128, 256, 128], "dropout": 0.0}).to(DEV).train()
        g = torch.Generator().manual_seed(4)
        batches = [torch.randn(B, 2, generator=g) for _ in range(40)]
        y = {"y": torch.zeros(B, dtype=torch.long)}
        finals, cal = [], []
        for use_graph in (True, False):
            net = copy.deepcopy(net0)
            d = GaussianMultinomialDiffusion(np.array([0]), 2, net, num_timesteps=1000, device=torch.device(DEV)).to(DEV).train()
            torch.manual_seed(5)                        # the Trainer takes its draw seed from torch's CPU generator
            tr = Trainer(d, None, lr=1e-3, weight_decay=1e-5, steps=10 ** 6, device=torch.device(DEV))
            tr._use_graph = use_graph
            tr.split_period = 6
            losses = []
            for x in batches:
                lm, lg = tr._run_step(x, y)
                losses.append(lg.clone())       # a replayed step returns the graph's static output tensor
            finals.append((tr.flat.clone(), tr.ema_flat.clone(), torch.stack(losses).cpu()))
            cal.append(getattr(tr, "_cal_graph", None) is not None and getattr(tr, "_graph", None) is not None)
            assert tr.opt_step == 40 and int(tr._step_word.item()) == 40 == tr._host_word
        assert cal == [True, False]                     # both captured graphs were in use in the first run, none in the second
        (wa, ea, la), (wb, eb, lb) = finals
        assert float((la - lb).abs().max()) < 1e-5 * float(lb.abs().max())
        # (not bit-equal: Adam turns rounding-level gradient differences of the two launch orders into O(lr * 1e-3) steps)
        assert float((wa - wb).abs().max()) < 2e-5 * float(wb.abs().max())
        assert float((ea - eb).abs().max()) < 2e-5 * float(eb.abs().max())
    finally:
        split_nets.set_gemm_mode(prev)


def test_alternating_batch_shapes_keep_their_captured_graphs():
    """The reference's loader ends every epoch with a short batch: each batch shape keeps its own staging buffers and
    captured graphs, so alternating shapes replay instead of re-capturing."""
    from clavaddpm_amd import split_nets
    prev = split_nets.get_gemm_mode()
    split_nets.set_gemm_mode("f16x3")
    try:
        torch.manual_seed(0)
        net = MLPDiffusion(2, 0, "none", {"d_layers": [128, 256, 128], "dropout": 0.0}).to(DEV).train()
        d = GaussianMultinomialDiffusion(np.array([0]), 2, net, num_timesteps=1000, device=torch.device(DEV)).to(DEV).train()
        tr = Trainer(d, None, lr=1e-3, weight_decay=0.0, steps=10 ** 6, device=torch.device(DEV))
        g = torch.Generator().manual_seed(1)
        full, short = torch.randn(512, 2, generator=g), torch.randn(160, 2, generator=g)
        y = lambda x: {"y": torch.zeros(len(x), dtype=torch.long)}
        for x in [full] * 4 + [short] * 3:              # calibration, warm-up and capture of both shapes
            tr._run_step(x, y(x))
        graphs = {k: sg["graph"]["graph"] for k, sg in tr._stages.items()}
        assert len(graphs) == 2 and all(v is not None for v in graphs.values())
        w0 = tr.flat.clone()
        for x in [full, full, short, full, short, short, full]:
            lm, lg = tr._run_step(x, y(x))
            assert bool(torch.isfinite(lg))
        assert {k: sg["graph"]["graph"] for k, sg in tr._stages.items()} == graphs       # replayed, not re-captured
        assert tr.opt_step == 14 and int(tr._step_word.item()) == 14 == tr._host_word
        assert bool(torch.isfinite(tr.flat).all()) and not torch.equal(tr.flat, w0)
    finally:
        split_nets.set_gemm_mode(prev)


def test_alternating_batch_shapes_on_a_wide_table_with_categorical_columns():
    """on a wide table with categorical columns the captured step packs its weights through a plan
    WITHOUT layer 0 (gather form of the first layer), while the eager warm-up step of a new batch shape packs layer 0 too.
    The captured graph holds the address of its plan's device descriptor table, so the eager step must not replace (free)
    that plan: full, full, full, short, short, short, full - the last replay reads the first plan's table again.  Both
    shapes keep their graphs, every loss stays finite and the packed operands of the replayed step match a fresh pack."""
    from clavaddpm_amd import split_nets
    prev = split_nets.get_gemm_mode()
    split_nets.set_gemm_mode("f16x3")
    try:
        torch.manual_seed(0)
        K, n_num = [100, 80, 60, 30], 40                     # D_in = 310 >= 256: the end layers run on the split kernels
        net = MLPDiffusion(n_num + sum(K), 0, "none", {"d_layers": [128, 256, 128], "dropout": 0.0}).to(DEV).train()
        d = GaussianMultinomialDiffusion(np.array(K), n_num, net, num_timesteps=100, device=torch.device(DEV)).to(DEV).train()
        tr = Trainer(d, None, lr=1e-3, weight_decay=0.0, steps=10 ** 6, device=torch.device(DEV))
        g = torch.Generator().manual_seed(1)

        def batch(B):
            cats = torch.stack([torch.randint(0, k, (B,), generator=g) for k in K], 1).float()
            return torch.cat([torch.randn(B, n_num, generator=g), cats], 1)

        y = lambda x: {"y": torch.zeros(len(x), dtype=torch.long)}
        full, short = batch(512), batch(256)
        for x in [full] * 4:                                    # calibration, warm-up, capture, replay
            tr._run_step(x, y(x))
        st = tr._split_state()
        assert st.gather_ok(d, net, 512) and st.gather_ok(d, net, 256), "the test must exercise the gather form"
        plans_full = dict(st._pack_plans)
        tables = {k: p.dev.data_ptr() for k, p in plans_full.items()}
        for x in [short] * 3 + [full] * 2 + [short, full, short, short]:
            lm, lg = tr._run_step(x, y(x))
            assert bool(torch.isfinite(lg)) and bool(torch.isfinite(lm)), (lm, lg)
        torch.cuda.synchronize()
        w_before = [w.detach().clone() for w in net.stack()[0]]     # what the next replay packs (its optimizer runs after the pack)
        lm, lg = tr._run_step(full, y(full))
        assert bool(torch.isfinite(lg)) and bool(torch.isfinite(lm)), (lm, lg)
        graphs = {k: sg["graph"]["graph"] for k, sg in tr._stages.items()}
        assert len(graphs) == 2 and all(v is not None for v in graphs.values())
        # every plan a graph may hold is still alive at its address
        for k, p in plans_full.items():
            assert st._pack_plans[k] is p and p.dev.data_ptr() == tables[k]
        assert len(st._pack_plans) >= 2                         # the dense (eager warm-up) and the gather (captured) layer sets
        torch.cuda.synchronize()
        assert bool(torch.isfinite(tr.flat).all())
        # the operands the last replay packed are the current weights' planes
        from clavaddpm_amd import ops
        for l in range(1, len(w_before) - 1):
            got = ops.split_unpack(st.bufs[("w", l)])
            assert torch.allclose(got, w_before[l], rtol=0, atol=2e-6 * float(w_before[l].abs().max())), l
    finally:
        split_nets.set_gemm_mode(prev)
