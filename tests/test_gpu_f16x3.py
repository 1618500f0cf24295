"""GPU parity of the split-f16 ("f16x3") layer kernels, through the C ABI: float32-level accuracy on the f16 matrix
cores (csrc/split.h).  Bars: the same as the float32 MFMA kernels (2e-6 of the largest output per product against a
float64 reference; chains within 1e-4 of the reference's golden vectors, categories exact - tests/test_gpu_parity.py
runs every chain on both kernel families)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden, relerr, sub

from clavaddpm_amd import _lib, ops, split_nets
from clavaddpm_amd.pipeline_utils import Classifier, make_cond_fn
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion

DEV = "cuda"


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def test_pack_unpack_round_trip():
    g = _gen(1)
    x = torch.randn(301, 77, device=DEV, generator=g) * torch.exp(2 * torch.randn(301, 77, device=DEV, generator=g))
    flag = ops.overflow_flag(DEV)
    x = x.clamp(-2000.0, 2000.0)
    for shift in (-3, 0, 4):
        s = ops.split_pack(x, shift, flag)
        assert s.pitch == 4 * 96
        back = ops.split_unpack(s)
        assert float((back - x).abs().max() / x.abs().max()) < 2.5e-7       # 22 significand bits of the largest element
    assert int(flag.item()) == 0
    ops.split_pack(x * 1e6, 6, flag)
    assert int(flag.item()) == 1                                            # out of the float16 range: flagged, not silent


@pytest.mark.parametrize("M,N,K", [(333, 70, 45), (1000, 300, 96), (513, 1024, 128), (700, 50, 128), (321, 2, 512), (5, 3, 1),
                                   (20000, 512, 1024)])
def test_linear_every_tile_configuration(M, N, K):
    g = _gen(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) * 0.1      # asymmetric operands: a transposed C map cannot pass
    b = torch.randn(N, device=DEV, generator=g)
    ref = (x.double() @ w.double().T + b.double()).clamp_min(0)
    flag = ops.overflow_flag(DEV)
    xs, wsp = ops.split_pack(x, 3, flag), ops.split_pack(w, 12, flag)
    n_cfg = _lib.lib().clava_gemm_f16x3_config_count_()
    for cfg in range(-1, n_cfg if M < 5000 else 0):
        y = ops.linear_f16x3(xs, wsp, b, act="relu", out_shift=5, overflow=flag, tile_config=cfg)
        assert relerr(ops.split_unpack(y), ref) < 2e-6, cfg
        yf = ops.linear_f16x3(xs, wsp, b, act="relu", out_f32=True, overflow=flag, tile_config=cfg)
        assert relerr(yf, ref) < 2e-6, cfg
        # pad columns of the split output are zero (they are the next layer's K padding)
        raw = y.buf.view(M, y.pitch).view(torch.float16).view(M, -1, 64)
        pad = torch.arange(32, device=DEV)[None, :] + 32 * torch.arange(raw.shape[1], device=DEV)[:, None] >= N
        assert bool((raw[:, :, :32][:, pad] == 0).all() and (raw[:, :, 32:][:, pad] == 0).all()), cfg
    assert int(flag.item()) == 0


def test_linear_identity_and_gradient_epilogue():
    n = 96
    eye = ops.split_pack(torch.eye(n, device=DEV), 0)
    w = torch.arange(n * n, device=DEV, dtype=torch.float32).reshape(n, n) / 64.0
    out = ops.linear_f16x3(eye, ops.split_pack(w, 0), out_f32=True)
    assert torch.equal(out, w.T.contiguous())            # integers / 64 below 2^22: exact in 22 bits
    # input-gradient epilogue: alpha * act'(act_out) * (dy @ W) with W given transposed
    g = _gen(9)
    M, N, K = 200, 130, 64
    dy = torch.randn(M, N, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * 0.2
    a = torch.randn(M, K, device=DEV, generator=g)
    ref = 1.5 * torch.where(a > 0, 1.0, 0.01).double() * (dy.double() @ W.double())
    got = ops.linear_f16x3(ops.split_pack(dy, 8), ops.split_pack(W.t().contiguous(), 10), act_out=ops.split_pack(a, 2), slope=0.01,
                           alpha=1.5, out_f32=True)
    assert relerr(got, ref) < 2e-6


def _cond_setup():
    g = golden("g5_chain_cond.npz")
    T, n_num, C = int(g["T"]), int(g["n_num"]), int(g["C"])
    net = MLPDiffusion(n_num, 0, "none", {"d_layers": [64, 64], "dropout": 0.0})
    net.load_state_dict(sub(g, "p:"))
    clf = Classifier(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=g["hidden"].tolist())
    clf.load_state_dict(sub(g, "c:"))
    d = GaussianMultinomialDiffusion(np.array([0]), n_num, net.to(DEV).eval(), num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
    return g, d, clf.to(DEV).eval()


def test_overflow_falls_back_to_float32_and_still_matches():
    """A chain whose state leaves the float16 range (here: the guidance scaled by 1e9) must not return garbage: the flag
    is raised, the batch is rerun on the float32 kernels, and the result equals the float32 run."""
    g, d, clf = _cond_setup()
    ys = g["ys"]
    split_nets.set_gemm_mode("f32")
    ref, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=make_cond_fn(clf, 1e9), noise=sub(g, "nz:"), n_steps=6)
    split_nets.set_gemm_mode("f16x3")
    n0 = split_nets.fallback_count
    with pytest.warns(UserWarning, match="float16 range"):
        x, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=make_cond_fn(clf, 1e9), noise=sub(g, "nz:"), n_steps=6)
    assert split_nets.fallback_count == n0 + 1
    assert torch.equal(x, ref)


def test_full_size_chain_matches_float32_kernels():
    """BASELINE configs[1] shapes (20000 rows, [512,1024x4,512] denoiser, 7-layer classifier): 8 guided steps on both
    kernel families from the same seed agree to float32 rounding."""
    from weights import classifier_shapes, mlp_shapes, seeded_state
    dl, cl, n_num, C, T = [512, 1024, 1024, 1024, 1024, 512], [128, 256, 512, 1024, 512, 256, 128], 2, 50, 1000
    net = MLPDiffusion(n_num, 0, "none", {"d_layers": list(dl), "dropout": 0.0})
    net.load_state_dict(seeded_state(mlp_shapes(n_num, dl), 3))
    clf = Classifier(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=cl)
    clf.load_state_dict(seeded_state(classifier_shapes(n_num, C, cl), 4))
    d = GaussianMultinomialDiffusion(np.array([0]), n_num, net.to(DEV).eval(), num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
    ys = torch.randint(0, C, (20000, 1), generator=torch.Generator().manual_seed(0))
    cf = make_cond_fn(clf.to(DEV).eval(), 1.0)
    out = {}
    n0 = split_nets.fallback_count
    for mode in ("f16x3", "f32"):
        split_nets.set_gemm_mode(mode)
        out[mode], _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=cf, seed=11, n_steps=8)
    split_nets.set_gemm_mode("f16x3")
    assert split_nets.fallback_count == n0
    assert torch.isfinite(out["f32"]).all()
    assert relerr(out["f16x3"], out["f32"]) < 1e-5


def test_mask_bits_give_the_same_input_gradient_as_the_activation_planes(monkeypatch):
    """The guidance chain hands the wide classifier layers' activation derivatives to the input-gradient products as bits
    (csrc/split.h `bits_out` / `mask_bits`) instead of re-reading the stored activations: the same 0/1 factors, so the
    guided samples are bit-identical to CLAVA_MASK_BITS=0 (config classifier widths; ragged row count)."""
    from weights import classifier_shapes, mlp_shapes, seeded_state
    dl, cl, n_num, C, T = [512, 1024, 512], [128, 256, 512, 1024, 512, 256, 128], 5, 50, 1000
    net = MLPDiffusion(n_num, 0, "none", {"d_layers": list(dl), "dropout": 0.0})
    net.load_state_dict(seeded_state(mlp_shapes(n_num, dl), 3))
    clf = Classifier(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=cl)
    clf.load_state_dict(seeded_state(classifier_shapes(n_num, C, cl), 4))
    d = GaussianMultinomialDiffusion(np.array([0]), n_num, net.to(DEV).eval(), num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
    ys = torch.randint(0, C, (1237, 1), generator=torch.Generator().manual_seed(0))
    cf = make_cond_fn(clf.to(DEV).eval(), 1.0)
    out = {}
    for bits in ("1", "0"):
        monkeypatch.setenv("CLAVA_MASK_BITS", bits)
        out[bits], _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=cf, seed=5, n_steps=6)
    assert torch.isfinite(out["1"]).all()
    assert torch.equal(out["1"], out["0"])


def test_gaussian_step_inside_the_guidance_s_last_launch_changes_no_bit(monkeypatch):
    """The guided chain runs the Gaussian posterior step inside the launch that produces its guidance gradient
    (narrow_split_gauss: one launch and one cross-stream hand-over less per step; CLAVA_FUSE_GAUSS=0: separate launch on the
    denoiser stream).  Same element function, same inputs: identical samples, with in-kernel and with injected noise."""
    from weights import classifier_shapes, mlp_shapes, seeded_state
    dl, cl, n_num, C, T = [512, 1024, 512], [128, 256, 512, 1024, 512, 256, 128], 5, 50, 1000
    net = MLPDiffusion(n_num, 0, "none", {"d_layers": list(dl), "dropout": 0.0})
    net.load_state_dict(seeded_state(mlp_shapes(n_num, dl), 3))
    clf = Classifier(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=cl)
    clf.load_state_dict(seeded_state(classifier_shapes(n_num, C, cl), 4))
    d = GaussianMultinomialDiffusion(np.array([0]), n_num, net.to(DEV).eval(), num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
    ys = torch.randint(0, C, (1237, 1), generator=torch.Generator().manual_seed(0))
    cf = make_cond_fn(clf.to(DEV).eval(), 1.0)
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("CLAVA_FUSE_GAUSS", fuse)
        out[fuse], _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=cf, seed=5, n_steps=7)
    assert torch.isfinite(out["1"]).all()
    assert torch.equal(out["1"], out["0"])
    assert not torch.equal(out["1"], d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=cf, seed=6, n_steps=7)[0])


@pytest.mark.parametrize("rows,cols", [(301, 77), (64, 64), (4096, 1024), (33, 130), (1, 5)])
def test_split_transpose_moves_the_planes_exactly(rows, cols):
    g = _gen(rows + cols)
    x = torch.randn(rows, cols, device=DEV, generator=g)
    s = ops.split_pack(x, 4)
    t = ops.split_transpose(s)
    assert (t.rows, t.cols, t.shift) == (cols, rows, 4)
    assert torch.equal(ops.split_unpack(t), ops.split_unpack(s).T)          # planes moved, never re-rounded
    # pad columns of the transposed rows (the contraction padding of the weight-gradient product) are zero
    raw = t.buf.view(cols, t.pitch).view(torch.float16).view(cols, -1, 64)
    pad = torch.arange(32, device=DEV)[None, :] + 32 * torch.arange(raw.shape[1], device=DEV)[:, None] >= rows
    assert bool((raw[:, :, :32][:, pad] == 0).all() and (raw[:, :, 32:][:, pad] == 0).all())


@pytest.mark.parametrize("M,N,K,splits", [(4096, 1024, 512, 8), (4096, 512, 1024, 8), (1000, 70, 45, 4), (256, 130, 64, 8), (96, 16, 8, 3),
                                          (4096, 1024, 1024, 1)])
def test_weight_gradient_slabs_on_the_f16_matrix_cores(M, N, K, splits):
    g = _gen(M + N + K)
    dy = torch.randn(M, N, device=DEV, generator=g) * 0.01
    x = torch.randn(M, K, device=DEV, generator=g).clamp_min(0)
    ref = dy.double().T @ x.double()
    dyt = ops.split_transpose(ops.split_pack(dy, 12))
    xt = ops.split_transpose(ops.split_pack(x, 4))
    stride = N * K + 8
    n_cfg = _lib.lib().clava_gemm_f16x3_config_count_()
    for cfg in range(-1, n_cfg if M < 4096 else 0):
        slabs = torch.full((splits, stride), float("nan"), device=DEV)
        used = ops.linear_wgrad_f16x3(dyt, xt, splits, slabs, stride, tile_config=cfg)
        assert used == splits
        got = slabs[:, :N * K].double().sum(0).view(N, K)
        assert relerr(got, ref) < 2e-6, cfg
        assert bool(torch.isnan(slabs[:, N * K:]).all())                    # nothing written beyond each slab's matrix
        # slab s holds exactly the batch range it was given
        per = ((M + 31) // 32 + splits - 1) // splits * 32
        s0 = dy[:per].double().T @ x[:per].double()
        assert relerr(slabs[0, :N * K].view(N, K), s0) < 2e-6, cfg
    # the f32 kernel's slabs add up to the same matrix
    f32 = ops.linear_wgrad(dy, x, splits=splits).double().sum(0)
    assert relerr(got, f32) < 4e-6


@pytest.mark.parametrize("M,N,K,splits", [(4096, 1024, 512, 8), (4096, 512, 1024, 8), (4096, 1024, 1024, 1), (1024, 70, 45, 4),
                                          (256, 130, 64, 8), (96, 16, 8, 3), (4096, 8397, 128, 8), (4096, 128, 8397, 8)])
def test_weight_gradient_slabs_from_batch_major_operands(M, N, K, splits):
    """clava_gemm_wgrad_f16x3_tn reads dy [M, N] and x [M, K] as they lie in memory (LDS tiles read transposed): same slabs
    as the product of the transposed copies (identical planes, identical products; only the order of float32 additions
    inside a k block of 32 may differ) and float64 to 2e-6; ragged tile edges, pitches beyond the row, nothing written
    outside a slab's matrix."""
    g = _gen(M + N + K + 1)
    dy = torch.randn(M, N, device=DEV, generator=g) * 0.01
    x = torch.randn(M, K, device=DEV, generator=g).clamp_min(0)
    ref = dy.double().T @ x.double()
    dys, xs = ops.split_pack(dy, 12), ops.split_pack(x, 4)
    stride = N * K + 8
    slabs = torch.full((splits, stride), float("nan"), device=DEV)
    used = ops.linear_wgrad_f16x3_tn(dys, xs, splits, slabs, stride)
    assert used == splits
    got = slabs[:, :N * K].double().sum(0).view(N, K)
    assert relerr(got, ref) < 2e-6
    assert bool(torch.isnan(slabs[:, N * K:]).all())
    per = (M // 32 + splits - 1) // splits * 32
    assert relerr(slabs[0, :N * K].view(N, K), dy[:per].double().T @ x[:per].double()) < 2e-6
    old = torch.zeros((splits, stride), device=DEV)
    ops.linear_wgrad_f16x3(ops.split_transpose(dys), ops.split_transpose(xs), splits, old, stride)
    assert relerr(slabs[:, :N * K], old[:, :N * K]) < 5e-7
    with pytest.raises(_lib.ClavaHipError):
        ops.linear_wgrad_f16x3_tn(ops.split_pack(dy[:M - 1], 12), ops.split_pack(x[:M - 1], 4), splits, slabs, stride)


@pytest.mark.parametrize("d_in", [6, 300])
def test_split_training_backward_with_f16_weight_gradients_matches_the_float32_backward(d_in):
    """Batch 4096 (the configs' training batch): the whole backward into the gradient slabs, hidden-block weight gradients
    on the split-f16 kernels, against the float32-kernel backward of the same step.  d_in = 300: the wide-table case, where
    proj and head (forward, input gradient, weight gradient) run on the split-f16 kernels too."""
    from clavaddpm_amd import training as tr
    torch.manual_seed(0)
    net = MLPDiffusion(d_in, 0, "none", {"d_layers": [256, 512, 512, 256], "dropout": 0.0}).to(DEV).train()
    B = 4096
    g = _gen(5)
    x_in = torch.randn(B, d_in, device=DEV, generator=g)
    x_in[:, d_in // 2:] = torch.where(x_in[:, d_in // 2:] > 1.0, 0.0, -69.07755279541016)     # log-one-hot-like columns
    t = torch.randint(0, 1000, (B,), device=DEV, generator=g)
    d_out = torch.randn(B, d_in, device=DEV, generator=g) / B
    layout = tr.FlatLayout(net)
    slabs32 = torch.zeros((tr.GRAD_SPLITS, layout.total), device=DEV)
    out32, saved32 = tr.denoiser_forward(net, x_in, t)
    gs = [None] * 6
    tr.denoiser_backward(net, saved32, d_out, layout, slabs32, collect=gs)
    st = tr.SplitTrainState()
    ws, _, _ = net.stack()
    st.calibrate(ws, saved32[4], gs, x_in=x_in, d_out=d_out)
    assert (d_in >= tr.WIDE_END_LAYERS) == (d_in == 300)
    res = {}
    for on in (False, True):        # same split-f16 forward / input-gradient chain; only the weight-gradient kernels differ
        tr.WGRAD_F16X3 = on
        try:
            slabs16 = torch.full((tr.GRAD_SPLITS, layout.total), float("nan"), device=DEV)
            out16, saved16 = tr.denoiser_forward_split(net, x_in, t, st)
            tr.denoiser_backward_split(net, saved16, d_out, layout, slabs16, st)
            torch.cuda.synchronize()
        finally:
            tr.WGRAD_F16X3 = True
        res[on] = slabs16.double().sum(0)
    assert int(st.flag.item()) == 0
    assert (saved16[6] is not None) == (d_in == 300)
    assert relerr(out16, out32) < 1e-5
    g32 = slabs32.double().sum(0)
    for name in layout.names:
        a, b, c = layout.view(res[True], name), layout.view(res[False], name), layout.view(g32, name)
        assert bool(torch.isfinite(a).all()), name                          # every slab of every parameter was written
        assert relerr(a, b) < 4e-6, name
        # against the all-float32 backward the two chains differ where a pre-activation sits within rounding of the ReLU
        # kink: one flipped mask moves a sum of 4096 random-sign terms by ~1/64 of its size
        assert relerr(a, c) < 5e-2, name


@pytest.mark.parametrize("B,d_in", [(4096, 2), (1003, 7), (517, 130), (16, 200)])
def test_fused_training_front_matches_the_separate_kernels(B, d_in):
    """csrc/train_front.hip (timestep embedding, time MLP, proj + emb, split pack of h in one launch) against the six
    launches it replaces (tab_ddpm/modules.py:445-459).  te is bit-equal (same cosf / sinf on the same argument); the
    products differ in summation order only: 2e-6 of the largest value, the float32 kernels' own bar.  Row counts off the
    16-row tile and input widths off the 128-column pass are covered."""
    from clavaddpm_amd import training as tr
    torch.manual_seed(1)
    net = MLPDiffusion(d_in, 0, "none", {"d_layers": [256, 256], "dropout": 0.0}).to(DEV).train()
    g = _gen(7)
    x_in = torch.randn(B, d_in, device=DEV, generator=g)
    t = torch.randint(0, 1000, (B,), device=DEV, generator=g)
    out32, saved32 = tr.denoiser_forward(net, x_in, t)
    gs = [None] * 4
    layout = tr.FlatLayout(net)
    tr.denoiser_backward(net, saved32, torch.randn(B, d_in, device=DEV, generator=g) / B, layout,
                         torch.zeros((tr.GRAD_SPLITS, layout.total), device=DEV), collect=gs)
    st = tr.SplitTrainState()
    st.calibrate(net.stack()[0], saved32[4], gs)
    res = {}
    for fused in (False, True):
        tr.FRONT_FUSED = fused
        try:
            for wgrad in (True, False):                  # False: the float32 copy of h is written too
                tr.WGRAD_F16X3 = wgrad
                out, saved = tr.denoiser_forward_split(net, x_in, t, st)
                torch.cuda.synchronize()
                res[fused, wgrad] = (out, saved)
        finally:
            tr.FRONT_FUSED, tr.WGRAD_F16X3 = True, True
    assert int(st.flag.item()) == 0
    for wgrad in (True, False):
        (out_a, sa), (out_b, sb) = res[True, wgrad], res[False, wgrad]
        assert torch.equal(sa[1], sb[1])                                            # te
        assert relerr(sa[2], sb[2]) < 2e-6 and relerr(sa[3], sb[3]) < 2e-6          # z1, a1
        ha, hb = ops.split_unpack(sa[5][0]), ops.split_unpack(sb[5][0])
        assert relerr(ha, hb) < 2e-6
        assert relerr(ha, saved32[4][0]) < 2e-6
        if not wgrad and B >= 32:
            assert relerr(sa[4][0], sb[4][0]) < 2e-6                                # float32 h
        assert relerr(out_a, out_b) < 1e-5 and relerr(out_a, out32) < 1e-5
