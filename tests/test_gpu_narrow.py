"""GPU parity of the narrow end-layer kernels of the training step (csrc/narrow.hip) through the C ABI, against float64
torch references of the same products (tab_ddpm/modules.py:460 and its autograd backward).  Bar: 2e-6 of the largest
output, the float32 kernels' own bar; split outputs are compared after unpacking (float32-level by construction)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import relerr

from clavaddpm_amd import ops
from clavaddpm_amd import training as tr

DEV = "cuda"


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


@pytest.mark.parametrize("B,K,D", [(4096, 512, 2), (1000, 512, 7), (33, 64, 16), (5, 1024, 1), (257, 260, 3)])
def test_narrow_head_matches_float64(B, K, D):
    g = _gen(B + D)
    h = torch.randn(B, K + 4, device=DEV, generator=g)[:, :K]          # rows with a stride beyond K
    w = torch.randn(D, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(D, device=DEV, generator=g)
    out = tr.narrow_head(h, w, b)
    ref = h.double() @ w.double().T + b.double()
    assert relerr(out, ref) < 2e-6


@pytest.mark.parametrize("B,D,J,mask,f32", [(4096, 2, 512, True, False), (999, 7, 130, True, True), (64, 16, 32, False, True),
                                            (31, 1, 77, True, True)])
def test_narrow_dgrad_matches_float64_and_packs_split_rows(B, D, J, mask, f32):
    g = _gen(B + J)
    dy = torch.randn(B, D + 3, device=DEV, generator=g)[:, :D] * 1e-3
    w = torch.randn(D, J, device=DEV, generator=g)
    act = torch.randn(B, J, device=DEV, generator=g).clamp_min(0.0) if mask else None
    flag = ops.overflow_flag(DEV)
    out = ops.split_empty(B, J, 0, DEV)
    out.buf.fill_(0xFF)                                                 # pad columns must be WRITTEN as zeros
    gf = tr.narrow_dgrad(dy, w, act, 0.0, f32, out, 9, flag)
    ref = dy.double() @ w.double()
    if mask:
        ref = ref * (act > 0)
    assert int(flag.item()) == 0
    assert relerr(ops.split_unpack(out), ref) < 2e-6
    raw = out.buf.view(B, out.pitch)[:, : 4 * ((J + 31) // 32 * 32)].view(B, -1, 2, 32, 2)   # [row][block][plane][col][byte]
    pad = (torch.arange(raw.shape[1] * 32, device=DEV).view(-1, 32) >= J)
    assert int(raw.permute(0, 2, 1, 3, 4)[:, :, pad].abs().max() if pad.any() else 0) == 0
    if f32:
        assert relerr(gf, ref) < 2e-6
    else:
        assert gf is None
    # a value beyond the float16 range at this scale raises the flag
    dy2 = dy.clone()
    dy2[0, 0] = 1e6
    tr.narrow_dgrad(dy2, w, None, 0.0, False, out, 9, flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("B,D,J,splits", [(4096, 2, 512, 8), (1000, 7, 130, 8), (300, 16, 64, 3), (5, 1, 10, 8)])
def test_narrow_wgrad_slabs_match_float64_in_both_layouts(B, D, J, splits):
    g = _gen(B + J + D)
    A = torch.randn(B, D + 1, device=DEV, generator=g)[:, :D]
    Bm = torch.randn(B, J + 2, device=DEV, generator=g)[:, :J]
    ref = A.double().T @ Bm.double()                                    # [D][J]
    total = D * J + D + J + 8
    for layout in ("head", "proj"):
        slabs = torch.full((splits, total), float("nan"), device=DEV)
        dw, dbn, dbw = slabs[:, :D * J], slabs[:, D * J:], slabs[:, D * J + D:]
        if layout == "head":
            tr.narrow_wgrad(A, Bm, splits, dw, J, 1, dbn, dbw, total)
            got = slabs[:, :D * J].double().sum(0).view(D, J)
        else:
            tr.narrow_wgrad(A, Bm, splits, dw, 1, D, dbn, dbw, total)
            got = slabs[:, :D * J].double().sum(0).view(J, D).T
        assert bool(torch.isfinite(slabs[:, :D * J + D + J]).all())    # every slab written, empty batch ranges as zeros
        assert relerr(got, ref) < 2e-6
        assert relerr(slabs[:, D * J:D * J + D].double().sum(0), A.double().sum(0)) < 2e-6
        assert relerr(slabs[:, D * J + D:D * J + D + J].double().sum(0), Bm.double().sum(0)) < 2e-6


def _cat_table(K, n_num, B, seed):
    import numpy as np
    from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
    K = np.array(K)
    D = n_num + int(K.sum())
    net = MLPDiffusion(D, 0, "none", {"d_layers": [64, 64], "dropout": 0.0}).to(DEV)
    d = GaussianMultinomialDiffusion(K, n_num, net, num_timesteps=100, device=torch.device(DEV)).to(DEV)
    g = _gen(seed)
    idx = torch.stack([torch.randint(0, int(k), (B,), device=DEV, generator=g) for k in K], 1).to(torch.int32).contiguous()
    return d, net, idx, D, g


@pytest.mark.parametrize("K,n_num,B,splits", [([2, 7, 130, 300, 64], 3, 1000, 8), ([1000, 2], 0, 4096, 8), ([5], 100, 77, 3)])
def test_cat_wgrad_matches_the_dense_product_with_the_log_one_hot_image(K, n_num, B, splits):
    """csrc/cat_wgrad.hip: dW[:, n_num + off_f + k] = L * (sum_b g[b] - sum_{b: idx[b][f] = k} g[b]) against g^T @
    index_to_log_onehot(idx) in float64 (tab_ddpm/utils.py:127-134; weight half of scripts/train.py:37); class ranges beyond
    128 (several workgroups per feature), batch ranges with a short last one, columns outside the categorical block untouched."""
    import ctypes
    from clavaddpm_amd._lib import c_float, c_int, check, lib, ptr, stream_ptr
    d, net, idx, D, g = _cat_table(K, n_num, B, sum(K) + B)
    st = tr.SplitTrainState()
    gp = st.gather_plan(d, DEV)
    gr = torch.randn(B, 128, device=DEV, generator=g)
    total = 128 * D + 128
    slabs = torch.full((splits, total), 7.0, device=DEV)
    rows_per = (B + splits - 1) // splits
    bsl = slabs[:, 128 * D:]
    for s_ in range(splits):
        bsl[s_] = gr[s_ * rows_per:(s_ + 1) * rows_per].sum(0)
    check(lib().clava_cat_wgrad_f32(ptr(gr), tr.c_ll(128), ptr(bsl), c_int(B), c_int(128), ptr(idx), c_int(gp["m"]), ptr(d._offsets_i32),
                                    ptr(gp["wf"]), ptr(gp["wk"]), c_int(gp["nwg"]), c_int(splits), c_float(gp["log_tiny"]), ptr(slabs),
                                    tr.c_ll(D), c_int(n_num), tr.c_ll(total), stream_ptr()), "clava_cat_wgrad_f32")
    x = torch.full((B, sum(K)), gp["log_tiny"], device=DEV, dtype=torch.float64)
    x.scatter_(1, idx.long() + d._offsets_i32.long()[:-1][None, :], 0.0)
    ref = gr.double().T @ x
    got = slabs[:, :128 * D].view(splits, 128, D)
    assert relerr(got[:, :, n_num:].double().sum(0), ref) < 2e-6
    assert bool((got[:, :, :n_num] == 7.0).all())                       # the numeric block belongs to another kernel


@pytest.mark.parametrize("K,n_num,B", [([4, 9, 200], 5, 333), ([3], 0, 64)])
def test_cat_input_with_per_row_embedding_matches_the_dense_first_layer(K, n_num, B):
    """csrc/cat_input.hip in its training form: per-row time embedding, split rows AND float32 output, against proj on the
    dense [B, n_num + sum K] input (tab_ddpm/modules.py:459) in float64."""
    from clavaddpm_amd._lib import c_float, c_int, check, lib, ptr, stream_ptr
    d, net, idx, D, g = _cat_table(K, n_num, B, 5 * B)
    st = tr.SplitTrainState()
    gp = st.gather_plan(d, DEV)
    w = torch.randn(128, D, device=DEV, generator=g) / D ** 0.5
    b = torch.randn(128, device=DEV, generator=g)
    emb = torch.randn(B, 128, device=DEV, generator=g)
    xn = torch.randn(B, max(n_num, 1), device=DEV, generator=g)[:, :n_num].contiguous()
    st.gather_weights(w, b, n_num)
    out = ops.split_empty(B, 128, 0, DEV)
    flag = ops.overflow_flag(DEV)
    h32 = torch.empty(B, 128, device=DEV)
    check(lib().clava_cat_input_f32(ptr(xn) if n_num else None, c_int(max(n_num, 1)), ptr(idx), c_int(B), c_int(n_num), c_int(gp["m"]), c_int(128),
                                    ptr(d._offsets_i32), ptr(st._wt0) if n_num else None, ptr(st._wt0[n_num:]), ptr(st._cat_bias), ptr(emb),
                                    tr.c_ll(128), c_float(gp["log_tiny"]), c_int(0), ptr(out.buf), tr.c_ll(out.pitch), c_int(3), ptr(h32),
                                    tr.c_ll(128), ptr(flag), stream_ptr()), "clava_cat_input_f32")
    out.shift = 3
    x = torch.full((B, sum(K)), gp["log_tiny"], device=DEV, dtype=torch.float64)
    x.scatter_(1, idx.long() + d._offsets_i32.long()[:-1][None, :], 0.0)
    ref = torch.cat([xn.double(), x], 1) @ w.double().T + b.double() + emb.double()
    assert int(flag.item()) == 0
    assert relerr(h32, ref) < 4e-6 and relerr(ops.split_unpack(out), ref) < 4e-6


def test_nan_and_inf_logits_give_in_range_indices_through_step_and_gather_layer():
    """Regression for the round-2 GPU fault: a chain whose activations overflowed hands the categorical step
    kernel all-NaN / infinite logits.  clava_multinomial_step_f32 must still write an index inside every feature's block
    (class 0 when every score of a block is NaN, what torch.argmax of the reference's `:506-516` returns), and clava_cat_input_f32 must
    accept those indices.  Blocks of every lane-group width (2 .. 300 classes), rows of NaN, +inf, -inf and mixed."""
    from clavaddpm_amd._lib import c_float, c_int, check, lib, ptr, stream_ptr
    K, n_num, B = [2, 7, 20, 40, 130, 300], 3, 256
    d, net, idx, D, g = _cat_table(K, n_num, B, 77)
    sumK = sum(K)
    logits = torch.randn(B, sumK, device=DEV, generator=g)
    logits[0::4] = float("nan")                       # whole rows NaN
    logits[1::4] = float("inf")
    logits[2::8] = float("-inf")
    logits[3::8, ::3] = float("nan")                  # NaN mixed with finite values inside every block
    t = torch.randint(0, 100, (B,), device=DEV, generator=g)
    t[:8] = 0
    offs = d._offsets_i32.long().cpu()
    for uniforms in (torch.rand(B, sumK, device=DEV, generator=g), None):      # injected and in-kernel draws
        out, _ = d.p_sample_index(logits, idx, t, uniforms)
        torch.cuda.synchronize()
        o = out.cpu().long()
        for f, k in enumerate(K):
            assert int(o[:, f].min()) >= 0 and int(o[:, f].max()) < k, (f, k, int(o[:, f].min()), int(o[:, f].max()))
        # all-NaN blocks at t = 0 (no log_add_exp between the logits and the scores): class 0, torch.argmax's answer; at other
        # t the kernel's branch-free exp may launder a NaN into a finite score - any class INSIDE the block is acceptable
        assert bool((o[0:8:4] == 0).all())
        # the gather layer on those indices (the launch that faulted): finite wherever its own inputs are
        st = tr.SplitTrainState()
        gp = st.gather_plan(d, DEV)
        w = torch.randn(128, D, device=DEV, generator=g) / D ** 0.5
        b = torch.randn(128, device=DEV, generator=g)
        emb = torch.randn(B, 128, device=DEV, generator=g)
        xn = torch.randn(B, n_num, device=DEV, generator=g)
        st.gather_weights(w, b, n_num)
        res = ops.split_empty(B, 128, 0, DEV)
        flag = ops.overflow_flag(DEV)
        h32 = torch.empty(B, 128, device=DEV)
        check(lib().clava_cat_input_f32(ptr(xn), c_int(n_num), ptr(out), c_int(B), c_int(n_num), c_int(gp["m"]), c_int(128),
                                        ptr(d._offsets_i32), ptr(st._wt0), ptr(st._wt0[n_num:]), ptr(st._cat_bias), ptr(emb),
                                        tr.c_ll(128), c_float(gp["log_tiny"]), c_int(0), ptr(res.buf), tr.c_ll(res.pitch), c_int(3), ptr(h32),
                                        tr.c_ll(128), ptr(flag), stream_ptr()), "clava_cat_input_f32")
        torch.cuda.synchronize()
        x = torch.full((B, sumK), gp["log_tiny"], device=DEV, dtype=torch.float64)
        x.scatter_(1, out.long() + d._offsets_i32.long()[:-1][None, :], 0.0)
        ref = torch.cat([xn.double(), x], 1) @ w.double().T + b.double() + emb.double()
        assert bool(torch.isfinite(h32).all()) and relerr(h32, ref) < 4e-6


def test_optimizer_with_device_scalars_is_bit_equal_to_the_argument_form():
    """clava_adamw_hyper_f32 + clava_adamw_ema_hyper_f32 (a captured step's optimizer) against clava_adamw_ema_f32 on the
    same state: identical bits for parameters, moments and EMA."""
    import ctypes
    from clavaddpm_amd._lib import c_float, c_int, check, lib, ptr, stream_ptr
    g = _gen(3)
    n, S = 100003, 4
    p0, m0 = torch.randn(n, device=DEV, generator=g), torch.randn(n, device=DEV, generator=g) * 1e-3
    v0 = torch.rand(n, device=DEV, generator=g) * 1e-6
    slabs = torch.randn(S, n, device=DEV, generator=g) * 1e-2
    cd = ctypes.c_double
    hp = dict(lr=3e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5, step=17, ema=0.999)
    res = []
    for form in ("args", "device"):
        p, e, m, v = p0.clone(), p0.clone(), m0.clone(), v0.clone()
        if form == "args":
            check(lib().clava_adamw_ema_f32(ptr(p), ptr(e), ptr(m), ptr(v), ptr(slabs), c_int(S), tr.c_ll(n), c_float(1.0), tr.c_ll(n), cd(hp["lr"]),
                                            cd(hp["b1"]), cd(hp["b2"]), cd(hp["eps"]), cd(hp["wd"]), tr.c_ll(hp["step"]), cd(hp["ema"]), stream_ptr()),
                  "clava_adamw_ema_f32")
        else:
            host = (ctypes.c_float * 8)()
            check(lib().clava_adamw_hyper_f32(cd(hp["lr"]), cd(hp["b1"]), cd(hp["b2"]), cd(hp["eps"]), cd(hp["wd"]), tr.c_ll(hp["step"]), host),
                  "clava_adamw_hyper_f32")
            hd = torch.tensor(list(host), device=DEV, dtype=torch.float32)
            check(lib().clava_adamw_ema_hyper_f32(ptr(p), ptr(e), ptr(m), ptr(v), ptr(slabs), c_int(S), tr.c_ll(n), c_float(1.0), tr.c_ll(n),
                                                  ptr(hd), cd(hp["ema"]), stream_ptr()), "clava_adamw_ema_hyper_f32")
        res.append((p, e, m, v))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert not torch.equal(res[0][0], p0)


@pytest.mark.parametrize("B", [4096, 37])
def test_time_embedding_backward_kernel_matches_float64(B):
    """clava_train_time_bwd_f32: d_z1 = SiLU'(z1) * (g @ W2) against float64 (autograd of tab_ddpm/modules.py:445-449)."""
    from clavaddpm_amd._lib import c_int, check, lib, ptr, stream_ptr
    g = _gen(B)
    gr = torch.randn(B, 130, device=DEV, generator=g)[:, :128] * 1e-3
    w2 = torch.randn(128, 128, device=DEV, generator=g) / 11
    z1 = torch.randn(B, 128, device=DEV, generator=g) * 3
    dz = torch.empty_like(z1)
    check(lib().clava_train_time_bwd_f32(ptr(gr), tr.c_ll(gr.stride(0)), c_int(B), c_int(128), ptr(w2), ptr(z1), ptr(dz), stream_ptr()),
          "clava_train_time_bwd_f32")
    zd = z1.double()
    sg = torch.sigmoid(zd)
    ref = (gr.double() @ w2.double()) * (sg * (1 + zd * (1 - sg)))
    assert relerr(dz, ref) < 2e-6


@pytest.mark.parametrize("M,K,N", [(20000, 512, 2), (777, 128, 5), (64, 1024, 16), (33, 40, 1)])
def test_narrow_product_from_split_rows_matches_float64(M, K, N):
    """clava_gemm_f16x3 with at most 16 outputs and a float32 result takes the memory-bound kernel of csrc/narrow.hip (the
    denoiser head and the guidance classifier's last input-gradient product of the sampling chain): against float64 at the
    split products' own bar, and against the tiled MFMA kernel (forced by a tile configuration) on the same operands."""
    g = _gen(M + K + N)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    xs, wsp = ops.split_pack(x, 4), ops.split_pack(w, 9)
    out = ops.linear_f16x3(xs, wsp, b, out_f32=True)
    ref = x.double() @ w.double().T + b.double()
    assert relerr(out, ref) < 2e-6
    tiled = ops.linear_f16x3(xs, wsp, b, out_f32=True, tile_config=7)
    assert relerr(out, tiled) < 2e-6


@pytest.mark.parametrize("K,n_num,B", [([2, 3, 8, 9, 16, 17, 31, 33, 64, 65, 130, 257, 300, 1100], 5, 96), ([4, 7], 0, 33), ([40], 3, 7)])
def test_loss_kernels_write_the_gradient_as_split_rows(K, n_num, B):
    """clava_gauss_mse_loss_split_f32 / clava_multinomial_loss_split_f32: the split rows are byte-for-byte what
    clava_split_pack_f32 makes of the float32 gradient (same scale, same two roundings), pad columns untouched (zero);
    blocks of every lane-group width (8 / 16 / 32 lanes packed, 64 lanes in registers and streamed)."""
    import numpy as np
    from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
    sk = sum(K)
    net = MLPDiffusion(n_num + sk, 0, "none", {"d_layers": [32, 32], "dropout": 0.0}).to(DEV)
    d = GaussianMultinomialDiffusion(np.array(K), n_num, net, num_timesteps=1000, device=torch.device(DEV)).to(DEV)
    g = _gen(B)
    x = torch.cat([torch.randn(B, n_num, device=DEV, generator=g)] +
                  [torch.randint(0, k, (B, 1), device=DEV, generator=g).float() for k in K], 1)
    word = torch.zeros(1, device=DEV, dtype=torch.int32)
    t, noise, x_in, idx0, idxt = tr.draw_noised_input(d, x, 7, word, dense=False)
    t[0] = 0                                                           # the decoder-NLL branch
    out = torch.randn(B, n_num + sk, device=DEV, generator=g) * 3.0
    lm, lg, d_out = tr.loss_and_grad(d, out, noise, idx0, idxt, t)
    shift = 4
    flag = ops.overflow_flag(DEV)
    ds = ops.split_empty(B, n_num + sk, shift, DEV)
    lm2, lg2, ret = tr.loss_and_grad(d, out, noise, idx0, idxt, t, split_out=ds, overflow=flag)
    assert ret is ds
    assert float(lm2) == float(lm) and (lg is None or float(lg2) == float(lg))
    ref = ops.split_pack(d_out.contiguous(), shift, flag)
    assert int(flag.item()) == 0
    assert torch.equal(ds.buf, ref.buf)
    # an overflowing scale raises the flag
    big = ops.split_empty(B, n_num + sk, 60, DEV)
    tr.loss_and_grad(d, out, noise, idx0, idxt, t, split_out=big, overflow=flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("D,n_num", [(8397, 100), (130, 0), (64, 64), (77, 5)])
def test_gather_form_weight_operands_match_torch(D, n_num):
    """clava_cat_input_weights_f32: proj's weight transposed exactly, and b + log(1e-30) * (float64 row sums of the
    categorical columns) rounded to float32 (the constant part of the log-one-hot columns, tab_ddpm/utils.py:127-134)."""
    g = _gen(D)
    w = torch.randn(128, D + 3, device=DEV, generator=g)[:, :D]
    b = torch.randn(128, device=DEV, generator=g)
    st = tr.SplitTrainState()
    st.gather_weights(w, b, n_num)
    log_tiny = float(torch.log(torch.tensor(1e-30, dtype=torch.float32)))
    assert torch.equal(st._wt0, w.t().contiguous())
    ref = b.double() + log_tiny * w[:, n_num:].double().sum(dim=1)
    assert float((st._cat_bias.double() - ref).abs().max()) <= 1e-7 * float(ref.abs().max())
