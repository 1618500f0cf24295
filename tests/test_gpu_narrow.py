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
