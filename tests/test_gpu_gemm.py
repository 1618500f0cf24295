"""GPU parity of the fp32 MFMA GEMM family, through the C ABI (ctypes), against a float64 reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from clavaddpm_amd import ops


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (257, 130, 70), (4096, 1024, 1024), (20000, 512, 128),
                                   (333, 15, 512), (1000, 128, 15), (64, 64, 8), (5, 3, 1)])
def test_linear_asymmetric(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)      # asymmetric operands: a transposed C map cannot pass
    b = torch.randn(N, device="cuda", generator=g)
    ref = x.double() @ w.double().T + b.double()
    y = ops.linear(x, w, b)
    assert _rel(y, ref) < 2e-6
    y = ops.linear(x, w, b, act="relu")
    assert _rel(y, ref.clamp_min(0)) < 2e-6


def test_linear_identity_operand():
    # A = I with an asymmetric B: output must be B^T exactly (catches row/col swaps in the C write)
    n = 96
    eye = torch.eye(n, device="cuda")
    w = torch.arange(n * n, device="cuda", dtype=torch.float32).reshape(n, n)
    assert torch.equal(ops.linear(eye, w), w.T.contiguous())


def test_linear_epilogue_variants():
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 300, 200, 96
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    R = torch.randn(M, N, device="cuda", generator=g)
    r1 = torch.randn(N, device="cuda", generator=g)
    sc = torch.rand(K, device="cuda", generator=g) + 0.5
    sh = torch.randn(K, device="cuda", generator=g)
    base = x.double() @ w.double().T + b.double()
    assert _rel(ops.linear(x, w, b, rowadd=R), base + R.double()) < 2e-6
    assert _rel(ops.linear(x, w, b, rowadd=r1), base + r1.double()) < 2e-6
    z = base + R.double()
    assert _rel(ops.linear(x, w, b, act="leaky_relu", rowadd=R), torch.where(z > 0, z, 0.01 * z)) < 2e-6
    assert _rel(ops.linear(x, w, b, act="silu"), base * torch.sigmoid(base)) < 2e-6
    xa = x.double() * sc.double() + sh.double()
    assert _rel(ops.linear(x, w, b, in_scale=sc, in_shift=sh), xa @ w.double().T + b.double()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 512), (300, 7, 128), (129, 128, 15)])
def test_dgrad(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(11)
    dy = torch.randn(M, N, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    a = torch.randn(M, K, device="cuda", generator=g)
    cs = torch.rand(K, device="cuda", generator=g) + 0.5
    ref = dy.double() @ w.double()
    assert _rel(ops.linear_dgrad(dy, w), ref) < 2e-6
    m = torch.where(a > 0, 1.0, 0.01).double()
    assert _rel(ops.linear_dgrad(dy, w, act_out=a, slope=0.01, col_scale=cs, alpha=1.5), 1.5 * ref * m * cs.double()) < 2e-6


@pytest.mark.parametrize("M,N,K,S", [(4096, 1024, 512, 4), (4096, 128, 15, 8), (1000, 14, 64, 3), (4096, 512, 1024, 1)])
def test_wgrad_slabs(M, N, K, S):
    g = torch.Generator(device="cuda").manual_seed(13)
    dy = torch.randn(M, N, device="cuda", generator=g)
    x = torch.randn(M, K, device="cuda", generator=g)
    slabs = ops.linear_wgrad(dy, x, splits=S)
    assert slabs.shape == (S, N, K)
    assert _rel(slabs.sum(0), dy.double().T @ x.double()) < 3e-6
