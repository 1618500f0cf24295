"""The entry point that allocates nothing (`clava_gmm_estep_noalloc_f64`, static device scratch) against the existing
`clava_gmm_estep_f64` (caller workspace), through the C ABI: two calls in a row, the second smaller, bit-identical
results (same kernels, same tables, only the tables' address differs); the size limit is an error, not a fault."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clavaddpm_amd._lib import ClavaHipError
from clavaddpm_amd.clustering import gmm_estep, gmm_estep_noalloc

DEV = "cuda"


def _case(N, d, k, seed):
    rng = np.random.RandomState(seed)
    f = lambda a: torch.from_numpy(a).to(DEV)
    return (f(rng.standard_normal((N, d)) * 2.0), f(rng.dirichlet(np.ones(k))), f(rng.standard_normal((k, d))),
            f(rng.uniform(0.3, 3.0, (k, d))))


def _run_noalloc(X, w, m, pc, with_resp=True):
    N, k = X.shape[0], m.shape[0]
    # poisoned outputs: a call that silently wrote nothing would be caught
    lpn = torch.full((N,), float("nan"), device=DEV, dtype=torch.float64)
    resp = torch.full((N, k), float("nan"), device=DEV, dtype=torch.float64) if with_resp else None
    lab = torch.full((N,), -7, device=DEV, dtype=torch.int32)
    gmm_estep_noalloc(X, w, m, pc, lpn, resp, lab)          # warm-up: code-object load and the one-time symbol lookup are not the claim
    lpn.fill_(float("nan"))
    lab.fill_(-7)
    if resp is not None:
        resp.fill_(float("nan"))
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    free_before = torch.cuda.mem_get_info()[0]              # hipMemGetInfo: sees a hipMalloc inside the LIBRARY, which torch's
    gmm_estep_noalloc(X, w, m, pc, lpn, resp, lab)          # allocator statistics do not
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before          # the wrapper allocated no device memory either
    assert torch.cuda.mem_get_info()[0] == free_before, "the device's free memory changed across the call"
    return lpn, resp, lab


def test_noalloc_estep_equals_the_workspace_entry_point_twice_in_a_row():
    # first call: the largest table the fast (LDS-resident) kernel takes, 50 components; second call: smaller N, d and k, so the
    # static array still holds the first call's larger tables beyond what the second call rewrites
    for (N, d, k, seed) in [(20000, 128, 50, 1), (777, 9, 5, 2), (4097, 300, 64, 3), (1, 1, 1, 4)]:
        X, w, m, pc = _case(N, d, k, seed)
        lpn_r, resp_r, lab_r = gmm_estep(X, w, m, pc)
        lpn, resp, lab = _run_noalloc(X, w, m, pc)
        assert torch.equal(lpn, lpn_r) and torch.equal(resp, resp_r) and torch.equal(lab, lab_r), (N, d, k)
    # and against the CPU oracle, to the E-step's own bar (labels exact, float64 sums in another order than BLAS)
    from oracle import gmm as ogmm
    X, w, m, pc = _case(3000, 40, 17, 5)
    lpn, resp, lab = _run_noalloc(X, w, m, pc)
    lpn_o, resp_o, lab_o = ogmm.estep(X.cpu().numpy(), w.cpu().numpy(), m.cpu().numpy(), pc.cpu().numpy())
    np.testing.assert_allclose(lpn.cpu().numpy(), lpn_o, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(resp.cpu().numpy(), resp_o, rtol=1e-9, atol=1e-8)
    assert np.array_equal(lab.cpu().numpy().astype(np.int64), lab_o)


def test_noalloc_estep_optional_outputs_and_row_stride():
    X, w, m, pc = _case(1000, 33, 16, 6)
    wide = torch.zeros((1000, 40), device=DEV, dtype=torch.float64)
    wide[:, :33] = X
    lpn_r, _, lab_r = gmm_estep(X, w, m, pc, want_resp=False)
    lpn, resp, lab = _run_noalloc(wide[:, :33], w, m, pc, with_resp=False)
    assert resp is None and torch.equal(lpn, lpn_r) and torch.equal(lab, lab_r)


def test_noalloc_estep_rejects_sizes_beyond_the_static_scratch():
    import ctypes
    from clavaddpm_amd._lib import lib
    L = lib()
    L.clava_gmm_noalloc_max_doubles.restype = ctypes.c_longlong
    cap = int(L.clava_gmm_noalloc_max_doubles())
    assert cap == 2 * 64 * 512 + 3 * 64
    z = lambda *s: torch.zeros(s, device=DEV, dtype=torch.float64)
    o = lambda *s: torch.ones(s, device=DEV, dtype=torch.float64)
    with pytest.raises(ClavaHipError, match="static scratch"):            # 64 x 513 features: one column too many
        gmm_estep_noalloc(z(4, 513), o(64) / 64, z(64, 513), o(64, 513), z(4))
    with pytest.raises(ClavaHipError, match="static scratch"):            # more than 64 components
        gmm_estep_noalloc(z(4, 3), o(65) / 65, z(65, 3), o(65, 3), z(4))
    # the limit itself is accepted (k = 64, d = 512), and a narrow mixture may be wider than 512 features
    X, w, m, pc = _case(130, 512, 64, 7)
    a = _run_noalloc(X, w, m, pc)
    b = gmm_estep(X, w, m, pc)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    X, w, m, pc = _case(100, 2000, 8, 8)
    a = _run_noalloc(X, w, m, pc)
    b = gmm_estep(X, w, m, pc)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
