"""GPU parity of the GMM E-step against vectors captured from scikit-learn (parity otherwise unpinned: third-party code)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden

from clavaddpm_amd.clustering import gmm_estep
from oracle import gmm as ogmm

DEV = "cuda"


def test_g9_gmm_estep_golden():
    g = golden("g9_gmm.npz")
    lpn, resp, lab = gmm_estep(g["X"].to(DEV), g["weights"].to(DEV), g["means"].to(DEV), g["precisions_chol"].to(DEV))
    # float64 with a different summation order than BLAS dgemm; the expanded square mu^2 P - 2 x mu P + x^2 P cancels
    # large terms (precisions up to 1e6 on the scaled key column), so agreement is ~1e-11 relative, not bitwise
    np.testing.assert_allclose(lpn.cpu().numpy(), g["log_prob_norm"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(resp.cpu().numpy(), g["log_resp"].numpy(), rtol=1e-8, atol=1e-8)
    assert np.array_equal(lab.cpu().numpy().astype(np.int64), g["labels"].numpy())        # cluster assignments: exact


@pytest.mark.parametrize("N,d,k", [(1, 1, 1), (63, 5, 2), (1000, 33, 16), (4097, 70, 17), (20000, 131, 50), (300, 9, 64)])
def test_gmm_estep_vs_oracle(N, d, k):
    rng = np.random.RandomState(N + d + k)
    X = rng.standard_normal((N, d)) * 2.0
    means = rng.standard_normal((k, d))
    pc = rng.uniform(0.3, 3.0, (k, d))
    w = rng.dirichlet(np.ones(k))
    lpn_r, resp_r, lab_r = ogmm.estep(X, w, means, pc)
    f = lambda a: torch.from_numpy(a).to(DEV)
    lpn, resp, lab = gmm_estep(f(X), f(w), f(means), f(pc))
    np.testing.assert_allclose(lpn.cpu().numpy(), lpn_r, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(resp.cpu().numpy(), resp_r, rtol=1e-9, atol=1e-8)
    assert np.array_equal(lab.cpu().numpy().astype(np.int64), lab_r)
    # size-independent property: responsibilities of every row sum to one
    assert float((resp.exp().sum(1) - 1).abs().max()) < 1e-9


def test_gmm_limits_and_errors():
    from clavaddpm_amd._lib import ClavaHipError
    X = torch.zeros((4, 3), device=DEV, dtype=torch.float64)
    with pytest.raises(ClavaHipError, match="exceeds"):
        gmm_estep(X, torch.ones(65, device=DEV, dtype=torch.float64) / 65, torch.zeros((65, 3), device=DEV, dtype=torch.float64),
                  torch.ones((65, 3), device=DEV, dtype=torch.float64))


def test_hip_gaussian_mixture_fit_predict_matches_sklearn():
    from sklearn.mixture import GaussianMixture
    from clavaddpm_amd.clustering import HipGaussianMixture
    rng = np.random.RandomState(0)
    centers = rng.uniform(-4, 4, (5, 6))
    X = np.concatenate([c + 0.3 * rng.standard_normal((400, 6)) for c in centers])
    kw = dict(n_components=5, covariance_type="diag", init_params="k-means++", tol=1e-4, random_state=0)
    ref = GaussianMixture(**kw).fit(X)
    got = HipGaussianMixture(**kw).fit(X)
    assert got.n_iter_ == ref.n_iter_
    np.testing.assert_allclose(got.means_, ref.means_, rtol=1e-8, atol=1e-10)
    assert np.array_equal(got.predict(X), ref.predict(X))


def test_row_sharded_estep_equals_the_single_call():
    from clavaddpm_amd import parallel
    rng = np.random.RandomState(3)
    N, d, k = 5001, 20, 7
    X, means, pc, w = rng.standard_normal((N, d)), rng.standard_normal((k, d)), rng.uniform(0.5, 2.0, (k, d)), rng.dirichlet(np.ones(k))
    f = lambda a: torch.from_numpy(a).to(DEV)
    lpn, _, lab = gmm_estep(f(X), f(w), f(means), f(pc), want_resp=False)
    parts = [parallel.gmm_estep_rows(X, w, means, pc, rank=r, world=3, gather=False) for r in range(3)]
    assert torch.equal(torch.cat([p[1] for p in parts]), lab.cpu().long())
    assert torch.equal(torch.cat([p[0] for p in parts]), lpn.cpu())        # rows are independent: bit-identical


# ------------------------------------------------------------------------------------------------ M-step / EM loop
def _mstep(X, log_resp, lpn=None, reg=1e-6, ldx=None):
    from clavaddpm_amd.clustering import gmm_mstep, unpack_mstep
    Xd = torch.from_numpy(X).to(DEV)
    if ldx is not None:                      # rows embedded in a wider buffer
        wide = torch.zeros((X.shape[0], ldx), device=DEV, dtype=torch.float64)
        wide[:, :X.shape[1]] = Xd
        Xd = wide[:, :X.shape[1]]
    out = gmm_mstep(Xd, torch.from_numpy(log_resp).to(DEV), None if lpn is None else torch.from_numpy(lpn).to(DEV), reg)
    return [t.cpu().numpy() for t in unpack_mstep(out, log_resp.shape[1], X.shape[1])]


def test_g11_mstep_golden():
    g, g9 = golden("g11_gmm_fit.npz"), golden("g9_gmm.npz")
    st, w, m, cv, pc = _mstep(g9["X"].numpy(), g9["log_resp"].numpy(), g9["log_prob_norm"].numpy())
    np.testing.assert_allclose(w, g["m_weights"].numpy(), rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(m, g["m_means"].numpy(), rtol=1e-10, atol=1e-12)
    # covariances subtract means^2 from the second moment: absolute error scales with means^2 * 1e-16
    np.testing.assert_allclose(cv, g["m_cov"].numpy(), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(pc, g["m_prec_chol"].numpy(), rtol=1e-8, atol=1e-12)
    assert abs(st[0] - g9["log_prob_norm"].numpy().mean()) < 1e-12 and st[1] == 0.0


@pytest.mark.parametrize("N,d,k,ldx", [(1, 1, 1, None), (17, 3, 2, None), (1000, 5, 3, 8), (777, 129, 64, None),
                                       (50000, 40, 50, None), (4096, 32, 16, 40), (65, 33, 17, None)])
def test_gmm_mstep_vs_oracle(N, d, k, ldx):
    rng = np.random.RandomState(N + d + k)
    X = rng.standard_normal((N, d)) * 2.0 + rng.uniform(-3, 3, (1, d))
    logits = rng.standard_normal((N, k)) * 3.0
    log_resp = logits - np.log(np.exp(logits).sum(1, keepdims=True))
    log_resp[::7, 0] = -800.0                                  # responsibilities that underflow to zero
    lpn = rng.standard_normal(N)
    w_r, m_r, c_r, pc_r = ogmm.mstep(X, log_resp, reg_covar=1e-6)
    st, w, m, cv, pc = _mstep(X, log_resp, lpn, 1e-6, ldx)
    np.testing.assert_allclose(w, w_r, rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(m, m_r, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(cv, c_r, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(pc, pc_r, rtol=1e-8, atol=1e-10)
    assert abs(st[0] - lpn.mean()) < 1e-12 and st[1] == 0.0
    assert abs(w.sum() - 1.0) < 1e-14                          # size-independent property: weights are a distribution


def test_gmm_mstep_is_bitwise_reproducible_and_flags_bad_covariances():
    rng = np.random.RandomState(5)
    X = rng.standard_normal((30000, 24))
    lr = np.log(rng.dirichlet(np.ones(20), 30000))
    a = _mstep(X, lr)
    b = _mstep(X, lr)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    X[:, 3] = 1.5                                              # a constant column: zero variance without regularisation
    assert _mstep(X, lr, reg=0.0)[0][1] == 1.0
    from clavaddpm_amd.clustering import HipGaussianMixture
    with pytest.raises(ValueError, match="ill-defined empirical covariance"):
        HipGaussianMixture(n_components=3, covariance_type="diag", reg_covar=0.0, random_state=0).fit(X[:2000])


def test_g11_device_em_loop_matches_sklearn_run():
    """`fit` with E-step, M-step and lower bound on the device against scikit-learn's own run (recorded in G11): same
    number of EM iterations, parameters to 1e-8, identical cluster labels."""
    from clavaddpm_amd.clustering import HipGaussianMixture
    g = golden("g11_gmm_fit.npz")
    X = g["X"].numpy()
    gm = HipGaussianMixture(n_components=g["means"].shape[0], covariance_type="diag", init_params="k-means++", tol=1e-4,
                            random_state=3)
    labels = gm.fit_predict(X)
    assert gm.n_iter_ == int(g["n_iter"]) and gm.converged_ == bool(int(g["converged"]))
    # 16 EM iterations of float64 sums in another order than BLAS (E-step agreement is ~1e-10 per call, see above)
    assert abs(gm.lower_bound_ - float(g["lower_bound"])) < 1e-8
    np.testing.assert_allclose(gm.weights_, g["weights"].numpy(), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(gm.means_, g["means"].numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(gm.covariances_, g["cov"].numpy(), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(gm.precisions_cholesky_, g["prec_chol"].numpy(), rtol=1e-7, atol=1e-10)
    assert np.array_equal(labels, g["labels"].numpy()) and np.array_equal(gm.predict(X), g["labels"].numpy())
    # the numpy-facing pieces still work after a device fit
    lpn, lr = gm._estimate_log_prob_resp(X[:100])
    assert lpn.shape == (100,) and lr.shape == (100, gm.n_components)
