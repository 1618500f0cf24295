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
