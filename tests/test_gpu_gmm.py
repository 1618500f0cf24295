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


@pytest.mark.parametrize("N,d,k,cuts", [(1000, 5, 3, (0, 400, 1000)), (20001, 33, 50, (0, 1, 9000, 9000, 20001)), (300, 70, 64, (0, 300))])
def test_row_sharded_mstep_totals_sum_to_the_single_call(N, d, k, cuts):
    """per-shard sums (`clava_gmm_mstep_partial_f64`), added (what the RCCL sum all-reduce in
    parallel.gmm_mstep_allreduce does; one rank here), finalised with the global row count - against the one-call M-step
    and the numpy oracle; a shard with zero rows contributes zeros; world 1 goes through the same entry points."""
    from clavaddpm_amd import clustering as cl, parallel
    rng = np.random.RandomState(N + d)
    X = rng.standard_normal((N, d)) * 2.0
    lr_ = rng.standard_normal((N, k))
    lr_ -= np.log(np.exp(lr_).sum(1, keepdims=True))
    lpn = rng.standard_normal(N)
    Xd, ld, pd_ = (torch.from_numpy(a).to(DEV) for a in (X, lr_, lpn))
    tot = torch.zeros(cl.mstep_totals_size(k, d), device=DEV, dtype=torch.float64)
    for a, b in zip(cuts[:-1], cuts[1:]):
        tot += cl.gmm_mstep_partial(Xd[a:b], ld[a:b], pd_[a:b])
    sx, sxx, nk, lp = cl.unpack_mstep_totals(tot.cpu(), k, d)
    o_nk, o_sx, o_sxx, o_lp = ogmm.mstep_totals(X, lr_, lpn)
    np.testing.assert_allclose(sx.numpy(), o_sx, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(sxx.numpy(), o_sxx, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(nk.numpy(), o_nk, rtol=1e-12)
    assert abs(float(lp) - o_lp) < 1e-9
    st, w, m, cv, pc = cl.unpack_mstep(cl.gmm_mstep_finalize(tot, N, k, d).cpu(), k, d)
    st1, w1, m1, cv1, pc1 = cl.unpack_mstep(cl.gmm_mstep(Xd, ld, pd_).cpu(), k, d)
    want = ogmm.mstep(X, lr_)
    for got, one, ora in zip((w, m, cv, pc), (w1, m1, cv1, pc1), want):
        np.testing.assert_allclose(got.numpy(), ora, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(got.numpy(), one.numpy(), rtol=1e-10, atol=1e-11)
    assert abs(float(st[0]) - lpn.mean()) < 1e-12 and float(st[1]) == 0.0
    alone = cl.unpack_mstep(parallel.gmm_mstep_allreduce(Xd, ld, pd_).cpu(), k, d)       # no process group: one shard
    assert all(torch.equal(a_, b_) for a_, b_ in zip(alone, (st1, w1, m1, cv1, pc1)))


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


# ------------------------------------------------------------------------------------------------ k-means++ seeding
@pytest.mark.parametrize("N,d,k,seed", [(600, 10, 6, 0), (4000, 11, 9, 3), (50000, 32, 50, 7), (257, 1, 2, 1), (3000, 70, 25, 5)])
def test_kmeans_plusplus_selects_sklearns_seeds(N, d, k, seed):
    """scikit-learn is the reference here (third-party, importable on the box): same RandomState in, same seed rows out."""
    from sklearn.cluster import kmeans_plusplus as sk_kmpp
    from clavaddpm_amd.clustering import kmeans_plusplus
    rng = np.random.RandomState(100 + seed)
    centers = rng.uniform(-5, 5, (k, d))
    X = centers[rng.randint(0, k, N)] + rng.standard_normal((N, d)) * 0.7
    _, ref = sk_kmpp(X, k, random_state=np.random.RandomState(seed))
    rs = np.random.RandomState(seed)
    got = kmeans_plusplus(torch.from_numpy(X).to(DEV), k, rs)
    assert np.array_equal(got, ref)
    # the generator was consumed exactly as scikit-learn consumes it
    rs_ref = np.random.RandomState(seed)
    sk_kmpp(X, k, random_state=rs_ref)
    assert rs.uniform() == rs_ref.uniform()


def test_kmpp_primitives_vs_numpy():
    import ctypes
    from clavaddpm_amd._lib import c_int, check, lib, ptr, stream_ptr
    rng = np.random.RandomState(2)
    N, d, T = 10007, 13, 5
    X = rng.standard_normal((N, d)) * 3
    Xd = torch.from_numpy(X).to(DEV)
    L = lib()
    L.clava_kmpp_workspace_doubles.restype = ctypes.c_longlong
    ws = torch.empty(int(L.clava_kmpp_workspace_doubles()), device=DEV, dtype=torch.float64)
    xsq = torch.empty(N, device=DEV, dtype=torch.float64)
    check(L.clava_kmpp_row_norms_f64(ptr(Xd), ctypes.c_longlong(d), ctypes.c_longlong(N), c_int(d), ptr(xsq), stream_ptr()), "norms")
    np.testing.assert_allclose(xsq.cpu().numpy(), (X * X).sum(1), rtol=1e-14)
    cand = torch.tensor([5, 900, 10006, 0, 5], device=DEV, dtype=torch.int64)
    closest = torch.from_numpy(rng.uniform(0, 30, N)).to(DEV)
    newd = torch.empty((T, N), device=DEV, dtype=torch.float64)
    pots = torch.empty(T, device=DEV, dtype=torch.float64)
    check(L.clava_kmpp_candidates_f64(ptr(Xd), ctypes.c_longlong(d), ctypes.c_longlong(N), c_int(d), ptr(xsq), ptr(closest), ptr(cand),
                                      c_int(T), ptr(ws), ptr(newd), ptr(pots), stream_ptr()), "candidates")
    C = X[cand.cpu().numpy()]
    dist = np.maximum(-2 * C @ X.T + (C * C).sum(1)[:, None] + (X * X).sum(1)[None, :], 0)
    ref = np.minimum(closest.cpu().numpy()[None, :], dist)
    np.testing.assert_allclose(newd.cpu().numpy(), ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(pots.cpu().numpy(), ref.sum(1), rtol=1e-13)
    assert float(newd[0, 5]) == 0.0 or float(newd[0, 5]) < 1e-12          # a candidate is at distance 0 from itself
    # cumulative-distribution search, incl. both ends
    r = torch.tensor([0.0, 0.25, 0.5, 0.999999, 1.0], device=DEV, dtype=torch.float64)
    ids = torch.empty(T, device=DEV, dtype=torch.int64)
    total = torch.empty(1, device=DEV, dtype=torch.float64)
    check(L.clava_kmpp_pick_f64(ptr(closest), ctypes.c_longlong(N), ptr(r), c_int(T), ptr(ws), ptr(ids), ptr(total), stream_ptr()), "pick")
    cl = closest.cpu().numpy()
    cum = np.cumsum(cl)
    want = np.clip(np.searchsorted(cum, r.cpu().numpy() * cum[-1]), None, N - 1)
    got = ids.cpu().numpy()
    assert abs(float(total) - cum[-1]) < 1e-9 * cum[-1]
    assert np.all(np.abs(got - want) <= 1) and np.array_equal(got[[0]], want[[0]])    # a rounding-level tie may move one slot


def test_hip_gaussian_mixture_fit_is_device_seeded_and_matches_sklearn():
    """End to end with the device k-means++ in front of the device EM loop: scikit-learn's whole `fit` (G11 recording)."""
    from unittest import mock
    from clavaddpm_amd import clustering
    g = golden("g11_gmm_fit.npz")
    X = g["X"].numpy()
    with mock.patch("sklearn.mixture._base.kmeans_plusplus", side_effect=AssertionError("host k-means++ was used")):
        gm = clustering.HipGaussianMixture(n_components=9, covariance_type="diag", init_params="k-means++", tol=1e-4,
                                           random_state=3).fit(X)
    assert gm.n_iter_ == int(g["n_iter"])
    np.testing.assert_allclose(gm.means_, g["means"].numpy(), rtol=1e-8, atol=1e-10)
    assert np.array_equal(gm.predict(X), g["labels"].numpy())


# ---------------------------------------------------------------------------------------------- majority vote (8f-2)
def _prob_rows(prob):
    return np.array([[lab, ln, p] for lab, d in prob.items() for ln, p in d.items()], dtype=np.float64)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g13_group_vote_matches_the_reference_statements(tag):
    from conftest import golden, sub
    from clavaddpm_amd.clustering import group_vote
    g = sub(golden("g13_group_vote.npz"), tag + "_")
    labels, agree, assign, prob = group_vote(g["labels"].numpy(), g["lengths"].numpy())
    assert np.array_equal(labels, g["group_labels"].numpy())                 # winners exact, lowest label on ties
    assert np.array_equal(agree, g["agree"].numpy())                         # same integer ratio in float64
    assert np.array_equal(assign, g["assignment"].numpy())
    assert np.array_equal(_prob_rows(prob), g["prob"].numpy())               # values and dict order
    assert float(np.mean(agree)) == float(g["mean_agree"])


def test_group_vote_large_table_against_the_oracle_and_device_labels():
    from clavaddpm_amd.clustering import group_vote
    from oracle import vote
    rng = np.random.RandomState(3)
    G, k = 200_000, 50
    lengths = rng.poisson(2.5, G) + 1
    lengths[::5003] = 5000                                                   # long groups: many rows per wavefront lane
    N = int(lengths.sum())
    labels = rng.randint(0, k, N).astype(np.int32)
    ref = vote.group_vote(labels, lengths)
    got = group_vote(torch.from_numpy(labels).cuda(), lengths, n_labels=k)   # labels as they leave the device E-step
    for a, b in zip(got[:3], ref[:3]):
        assert np.array_equal(a, b)
    assert got[3] == ref[3] and list(got[3]) == list(ref[3])
    # properties at full size: every winner's count is the maximum of its group, the assignment is constant per group
    assert int(got[2].shape[0]) == N and float(got[1].max()) <= 1.0 and float(got[1].min()) > 0.0


def test_group_vote_rejects_empty_groups_and_bad_labels():
    from clavaddpm_amd.clustering import group_vote
    with pytest.raises(ValueError):
        group_vote(np.array([0, 1, 1]), np.array([2, 0, 1]))
    with pytest.raises(ValueError):
        group_vote(np.array([0, 1, 7]), np.array([2, 1]), n_labels=3)
    with pytest.raises(ValueError):
        group_vote(np.array([0, 1]), np.array([2, 1]))
