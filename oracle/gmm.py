"""Diagonal-covariance GMM E-step (oracle; test infrastructure only).

The arithmetic is not in the reference: `pipeline_modules.py:268-277` calls
`sklearn.mixture.GaussianMixture(covariance_type='diag')` (third-party
scikit-learn, pinned upstream at 1.4.1.post1, `requirements.txt:199`; 1.7.2 in
this image).  This restates its published algorithm
(`sklearn/mixture/_gaussian_mixture.py` `_estimate_log_gaussian_prob` 'diag'
branch and `_compute_log_det_cholesky`; `_base.py` `_estimate_weighted_log_prob`,
`_estimate_log_prob_resp`, `predict`) in numpy float64.  Pinned only by vectors
captured from the installed scikit-learn 1.7.2 (tests/golden/g9_gmm.npz); the
reference holds no fixture for it.
"""
import numpy as np


def estep(X, weights, means, precisions_chol):
    """X [N,d], weights [k], means [k,d], precisions_chol [k,d] (all float64).

    Returns (log_prob_norm [N], log_resp [N,k], labels [N] int64).
    """
    d = X.shape[1]
    prec = precisions_chol ** 2
    log_det = np.sum(np.log(precisions_chol), axis=1)
    lp = np.sum(means ** 2 * prec, 1) - 2.0 * np.dot(X, (means * prec).T) + np.dot(X ** 2, prec.T)
    wlp = (-0.5 * (d * np.log(2 * np.pi) + lp) + log_det) + np.log(weights)
    m = wlp.max(axis=1, keepdims=True)
    lpn = (m + np.log(np.exp(wlp - m).sum(axis=1, keepdims=True)))[:, 0]   # scipy logsumexp
    return lpn, wlp - lpn[:, None], wlp.argmax(axis=1)


def mstep(X, log_resp, reg_covar=1e-6):
    """scikit-learn's `GaussianMixture._m_step`, 'diag' branch (`sklearn/mixture/_gaussian_mixture.py`:
    `_estimate_gaussian_parameters`, `_estimate_gaussian_covariances_diag`, `_compute_precision_cholesky`;
    scikit-learn 1.7.2 as installed - the release the reference pins, 1.4.1.post1, computes the same covariance as
    avg_X2 - 2 avg_X_means + avg_means2, equal in exact arithmetic).

    Returns (weights [k], means [k,d], covariances [k,d], precisions_chol [k,d])."""
    resp = np.exp(log_resp)
    nk = resp.sum(axis=0) + 10 * np.finfo(resp.dtype).eps
    means = np.dot(resp.T, X) / nk[:, None]
    cov = np.dot(resp.T, X * X) / nk[:, None] - means ** 2 + reg_covar
    weights = nk / nk.sum()
    if np.any(cov <= 0.0):
        raise ValueError("ill-defined empirical covariance")
    return weights, means, cov, 1.0 / np.sqrt(cov)


def mstep_totals(X, log_resp, log_prob_norm=None):
    """The sums `_estimate_gaussian_parameters` takes over rows, for ONE shard of the rows: (sum resp [k], sum resp*x
    [k,d], sum resp*x^2 [k,d], sum log_prob_norm).  Adding the tuples of all shards and calling `mstep_from_totals` is
    the row-sharded M-step."""
    resp = np.exp(log_resp)
    lp = 0.0 if log_prob_norm is None else float(np.sum(log_prob_norm))
    return resp.sum(axis=0), np.dot(resp.T, X), np.dot(resp.T, X * X), lp


def mstep_from_totals(nk_sum, sx, sxx, reg_covar=1e-6):
    """`mstep` from the summed totals of all shards; same returns."""
    nk = nk_sum + 10 * np.finfo(np.float64).eps
    means = sx / nk[:, None]
    cov = sxx / nk[:, None] - means ** 2 + reg_covar
    if np.any(cov <= 0.0):
        raise ValueError("ill-defined empirical covariance")
    return nk / nk.sum(), means, cov, 1.0 / np.sqrt(cov)


def em(X, weights, means, precisions_chol, tol=1e-4, max_iter=100, reg_covar=1e-6):
    """The EM loop of `BaseMixture.fit_predict` (`sklearn/mixture/_base.py`) from given initial parameters, n_init = 1:
    E-step, M-step, lower bound = mean log_prob_norm, stop when its change is below `tol`; then the final E-step.
    Returns (weights, means, covariances, precisions_chol, n_iter, lower_bound, converged, labels)."""
    lower = -np.inf
    cov = None
    converged = False
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        prev = lower
        lpn, log_resp, _ = estep(X, weights, means, precisions_chol)
        weights, means, cov, precisions_chol = mstep(X, log_resp, reg_covar)
        lower = lpn.mean()
        if abs(lower - prev) < tol:
            converged = True
            break
    _, log_resp, _ = estep(X, weights, means, precisions_chol)
    return weights, means, cov, precisions_chol, n_iter, lower, converged, log_resp.argmax(axis=1)
