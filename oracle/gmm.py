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
