"""GMM E-step of the parent-child clustering stage on the HIP library.

`gmm_estep` is the raw operator; `HipGaussianMixture` plugs it into scikit-learn's `GaussianMixture`
(`pipeline_modules.py:268-277` constructs one with covariance_type='diag', init_params='k-means++', tol=1e-4) by
overriding only the E-step pieces (`_estimate_log_prob_resp`, `predict`): k-means++ initialisation and the M-step stay
scikit-learn's host code (DESIGN.md "next").
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_int, check, lib, ptr, stream_ptr

c_ll = ctypes.c_longlong


def gmm_estep(X, weights, means, precisions_chol, want_resp=True, want_labels=True):
    """X [N,d], weights [k], means [k,d], precisions_chol [k,d]: float64 device tensors.
    Returns (log_prob_norm [N], log_resp [N,k] or None, labels int32 [N] or None)."""
    _lib.require_cuda(X, weights, means, precisions_chol)
    for t in (X, weights, means, precisions_chol):
        if t.dtype != torch.float64:
            raise _lib.ClavaHipError("gmm_estep: float64 tensors expected")
    if X.stride(1) != 1:
        X = X.contiguous()
    N, d = X.shape
    k = means.shape[0]
    dev = X.device
    need = lib().clava_gmm_workspace_doubles
    need.restype = ctypes.c_longlong
    ws = torch.empty(int(need(c_int(k), c_int(d))), device=dev, dtype=torch.float64)
    lpn = torch.empty(N, device=dev, dtype=torch.float64)
    resp = torch.empty((N, k), device=dev, dtype=torch.float64) if want_resp else None
    labels = torch.empty(N, device=dev, dtype=torch.int32) if want_labels else None
    if N:
        check(lib().clava_gmm_estep_f64(ptr(X), c_ll(X.stride(0)), c_ll(N), c_int(d), c_int(k), ptr(weights.contiguous()),
                                        ptr(means.contiguous()), ptr(precisions_chol.contiguous()), ptr(ws), ptr(lpn), ptr(resp),
                                        ptr(labels), stream_ptr()), "clava_gmm_estep_f64")
    return lpn, resp, labels


def _sk_gaussian_mixture():
    from sklearn.mixture import GaussianMixture
    return GaussianMixture


class HipGaussianMixture(_sk_gaussian_mixture()):
    """scikit-learn `GaussianMixture` (diag covariance) whose E-step and `predict` run on the MI355X.

    The data matrix is uploaded once per `fit` / `predict` call and stays resident in HBM across EM iterations."""

    def __init__(self, n_components=1, *, covariance_type="diag", tol=1e-3, reg_covar=1e-6, max_iter=100, n_init=1,
                 init_params="kmeans", weights_init=None, means_init=None, precisions_init=None, random_state=None,
                 warm_start=False, verbose=0, verbose_interval=10, device="cuda:0"):
        super().__init__(n_components=n_components, covariance_type=covariance_type, tol=tol, reg_covar=reg_covar,
                         max_iter=max_iter, n_init=n_init, init_params=init_params, weights_init=weights_init,
                         means_init=means_init, precisions_init=precisions_init, random_state=random_state,
                         warm_start=warm_start, verbose=verbose, verbose_interval=verbose_interval)
        self.device = device

    _x_cache = (None, None)

    def _device_x(self, X):
        key, dev_x = self._x_cache
        if key is not X:
            dev_x = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(self.device)
            self._x_cache = (X, dev_x)
        return dev_x

    def _params_dev(self):
        if self.covariance_type != "diag":
            raise NotImplementedError("HipGaussianMixture implements covariance_type='diag' (what ClavaDDPM uses)")
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)
        return f(self.weights_), f(self.means_), f(self.precisions_cholesky_)

    def _estimate_log_prob_resp(self, X):
        w, m, pc = self._params_dev()
        lpn, resp, _ = gmm_estep(self._device_x(X), w, m, pc, want_resp=True, want_labels=False)
        return lpn.cpu().numpy(), resp.cpu().numpy()

    def predict(self, X):
        from sklearn.utils.validation import check_is_fitted
        check_is_fitted(self)
        w, m, pc = self._params_dev()
        _, _, labels = gmm_estep(self._device_x(np.asarray(X)), w, m, pc, want_resp=False, want_labels=True)
        return labels.cpu().numpy().astype(np.int64)
