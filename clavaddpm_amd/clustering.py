"""GMM E-step and M-step of the parent-child clustering stage on the HIP library.

`gmm_estep` / `gmm_mstep` are the raw operators; `HipGaussianMixture` plugs them into scikit-learn's `GaussianMixture`
(`pipeline_modules.py:268-277` constructs one with covariance_type='diag', init_params='k-means++', tol=1e-4) by
overriding the E-step and M-step pieces (`_e_step`, `_m_step`, `_compute_lower_bound`, `_estimate_log_prob_resp`,
`predict`): during `fit` the data matrix, the responsibilities and the per-row log-likelihoods never leave the device;
one small packed buffer (the new parameters, the lower bound and a validity flag) comes back per EM iteration.
k-means++ initialisation stays scikit-learn's host code.
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


def gmm_estep_noalloc(X, weights, means, precisions_chol, log_prob_norm, log_resp=None, labels=None):
    """`clava_gmm_estep_noalloc_f64`: the E-step for callers that cannot allocate (graph capture, serving loops).  Every
    tensor, outputs included, is the caller's (float64 device, contiguous; labels int32); the library's own tables live in
    a static device array, so this call performs no device allocation at all - this wrapper allocates nothing either.
    Limit: k <= 64 and 2*k*d + 3*k <= 65728 doubles.  Calls that overlap on one GPU race on that array."""
    _lib.require_cuda(X, weights, means, precisions_chol, log_prob_norm, log_resp, labels)
    for t in (X, weights, means, precisions_chol, log_prob_norm, log_resp):
        if t is not None and (t.dtype != torch.float64 or not (t.dim() < 2 or t.stride(-1) == 1)):
            raise _lib.ClavaHipError("gmm_estep_noalloc: contiguous float64 tensors expected")
    for t in (weights, means, precisions_chol, log_prob_norm, log_resp, labels):
        if t is not None and not t.is_contiguous():
            raise _lib.ClavaHipError("gmm_estep_noalloc: contiguous tensors expected (this call makes no copies)")
    N, d = X.shape
    k = means.shape[0]
    if labels is not None and labels.dtype != torch.int32:
        raise _lib.ClavaHipError("gmm_estep_noalloc: labels must be int32")
    if (log_prob_norm.numel() != N or (log_resp is not None and tuple(log_resp.shape) != (N, k))
            or (labels is not None and labels.numel() != N) or tuple(precisions_chol.shape) != (k, d) or weights.numel() != k
            or means.shape[1] != d):
        raise _lib.ClavaHipError("gmm_estep_noalloc: output / parameter shapes do not match X [N,d] and means [k,d]")
    if N:
        check(lib().clava_gmm_estep_noalloc_f64(ptr(X), c_ll(X.stride(0)), c_ll(N), c_int(d), c_int(k), ptr(weights), ptr(means),
                                                ptr(precisions_chol), ptr(log_prob_norm), ptr(log_resp), ptr(labels), stream_ptr()),
              "clava_gmm_estep_noalloc_f64")
    return log_prob_norm, log_resp, labels


def gmm_mstep(X, log_resp, log_prob_norm=None, reg_covar=1e-6, workspace=None):
    """scikit-learn's diag M-step on the device.  X [N,d], log_resp [N,k], log_prob_norm [N] or None: float64 device
    tensors.  Returns one packed float64 device tensor [2 + k + 3*k*d] = (lower bound, bad-covariance flag, weights [k],
    means [k,d], covariances [k,d], precisions_chol [k,d]); `unpack_mstep` splits a (host or device) copy of it."""
    _lib.require_cuda(X, log_resp)
    if X.stride(1) != 1:
        X = X.contiguous()
    log_resp = log_resp.contiguous()
    N, d = X.shape
    k = log_resp.shape[1]
    dev = X.device
    if workspace is None:
        need = lib().clava_gmm_mstep_workspace_doubles
        need.restype = ctypes.c_longlong
        workspace = torch.empty(int(need(c_int(k), c_int(d))), device=dev, dtype=torch.float64)
    out = torch.empty(2 + k + 3 * k * d, device=dev, dtype=torch.float64)
    st, w, m, cv, pc = unpack_mstep(out, k, d)
    check(lib().clava_gmm_mstep_f64(ptr(X), c_ll(X.stride(0)), c_ll(N), c_int(d), c_int(k), ptr(log_resp), ptr(log_prob_norm),
                                    ctypes.c_double(reg_covar), ptr(workspace), ptr(w), ptr(m), ptr(cv), ptr(pc), ptr(st),
                                    stream_ptr()), "clava_gmm_mstep_f64")
    return out


def mstep_totals_size(k, d):
    f = lib().clava_gmm_mstep_totals_doubles
    f.restype = ctypes.c_longlong
    return int(f(c_int(k), c_int(d)))


def gmm_mstep_partial(X, log_resp, log_prob_norm=None, workspace=None, out=None):
    """This shard's M-step sums (`clava_gmm_mstep_partial_f64`): one float64 device tensor of `mstep_totals_size(k, d)`
    doubles, layout in include/clava_hip.h (`unpack_mstep_totals` gives views).  Sum the tensors of all row shards, then
    `gmm_mstep_finalize`."""
    _lib.require_cuda(X, log_resp)
    if X.stride(1) != 1:
        X = X.contiguous()
    log_resp = log_resp.contiguous()
    N, d = X.shape
    k = log_resp.shape[1]
    E = mstep_totals_size(k, d)
    if workspace is None:
        workspace = torch.empty(512 * E, device=X.device, dtype=torch.float64)
    if out is None:
        out = torch.empty(E, device=X.device, dtype=torch.float64)
    if N == 0:
        return out.zero_()
    check(lib().clava_gmm_mstep_partial_f64(ptr(X), c_ll(X.stride(0)), c_ll(N), c_int(d), c_int(k), ptr(log_resp), ptr(log_prob_norm),
                                            ptr(workspace), ptr(out), stream_ptr()), "clava_gmm_mstep_partial_f64")
    return out


def gmm_mstep_finalize(totals, n_rows, k, d, reg_covar=1e-6):
    """Parameters from summed totals (`clava_gmm_mstep_finalize_f64`); `n_rows` = rows of ALL shards.  Returns the packed
    tensor `gmm_mstep` returns."""
    _lib.require_cuda(totals)
    if totals.numel() != mstep_totals_size(k, d) or totals.dtype != torch.float64:
        raise _lib.ClavaHipError("gmm_mstep_finalize: totals must hold mstep_totals_size(k, d) float64 values")
    out = torch.empty(2 + k + 3 * k * d, device=totals.device, dtype=torch.float64)
    st, w, m, cv, pc = unpack_mstep(out, k, d)
    check(lib().clava_gmm_mstep_finalize_f64(ptr(totals), c_ll(int(n_rows)), c_int(d), c_int(k), ctypes.c_double(reg_covar), ptr(w), ptr(m),
                                             ptr(cv), ptr(pc), ptr(st), stream_ptr()), "clava_gmm_mstep_finalize_f64")
    return out


def unpack_mstep_totals(buf, k, d):
    """(sum resp*x [k,d], sum resp*x^2 [k,d], sum resp [k], sum log_prob_norm []) views of a totals buffer (host or device)."""
    KP, DP = (k + 15) // 16 * 16, (d + 31) // 32 * 32
    sx = buf[:KP * DP].reshape(KP, DP)[:k, :d]
    sxx = buf[KP * DP:2 * KP * DP].reshape(KP, DP)[:k, :d]
    return sx, sxx, buf[2 * KP * DP:2 * KP * DP + k], buf[2 * KP * DP + KP]


def pack_mstep_totals(sx, sxx, nk, lpn_sum):
    """Inverse of `unpack_mstep_totals` for host arrays (tests: oracle totals in the library's layout)."""
    k, d = sx.shape
    KP, DP = (k + 15) // 16 * 16, (d + 31) // 32 * 32
    buf = torch.zeros(2 * KP * DP + KP + 1, dtype=torch.float64)
    a, b, c, _ = unpack_mstep_totals(buf, k, d)
    a.copy_(torch.as_tensor(sx)); b.copy_(torch.as_tensor(sxx)); c.copy_(torch.as_tensor(nk))
    buf[2 * KP * DP + KP] = float(lpn_sum)
    return buf


def unpack_mstep(buf, k, d):
    """(stats [2], weights [k], means [k,d], covariances [k,d], precisions_chol [k,d]) views of a packed M-step result."""
    o = 2
    st = buf[:2]
    w = buf[o:o + k]; o += k
    m = buf[o:o + k * d].reshape(k, d); o += k * d
    cv = buf[o:o + k * d].reshape(k, d); o += k * d
    pc = buf[o:o + k * d].reshape(k, d)
    return st, w, m, cv, pc


def kmeans_plusplus(X, n_clusters, random_state):
    """Indices of the k-means++ seeds of scikit-learn's `kmeans_plusplus(X, n_clusters, random_state=...)`
    (`sklearn/cluster/_kmeans.py:_kmeans_plusplus`, unit sample weights) for a float64 device tensor X [N, d].
    `random_state` is a `numpy.random.RandomState`; it is consumed exactly as scikit-learn consumes it (one `choice`,
    then `uniform(size=2+int(log k))` per further centre), so a seeded run selects the same rows.  Distances, minima,
    potentials, the cumulative distribution and the searches run on the device; one small readback per centre."""
    _lib.require_cuda(X)
    if X.dtype != torch.float64:
        raise _lib.ClavaHipError("kmeans_plusplus: float64 tensor expected")
    if X.stride(1) != 1:
        X = X.contiguous()
    N, d = X.shape
    if N < n_clusters:
        raise ValueError(f"n_samples={N} should be >= n_clusters={n_clusters}.")
    dev = X.device
    T = 2 + int(np.log(n_clusters))
    L = lib()
    L.clava_kmpp_workspace_doubles.restype = ctypes.c_longlong
    ws = torch.empty(int(L.clava_kmpp_workspace_doubles()), device=dev, dtype=torch.float64)
    xsq = torch.empty(N, device=dev, dtype=torch.float64)
    check(L.clava_kmpp_row_norms_f64(ptr(X), c_ll(X.stride(0)), c_ll(N), c_int(d), ptr(xsq), stream_ptr()), "clava_kmpp_row_norms_f64")
    bufs = [torch.empty((T, N), device=dev, dtype=torch.float64) for _ in range(2)]
    small = torch.empty(2 * T + 1, device=dev, dtype=torch.float64)          # potentials [T], ids as int64 bits [T], total
    pots, ids, total = small[:T], small[T:2 * T].view(torch.int64), small[2 * T:]
    sw = np.ones(N, dtype=np.float64)
    center_id = int(random_state.choice(N, p=sw / sw.sum()))
    indices = np.full(n_clusters, -1, dtype=int)
    indices[0] = center_id
    ids[0] = center_id
    check(L.clava_kmpp_candidates_f64(ptr(X), c_ll(X.stride(0)), c_ll(N), c_int(d), ptr(xsq), None, ptr(ids), c_int(1), ptr(ws),
                                      ptr(bufs[0]), ptr(pots), stream_ptr()), "clava_kmpp_candidates_f64")
    closest, cur = bufs[0][0], 0
    for c in range(1, n_clusters):
        rand = torch.from_numpy(random_state.uniform(size=T)).to(dev)
        check(L.clava_kmpp_pick_f64(ptr(closest), c_ll(N), ptr(rand), c_int(T), ptr(ws), ptr(ids), ptr(total), stream_ptr()),
              "clava_kmpp_pick_f64")
        nxt = bufs[1 - cur]
        check(L.clava_kmpp_candidates_f64(ptr(X), c_ll(X.stride(0)), c_ll(N), c_int(d), ptr(xsq), ptr(closest), ptr(ids), c_int(T),
                                          ptr(ws), ptr(nxt), ptr(pots), stream_ptr()), "clava_kmpp_candidates_f64")
        host = small.cpu()                                                   # the one synchronisation per centre
        best = int(np.argmin(host[:T].numpy()))
        indices[c] = int(host[T:2 * T].view(torch.int64)[best])
        closest, cur = nxt[best], 1 - cur
    return indices


class _DeviceResponsibilities:
    """What `_e_step` hands to `_m_step` during `fit`: device-resident log_resp / log_prob_norm / labels.  `argmax` serves
    the last line of scikit-learn's `fit_predict` (`log_resp.argmax(axis=1)`)."""

    def __init__(self, log_resp, log_prob_norm, labels):
        self.log_resp, self.log_prob_norm, self.labels = log_resp, log_prob_norm, labels

    def argmax(self, axis=1):
        return self.labels.cpu().numpy().astype(np.int64)


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
    _in_fit = False

    def fit_predict(self, X, y=None):
        """scikit-learn's EM driver (`fit` calls it too).  The device copy of the data matrix lives exactly as long as
        this call: inside it scikit-learn passes one validated array to `_initialize_parameters`, `_e_step` and `_m_step`,
        so one upload serves all EM iterations; nothing is cached across calls (an array mutated in place between `fit`
        and `predict` is uploaded afresh)."""
        self._in_fit = True
        try:
            return super().fit_predict(X, y)
        finally:
            self._in_fit = False
            self._x_cache = (None, None)

    def _device_x(self, X):
        key, dev_x = self._x_cache
        if key is X and self._in_fit:
            return dev_x
        dev_x = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(self.device)
        if self._in_fit:
            self._x_cache = (X, dev_x)
        return dev_x

    def _params_dev(self):
        if self.covariance_type != "diag":
            raise NotImplementedError("HipGaussianMixture implements covariance_type='diag' (what ClavaDDPM uses)")
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)
        return f(self.weights_), f(self.means_), f(self.precisions_cholesky_)

    def _initialize_parameters(self, X, random_state):
        """`BaseMixture._initialize_parameters` for init_params='k-means++' with no user-supplied initial parameters:
        the seeds come from the device k-means++, and the one-hot responsibilities scikit-learn would build (an N x k
        matrix whose M-step reduces to reading k rows) are replaced by that read: nk = 1 + 10 eps, means = X[seed] / nk,
        covariances = X[seed]^2 / nk - means^2 + reg_covar, weights = nk / N (`GaussianMixture._initialize`)."""
        if (self.init_params != "k-means++" or self.covariance_type != "diag" or self.weights_init is not None
                or self.means_init is not None or self.precisions_init is not None):
            return super()._initialize_parameters(X, random_state)
        idx = kmeans_plusplus(self._device_x(X), self.n_components, random_state)
        rows = np.asarray(X, dtype=np.float64)[idx]
        nk = np.ones(self.n_components) + 10 * np.finfo(np.float64).eps
        means = rows / nk[:, None]
        cov = (rows * rows) / nk[:, None] - means ** 2 + self.reg_covar
        self.weights_ = nk / X.shape[0]
        self.means_ = means
        self.covariances_ = cov
        self.precisions_cholesky_ = 1.0 / np.sqrt(cov)

    # ---- EM loop of BaseMixture.fit_predict: everything of size N stays on the device
    def _fresh_dev_params(self):
        """Device copies of the parameters: those of the last device M-step if the numpy attributes are still the arrays
        it produced, else an upload."""
        c = getattr(self, "_dev_params", None)
        if c is not None and c[0] is self.weights_ and c[1] is self.means_ and c[2] is self.precisions_cholesky_:
            return c[3], c[4], c[5]
        return self._params_dev()

    def _e_step(self, X):
        w, m, pc = self._fresh_dev_params()
        lpn, resp, labels = gmm_estep(self._device_x(X), w, m, pc, want_resp=True, want_labels=True)
        r = _DeviceResponsibilities(resp, lpn, labels)
        return r, r

    def _m_step(self, X, log_resp):
        if not isinstance(log_resp, _DeviceResponsibilities):       # a caller outside fit(): scikit-learn's host M-step
            self._lower = None
            return super()._m_step(X, log_resp)
        Xd = self._device_x(X)
        k, d = log_resp.log_resp.shape[1], Xd.shape[1]
        ws = getattr(self, "_mstep_ws", None)
        if ws is None or ws[0] != (k, d, str(Xd.device)):
            need = lib().clava_gmm_mstep_workspace_doubles
            need.restype = ctypes.c_longlong
            ws = ((k, d, str(Xd.device)), torch.empty(int(need(c_int(k), c_int(d))), device=Xd.device, dtype=torch.float64))
            self._mstep_ws = ws
        packed = gmm_mstep(Xd, log_resp.log_resp, log_resp.log_prob_norm, self.reg_covar, workspace=ws[1])
        host = packed.cpu().numpy()                                  # the one synchronisation of an EM iteration
        st, w, m, cv, pc = unpack_mstep(host, k, d)
        if st[1] != 0.0:
            raise ValueError("Fitting the mixture model failed because some components have ill-defined empirical "
                             "covariance (for instance caused by singleton or collapsed samples). Try to decrease the "
                             "number of components, increase reg_covar, or scale the input data.")
        self.weights_, self.means_, self.covariances_, self.precisions_cholesky_ = w.copy(), m.copy(), cv.copy(), pc.copy()
        _, dw, dm, _, dpc = unpack_mstep(packed, k, d)
        self._dev_params = (self.weights_, self.means_, self.precisions_cholesky_, dw, dm, dpc)
        self._lower = float(st[0])

    def _compute_lower_bound(self, log_resp, log_prob_norm):
        if isinstance(log_prob_norm, _DeviceResponsibilities):
            return self._lower
        return super()._compute_lower_bound(log_resp, log_prob_norm)

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


def group_vote(cluster_labels, child_group_lengths, n_labels=None):
    """Majority vote per parent group (`pipeline_modules.py:300-344`) on the device.

    cluster_labels: the child rows' labels in `sorted_child_data` order (array-like of ints, or an int32 device tensor
    straight from `gmm_estep` / `HipGaussianMixture.predict`); child_group_lengths: rows per parent group (all >= 1).
    Returns (group_cluster_labels int64 [G], agree_rates float64 [G], group_assignment int64 [N, 1],
    group_lengths_prob_dicts {label: {length: probability}}) - the four values the reference derives there."""
    lengths = np.asarray(child_group_lengths, dtype=np.int64).reshape(-1)
    G = len(lengths)
    if G == 0 or int(lengths.min()) < 1:
        # np.bincount of an empty group is empty and np.max of it raises in the reference (:306)
        raise ValueError("group_vote: every group needs at least one row (zero-size array to reduction operation maximum)")
    offsets = np.zeros(G + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    N = int(offsets[-1])
    if torch.is_tensor(cluster_labels) and cluster_labels.is_cuda:
        lab = cluster_labels.reshape(-1).to(torch.int32).contiguous()
    else:
        lab = torch.from_numpy(np.ascontiguousarray(np.asarray(cluster_labels).reshape(-1).astype(np.int32))).cuda()
    if lab.numel() != N:
        raise ValueError(f"group_vote: {lab.numel()} labels for groups of {N} rows")
    dev = lab.device
    if n_labels is None:
        n_labels = int(lab.max().item()) + 1
    max_len = int(lengths.max())
    off_d = torch.from_numpy(offsets).to(dev)
    g_lab = torch.empty(G, device=dev, dtype=torch.int32)
    g_cnt = torch.empty(G, device=dev, dtype=torch.int32)
    assign = torch.empty(N, device=dev, dtype=torch.int32)
    # the (label, length) table is dense: fall back to counting the pairs on the host when it would be huge
    dense = n_labels * (max_len + 1) <= (1 << 24)
    hist = torch.zeros((n_labels, max_len + 1), device=dev, dtype=torch.int64) if dense else None
    bad = torch.zeros(1, device=dev, dtype=torch.int32)
    check(lib().clava_group_vote_i32(ptr(lab), ptr(off_d), c_ll(G), c_int(n_labels), ptr(g_lab), ptr(g_cnt), ptr(assign), ptr(hist),
                                     c_int(max_len), ptr(bad), stream_ptr()), "clava_group_vote_i32")
    if int(bad.item()) != 0:
        raise ValueError("group_vote: a cluster label lies outside [0, n_labels)")
    labels_h = g_lab.cpu().numpy().astype(np.int64)
    agree = g_cnt.cpu().numpy().astype(np.int64) / lengths                  # most_common_label_count / group_length (:311)
    prob = {}
    if dense:
        hh = hist.cpu().numpy()
        first_seen = {}
        for i, l in enumerate(labels_h.tolist()):                            # dict order of the reference: first appearance
            if l not in first_seen:
                first_seen[l] = i
        for l in first_seen:
            row = hh[l]
            total = int(row.sum())
            seen = {}
            for ln in lengths[labels_h == l].tolist():
                if ln not in seen:
                    seen[ln] = int(row[ln]) / total
            prob[l] = seen
    else:
        from collections import defaultdict
        freq = {}
        for l, ln in zip(labels_h.tolist(), lengths.tolist()):
            freq.setdefault(l, defaultdict(int))[ln] += 1
        for l, fd in freq.items():
            tot = sum(fd.values())
            prob[l] = {k: v / tot for k, v in fd.items()}
    return labels_h, agree, assign.cpu().numpy().astype(np.int64).reshape(-1, 1), prob
