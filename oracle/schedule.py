"""Noise-schedule tables (oracle; test infrastructure only).

Restates `tab_ddpm/gaussian_multinomial_diffsuion.py:18-60` (beta schedules) and
`:105-156` (the tables the constructor derives), computed in float64 and rounded
to float32 exactly where the reference rounds.
"""
import math

import numpy as np


def betas(name, T):
    """`get_named_beta_schedule` / `betas_for_alpha_bar` (`:18-60`)."""
    if name == "linear":
        k = 1000 / T
        return np.linspace(k * 0.0001, k * 0.02, T, dtype=np.float64)
    if name == "cosine":
        def abar(u):
            return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - abar((i + 1) / T) / abar(i / T), 0.999) for i in range(T)])
    raise NotImplementedError(f"unknown beta schedule: {name}")


def _log1m_exp(a):
    """`log_1_min_a` (`tab_ddpm/utils.py:100-101`)."""
    return np.log(1 - np.exp(a) + 1e-40)


def tables(name, T):
    """All schedule tables of `GaussianMultinomialDiffusion.__init__` (`:105-156`).

    Returns a dict of numpy arrays.  Keys ending in `_f64` are the ones the
    reference leaves in float64 (`posterior_variance`, `:125-127`); everything
    else is float32 like the registered buffers (`:145-156`).
    """
    a = 1.0 - betas(name, T)              # alphas, float64 (:105-106)
    b = 1.0 - a                           # betas as the ctor recomputes them (:107)
    la = np.log(a)
    lca = np.cumsum(la)
    ac = np.cumprod(a, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    ac_next = np.append(ac[1:], 0.0)
    pv = b * (1.0 - ac_prev) / (1.0 - ac)  # posterior_variance, float64 (:125-127)
    f32 = lambda v: np.asarray(v, dtype=np.float64).astype(np.float32)
    out = {
        "alphas": f32(a),
        "log_alpha": f32(la),
        "log_1_min_alpha": f32(_log1m_exp(la)),
        "log_cumprod_alpha": f32(lca),
        "log_1_min_cumprod_alpha": f32(_log1m_exp(lca)),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(ac_prev),
        "alphas_cumprod_next": f32(ac_next),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)),
        "posterior_variance_f64": pv,
        "posterior_log_variance_clipped": f32(np.log(np.append(pv[1], pv[1:]))),
        "posterior_mean_coef1": f32(b * np.sqrt(ac_prev) / (1.0 - ac)),
        "posterior_mean_coef2": f32((1.0 - ac_prev) * np.sqrt(a) / (1.0 - ac)),
    }
    # "fixed-large" model variance rebuilt every step at `:209-211`:
    # [posterior_variance[1] (f64), (1 - alphas_f32)[1:] (f32, promoted)], then log in f64.
    mv = np.concatenate([[pv[1]], (np.float32(1.0) - out["alphas"])[1:].astype(np.float64)])
    out["model_variance_f64"] = mv
    out["model_log_variance_f64"] = np.log(mv)
    return out
