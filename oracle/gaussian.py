"""Gaussian half of the diffusion step (oracle; test infrastructure only).

Restates `tab_ddpm/gaussian_multinomial_diffsuion.py:172-237, 302-307, 315-328,
354-383` with torch CPU tensors so that dtype promotion (the float64 chain of
SURVEY section 0 item 7) is the reference's own.
"""
import torch


def _tab(tb, key, t, like):
    """`extract` (`tab_ddpm/utils.py:111-117`): per-row table lookup broadcast over columns."""
    v = torch.as_tensor(tb[key])
    return v[t].reshape(-1, *([1] * (like.dim() - 1))).expand(like.shape)


def q_sample(tb, x0, t, noise):
    """`gaussian_q_sample` (`:172-180`)."""
    return _tab(tb, "sqrt_alphas_cumprod", t, x0) * x0 + _tab(tb, "sqrt_one_minus_alphas_cumprod", t, x0) * noise


def xstart_from_eps(tb, x_t, t, eps):
    """`_predict_xstart_from_eps` (`:302-307`)."""
    return _tab(tb, "sqrt_recip_alphas_cumprod", t, x_t) * x_t - _tab(tb, "sqrt_recipm1_alphas_cumprod", t, x_t) * eps


def posterior_mean(tb, x0, x_t, t):
    """mean of `gaussian_q_posterior_mean_variance` (`:182-198`)."""
    return _tab(tb, "posterior_mean_coef1", t, x_t) * x0 + _tab(tb, "posterior_mean_coef2", t, x_t) * x_t


def p_mean_variance(tb, model_out, x, t):
    """`gaussian_p_mean_variance` (`:200-237`), eps parametrisation, fixed-large variance."""
    var = _tab(tb, "model_variance_f64", t, x)
    logvar = _tab(tb, "model_log_variance_f64", t, x)
    x0 = xstart_from_eps(tb, x, t, model_out)
    return {"mean": posterior_mean(tb, x0, x, t), "variance": var, "log_variance": logvar, "pred_xstart": x0}


def p_sample(tb, model_out, x, t, noise, grad=None):
    """`gaussian_p_sample` (`:354-383`) with the noise injected and the guidance
    gradient (`condition_mean`, `:315-328`) already evaluated by the caller."""
    out = p_mean_variance(tb, model_out, x, t)
    mean = out["mean"]
    if grad is not None:
        mean = mean.float() + out["variance"] * grad.float()
    keep = (t != 0).float().reshape(-1, *([1] * (x.dim() - 1)))
    sample = mean + keep * torch.exp(0.5 * out["log_variance"]) * noise
    return {"sample": sample, "pred_xstart": out["pred_xstart"], "mean": mean}
