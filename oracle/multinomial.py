"""Multinomial (categorical) half of the diffusion step (oracle; test infrastructure only).

Restates `tab_ddpm/gaussian_multinomial_diffsuion.py:387-472, 506-523, 544-572`
and the helpers in `tab_ddpm/utils.py:92-166`.  `K` is the list of category
sizes; one-hot blocks are laid side by side (width sum(K)).
"""
import numpy as np
import torch
import torch.nn.functional as F

LOG_TINY = float(np.log(np.float32(1e-30)))  # what log(clamp(onehot, 1e-30)) gives off the hot class


def offsets(K):
    return np.concatenate([[0], np.cumsum(np.asarray(K, dtype=np.int64))])


def log_k_expanded(K):
    """log of `num_classes_expanded` (`:87-89`, used at `:398,409`)."""
    k = np.concatenate([np.full(int(k), int(k)) for k in K])
    return torch.log(torch.from_numpy(k))  # int64 -> float32 log, as torch.log promotes


def index_to_log_onehot(idx, K):
    """`tab_ddpm/utils.py:127-134`."""
    hot = torch.cat([F.one_hot(idx[:, f], int(K[f])) for f in range(len(K))], dim=1)
    return torch.log(hot.float().clamp(min=1e-30))


def log_add_exp(a, b):
    """`tab_ddpm/utils.py:104-106`."""
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def _col(tb, key, t, like):
    return torch.as_tensor(tb[key])[t].reshape(-1, 1).expand(like.shape)


def q_pred(tb, log_x0, t, K):
    """`q_pred` (`:403-412`): cumulative forward marginal."""
    return log_add_exp(log_x0 + _col(tb, "log_cumprod_alpha", t, log_x0),
                       _col(tb, "log_1_min_cumprod_alpha", t, log_x0) - log_k_expanded(K))


def q_pred_one_timestep(tb, log_xt, t, K):
    """`q_pred_one_timestep` (`:391-401`)."""
    return log_add_exp(log_xt + _col(tb, "log_alpha", t, log_xt),
                       _col(tb, "log_1_min_alpha", t, log_xt) - log_k_expanded(K))


def predict_start(model_out, K):
    """`predict_start` (`:414-424`): per-feature log-softmax."""
    off = offsets(K)
    out = torch.empty_like(model_out)
    for f in range(len(K)):
        out[:, off[f]:off[f + 1]] = F.log_softmax(model_out[:, off[f]:off[f + 1]], dim=1)
    return out


def sliced_logsumexp(x, K):
    """`sliced_logsumexp` (`tab_ddpm/utils.py:146-166`): row-wide logcumsumexp,
    then log_sub_exp of the slice ends, repeated back over each slice."""
    off = torch.from_numpy(offsets(K))
    lse = torch.logcumsumexp(F.pad(x, [1, 0, 0, 0], value=-float("inf")), dim=-1)
    hi, lo = lse[:, off[1:]], lse[:, off[:-1]]
    m = torch.maximum(hi, lo)
    s = torch.log(torch.exp(hi - m) - torch.exp(lo - m)) + m
    return torch.repeat_interleave(s, off[1:] - off[:-1], dim=-1)


def q_posterior_unnormed(tb, log_x0, log_xt, t, K):
    tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)
    ev = q_pred(tb, log_x0, tm1, K)
    ev = torch.where((t == 0).reshape(-1, 1).expand(log_x0.shape), log_x0, ev.to(torch.float32))
    return ev + q_pred_one_timestep(tb, log_xt, t, K)


def q_posterior(tb, log_x0, log_xt, t, K, exact=False):
    """`q_posterior` (`:426-455`).  `exact=True` swaps the reference's cancellation-prone normaliser for a direct
    per-block float64 log-sum-exp (see `sliced_logsumexp_exact`): what the HIP kernels compute."""
    u = q_posterior_unnormed(tb, log_x0, log_xt, t, K)
    if exact:
        return u - sliced_logsumexp_exact(u, K).to(u.dtype)
    return u - sliced_logsumexp(u, K)


def sliced_logsumexp_exact(x, K):
    """Per-block log-sum-exp evaluated directly in float64 (no row-wide scan).  NOT what the reference computes:
    its `sliced_logsumexp` subtracts two row-wide float32 logcumsumexp values, which cancels catastrophically for
    blocks that sit behind a lot of earlier probability mass (errors up to ~4e-2 on the 1371-class fixture, and
    different on every platform).  Used only to check the HIP kernel's normaliser, which is a direct per-block
    reduction as well."""
    off = offsets(K)
    cols = [torch.logsumexp(x[:, off[f]:off[f + 1]].double(), dim=1, keepdim=True).expand(-1, int(K[f])) for f in range(len(K))]
    return torch.cat(cols, dim=1)


def p_pred_exact_norm(tb, model_out, log_xt, t, K):
    """`p_pred` with the exact normaliser (see `sliced_logsumexp_exact`); float64."""
    u = q_posterior_unnormed(tb, predict_start(model_out, K), log_xt, t, K)
    return u.double() - sliced_logsumexp_exact(u, K)


def p_pred(tb, model_out, log_xt, t, K, exact=False):
    """`p_pred` (`:457-466`), 'x0' parametrisation."""
    return q_posterior(tb, predict_start(model_out, K), log_xt, t, K, exact)


def sample_categorical(logits, uniforms, K):
    """`log_sample_categorical` (`:506-516`) with the uniforms injected; returns indices [B, m]."""
    off = offsets(K)
    cols = []
    for f in range(len(K)):
        u = uniforms[:, off[f]:off[f + 1]]
        g = -torch.log(-torch.log(u + 1e-30) + 1e-30)
        cols.append((g + logits[:, off[f]:off[f + 1]]).argmax(dim=1, keepdim=True))
    return torch.cat(cols, dim=1)


def p_sample(tb, model_out, idx_t, t, uniforms, K):
    """`p_sample` (`:468-472`) on index state: returns (new indices, log_model_prob)."""
    lp = p_pred(tb, model_out, index_to_log_onehot(idx_t, K), t, K)
    return sample_categorical(lp, uniforms, K), lp


def q_sample(tb, idx0, t, uniforms, K):
    """`q_sample` (`:518-523`)."""
    return sample_categorical(q_pred(tb, index_to_log_onehot(idx0, K), t, K), uniforms, K)


def multinomial_kl(lp1, lp2):
    """`multinomial_kl` (`:387-389`)."""
    return (lp1.exp() * (lp1 - lp2)).sum(dim=1)


def kl_prior(tb, log_x0, K, T):
    """`kl_prior` (`:544-553`)."""
    t = torch.full((log_x0.shape[0],), T - 1, dtype=torch.long)
    lq = q_pred(tb, log_x0, t, K)
    lh = -torch.log(torch.from_numpy(np.concatenate([np.full(int(k), int(k)) for k in K])) * torch.ones_like(lq))
    return multinomial_kl(lq, lh)


def compute_Lt(tb, model_out, log_x0, log_xt, t, K, exact=False):
    """`compute_Lt` (`:555-572`)."""
    true = q_posterior(tb, log_x0, log_xt, t, K, exact)
    model = p_pred(tb, model_out, log_xt, t, K, exact)
    kl = multinomial_kl(true, model)
    nll = -(log_x0.exp() * model).sum(dim=1)
    first = (t == 0).float()
    return first * nll + (1.0 - first) * kl


def multinomial_loss(tb, model_out, log_x0, log_xt, t, pt, K, T, exact=False):
    """`_multinomial_loss` 'vb_stochastic' (`:597-607`)."""
    return compute_Lt(tb, model_out, log_x0, log_xt, t, K, exact) / pt + kl_prior(tb, log_x0, K, T)


def onehot_to_index(ohe, K):
    """`ohe_to_categories` (`tab_ddpm/utils.py:92-98`)."""
    off = offsets(K)
    return torch.stack([ohe[:, off[f]:off[f + 1]].argmax(dim=1) for f in range(len(K))], dim=1)
