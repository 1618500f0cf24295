"""Reverse-diffusion sampling loops (oracle; test infrastructure only).

Restates `sample` (`tab_ddpm/gaussian_multinomial_diffsuion.py:1026-1070`),
`conditional_sample` (`:987-1022`) and the batching / NaN policy of `sample_all`
(`:1072-1099`) with every random draw injected.

Noise dict (all CPU tensors, in the reference's draw order):
  z0      [b, n_num]  f32   initial normal draw                      (:1029 / :990)
  u_init  [b, sumK]   f32   uniforms of the initial categorical draw (:1033-1035)
  noise   [T, b, n_num] f64 per executed step s (t = T-1-s), drawn even at t=0 (:372)
  u       [T, b, sumK] f32  per-step uniforms, feature blocks side by side (:508-513)
"""
import torch

from . import gaussian, multinomial as mn
from .network import mlp_diffusion


def reverse_chain(tb, T, net, n_num, K, z0, noise, idx0=None, u=None, grad_fn=None, steps=None, trace=None, t_start=None):
    """The loop body shared by `sample` and `conditional_sample`.

    `net` maps (x_in float32 [b, D_in], t long [b]) -> model_out.
    `grad_fn(x, t)` is the already-bound `cond_fn` (guidance), or None.
    `t_start` (default T-1, the reference's only entry point) resumes the same loop body at a later timestep: z0 / idx0
    are then the state at t_start (a float64 z0 = a state past the float32 first step); the loop body is unchanged.
    Returns (x_num float64 [b, n_num], idx long [b, m]).
    """
    b = z0.shape[0]
    has_cat = len(K) > 0 and int(K[0]) != 0
    x = z0
    idx = idx0
    log_z = mn.index_to_log_onehot(idx, K) if has_cat else torch.zeros((b, 0))
    t0 = T - 1 if t_start is None else t_start
    n_steps = t0 + 1 if steps is None else steps
    for s in range(n_steps):
        t = torch.full((b,), t0 - s, dtype=torch.long)
        out = net(torch.cat([x, log_z], dim=1).float(), t)
        g = grad_fn(x, t) if grad_fn is not None else None
        x = gaussian.p_sample(tb, out[:, :n_num], x, t, noise[s].to(x.dtype) if s == 0 else noise[s], g)["sample"]
        if has_cat:
            idx, _ = mn.p_sample(tb, out[:, n_num:], idx, t, u[s], K)
            log_z = mn.index_to_log_onehot(idx, K)
        if trace is not None:
            trace.append((x.clone(), None if idx is None else idx.clone()))
    return x, idx


def sample(tb, T, p, n_num, K, nz, dim_t=128, steps=None, t_start=None, idx_start=None):
    """`sample` (`:1026-1070`): unconditional chain; returns [b, n_num + m] float64."""
    has_cat = len(K) > 0 and int(K[0]) != 0
    idx0 = None
    if has_cat:
        idx0 = mn.sample_categorical(torch.zeros_like(nz["u_init"]), nz["u_init"], K) if idx_start is None else idx_start
    net = lambda xin, t: mlp_diffusion(p, xin, t, dim_t)
    x, idx = reverse_chain(tb, T, net, n_num, K, nz["z0"], nz["noise"], idx0, nz.get("u"), None, steps, t_start=t_start)
    return torch.cat([x, idx.to(x.dtype)], dim=1) if has_cat else x


def conditional_sample(tb, T, p, n_num, grad_fn, nz, dim_t=128, steps=None, t_start=None):
    """`conditional_sample` (`:987-1022`): Gaussian-only."""
    net = lambda xin, t: mlp_diffusion(p, xin, t, dim_t)
    x, _ = reverse_chain(tb, T, net, n_num, [0], nz["z0"], nz["noise"], None, None, grad_fn, steps, t_start=t_start)
    return x


def drop_nan_rows(x):
    """NaN policy of `sample_all` (`:1086-1093`): returns (kept rows, any_dropped)."""
    bad = torch.any(x.isnan(), dim=1)
    return x[~bad], bool(bad.any())
