"""Denoiser network (oracle; test infrastructure only).

Restates `tab_ddpm/modules.py:20-38` (`timestep_embedding`) and `:425-460`
(`MLPDiffusion.forward` with `is_y_cond='none'`, the only mode ClavaDDPM uses,
`pipeline_modules.py:52`) as a function of a parameter dict keyed like the
reference `state_dict` (`proj`, `time_embed.{0,2}`, `mlp.blocks.{i}.linear`,
`mlp.head`).
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim, max_period=10000):
    """`tab_ddpm/modules.py:20-38`."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    e = torch.cat([torch.cos(a), torch.sin(a)], dim=-1)
    if dim % 2:
        e = torch.cat([e, torch.zeros_like(e[:, :1])], dim=-1)
    return e


def time_mlp(p, prefix, t, dim_t):
    """`time_embed` = Linear -> SiLU -> Linear (`modules.py:445-449`, `pipeline_utils.py:597-601`)."""
    w0 = p[prefix + "0.weight"]
    h = F.linear(timestep_embedding(t, dim_t).to(w0.dtype), w0, p[prefix + "0.bias"])   # (float64 copies: conditioning checks)
    return F.linear(F.silu(h), p[prefix + "2.weight"], p[prefix + "2.bias"])


def n_blocks(p):
    n = 0
    while f"mlp.blocks.{n}.linear.weight" in p:
        n += 1
    return n


def mlp_diffusion(p, x, t, dim_t=128, return_hidden=False):
    """`MLPDiffusion.forward` (`modules.py:451-460`) + `MLP.forward` (`:238-243`), dropout 0."""
    h = F.linear(x.float(), p["proj.weight"], p["proj.bias"]) + time_mlp(p, "time_embed.", t, dim_t)
    hidden = [h]
    for i in range(n_blocks(p)):
        h = F.relu(F.linear(h, p[f"mlp.blocks.{i}.linear.weight"], p[f"mlp.blocks.{i}.linear.bias"]))
        hidden.append(h)
    out = F.linear(h, p["mlp.head.weight"], p["mlp.head.bias"])
    return (out, hidden) if return_hidden else out


def params_of(module):
    """Detach a module's state_dict into plain float32/long CPU tensors."""
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
