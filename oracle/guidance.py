"""Cluster-guidance classifier and its input gradient (oracle; test infrastructure only).

Restates `pipeline_utils.py:584-630` (`Classifier`, evaluated in eval mode as at
`:820`: BatchNorm uses running statistics, Dropout is the identity; the
`nn.Transformer` member is never called, `:628`) and the `cond_fn` closure at
`:853-864`.  Parameters are a dict keyed like the reference `state_dict`
(`proj`, `time_embed.{0,2}`, `model.{j}`).
"""
import torch
import torch.nn.functional as F

from .network import time_mlp

BN_EPS = 1e-5  # nn.BatchNorm1d default


def layer_plan(p):
    """Sequential indices of the Linear layers in `Classifier.model` (`:603-623`):
    Linear j is followed by an activation, BatchNorm1d j+2 and Dropout, except the last."""
    idx = sorted({int(k.split(".")[1]) for k in p if k.startswith("model.") and k.endswith(".running_mean")})
    lin = [j - 2 for j in idx]
    return lin, lin[-1] + 4


def classifier_logits(p, x, t, dim_t=128):
    """`Classifier.forward` (`:625-630`) in eval mode."""
    h = F.linear(x, p["proj.weight"], p["proj.bias"]) + time_mlp(p, "time_embed.", t, dim_t)
    lin, last = layer_plan(p)
    for n, j in enumerate(lin):
        z = F.linear(h, p[f"model.{j}.weight"], p[f"model.{j}.bias"])
        a = F.relu(z) if n == 0 else F.leaky_relu(z, 0.01)
        b = f"model.{j + 2}."
        h = F.batch_norm(a, p[b + "running_mean"], p[b + "running_var"], p[b + "weight"], p[b + "bias"],
                         training=False, eps=BN_EPS)
    return F.linear(h, p[f"model.{last}.weight"], p[f"model.{last}.bias"])


def cond_grad(p, x, t, y, scale=1.0, dim_t=128):
    """`cond_fn` (`pipeline_utils.py:853-864`): d/dx sum_b log_softmax(logits)[b, y_b], times scale."""
    with torch.enable_grad():
        xin = x.detach().float().requires_grad_(True)
        lp = F.log_softmax(classifier_logits(p, xin, t, dim_t), dim=-1)
        sel = lp[torch.arange(len(lp)), y.reshape(-1)]
        return torch.autograd.grad(sel.sum(), xin)[0] * scale
