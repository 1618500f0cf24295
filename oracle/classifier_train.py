"""Guidance-classifier training (oracle; test infrastructure only).

Restates, with every random draw injected, the training loop of `train_classifier`
(`pipeline_utils.py:783-817`) around `numerical_forward_backward_log` (`:655-700`):

  * `Classifier.forward` in TRAIN mode (`:625-630`, layers `:603-623`): BatchNorm1d normalises with the batch
    mean / biased variance and moves its running statistics with momentum 0.1 and the unbiased variance;
    Dropout(p) keeps an element where `keep` is 1 and scales it by 1/(1-p);
  * `batch = gaussian_q_sample(x, t)` (`:677`), `F.cross_entropy(..., reduction='none')` (`:683`),
    `compute_top_k` (`:640-645`);
  * the backward call `loss.backward(loss * len(sub_batch) / len(batch))` (`:696-700`) with microbatching off
    (`split_microbatches(-1, ...)`, `:680`): the batch-mean loss is back-propagated with ITSELF as the incoming
    gradient, i.e. every gradient is that of 0.5 * mean_loss**2;
  * `optim.AdamW(classifier.parameters(), lr=lr)` (`:766`; torch defaults: betas (0.9, 0.999), eps 1e-8,
    weight_decay 0.01).

Parameters are a dict keyed like the reference `state_dict`; pinned by tests/golden/g10_classifier_training.npz,
which tests/golden/make_golden.py produced by running the reference's own functions.
"""
import torch
import torch.nn.functional as F

from .guidance import BN_EPS, classifier_logits, layer_plan
from .network import time_mlp

BN_MOMENTUM = 0.1  # nn.BatchNorm1d default


def classifier_logits_train(p, x, t, keep, drop_p=0.5, dim_t=128):
    """Train-mode forward.  `keep` [B, sum(hidden)]: the Dropout keep flags of the hidden blocks, concatenated in layer
    order.  Updates the BatchNorm running statistics in `p` in place and returns the logits."""
    h = F.linear(x, p["proj.weight"], p["proj.bias"]) + time_mlp(p, "time_embed.", t, dim_t)
    lin, last = layer_plan(p)
    off = 0
    B = x.shape[0]
    for n, j in enumerate(lin):
        z = F.linear(h, p[f"model.{j}.weight"], p[f"model.{j}.bias"])
        a = F.relu(z) if n == 0 else F.leaky_relu(z, 0.01)
        b = f"model.{j + 2}."
        mean = a.mean(dim=0)
        var = ((a - mean) ** 2).mean(dim=0)                         # biased: what the normalisation uses
        with torch.no_grad():
            p[b + "running_mean"].mul_(1 - BN_MOMENTUM).add_(mean.detach(), alpha=BN_MOMENTUM)
            p[b + "running_var"].mul_(1 - BN_MOMENTUM).add_(var.detach() * (B / (B - 1)), alpha=BN_MOMENTUM)
            if b + "num_batches_tracked" in p:
                p[b + "num_batches_tracked"] += 1
        y = (a - mean) / torch.sqrt(var + BN_EPS) * p[b + "weight"] + p[b + "bias"]
        C = y.shape[1]
        h = y * keep[:, off:off + C].to(y.dtype) / (1.0 - drop_p)
        off += C
    return F.linear(h, p[f"model.{last}.weight"], p[f"model.{last}.bias"])


def top_k_hits(logits, labels, k):
    """`compute_top_k(..., reduction='none')` (`:640-645`)."""
    _, top = torch.topk(logits, k, dim=-1)
    return (top == labels[:, None]).float().sum(dim=-1)


def q_sample(tb, x, t, noise):
    """`gaussian_q_sample` (`tab_ddpm/gaussian_multinomial_diffsuion.py:172-180`)."""
    a = torch.as_tensor(tb["sqrt_alphas_cumprod"])[t][:, None].to(x.dtype)
    b = torch.as_tensor(tb["sqrt_one_minus_alphas_cumprod"])[t][:, None].to(x.dtype)
    return a * x + b * noise


def forward_backward(tb, p, names, x, y, t, noise, keep, drop_p=0.5):
    """One `numerical_forward_backward_log(prefix='train')`: returns (mean loss, acc@1, acc@5 or None, {name: grad}).
    Runs in the dtype of `p` (float32 like the reference; a float64 copy measures how well conditioned a case is)."""
    dt = p[names[0]].dtype
    x, noise = x.to(dt), noise.to(dt)
    leaves = [p[k] for k in names]
    for w in leaves:
        w.requires_grad_(True)
        w.grad = None
    logits = classifier_logits_train(p, q_sample(tb, x, t, noise), t, keep, drop_p)
    rows = F.cross_entropy(logits, y, reduction="none")
    loss = rows.mean()
    loss.backward(loss.detach())                                     # `loss.backward(loss * len(sub)/len(batch))`, :700
    a1 = float(top_k_hits(logits.detach(), y, 1).mean())
    a5 = float(top_k_hits(logits.detach(), y, 5).mean()) if logits.shape[1] >= 5 else None
    return float(loss.detach()), a1, a5, {k: p[k].grad.detach().clone() for k in names}


def evaluate(tb, p, x, y, t, noise):
    """The `prefix='val'` call (`:803-814`): eval-mode forward, no gradient."""
    with torch.no_grad():
        logits = classifier_logits(p, q_sample(tb, x, t, noise), t)
        rows = F.cross_entropy(logits, y, reduction="none")
        return float(rows.mean()), float(top_k_hits(logits, y, 1).mean())


def run_loop(tb, p, xs, ys, ts, noises, keeps, lr, weight_decay=0.01, drop_p=0.5, grads_at=(), adam_state=None):
    """`steps = len(xs)` iterations of `:785-801` (train forward/backward + `classifier_optimizer.step()`).
    `p` holds leaf tensors and BatchNorm buffers and is updated in place.  `adam_state` = (exp_avg dict, exp_avg_sq dict,
    steps already taken) resumes the optimizer mid-run.  Returns (per-step (loss, acc1, acc5), {step: gradients} for the
    steps in `grads_at`)."""
    names = [k for k in p if p[k].is_floating_point() and not k.endswith(("running_mean", "running_var"))]
    opt = torch.optim.AdamW([p[k] for k in names], lr=lr, weight_decay=weight_decay)
    if adam_state is not None:
        m, v, done = adam_state
        for k in names:
            opt.state[p[k]] = {"step": torch.tensor(float(done)), "exp_avg": m[k].clone(), "exp_avg_sq": v[k].clone()}
    hist, kept = [], {}
    for step in range(len(xs)):
        loss, a1, a5, g = forward_backward(tb, p, names, xs[step], ys[step], ts[step], noises[step], keeps[step], drop_p)
        if step in grads_at:
            kept[step] = g
        opt.step()
        hist.append((loss, a1, a5))
    return hist, kept
