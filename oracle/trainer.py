"""Training loop (oracle; test infrastructure only).

Restates `Trainer.run_loop` / `_run_step` / `_anneal_lr` (`scripts/train.py:24-71`)
and `update_ema` (`scripts/utils_train.py:20-29`) around `oracle.loss.mixed_loss`,
with the per-step randomness injected.
"""
import torch

from .loss import mixed_loss


def adamw_reference_step(w, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """One `torch.optim.AdamW` update (the optimizer of `scripts/train.py:17`) on flat tensors,
    written out so the fused device kernel has an arithmetic statement to match."""
    w.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    w.addcdiv_(m, denom, value=-lr / bc1)


def run_loop(tb, T, p, batches, draws, n_num, K, lr, weight_decay, steps, ema_rate=0.999, exact_norm=False):
    """Runs `len(batches)` iterations of a `steps`-long schedule.

    `p` is a dict of leaf tensors (updated in place); `draws[i]` = dict(t, noise, u).
    Returns (per-step (loss_multi, loss_gauss) list, ema parameter dict).
    """
    names = list(p.keys())
    for k in names:
        p[k].requires_grad_(True)
    ema = {k: p[k].detach().clone() for k in names}
    opt = torch.optim.AdamW([p[k] for k in names], lr=lr, weight_decay=weight_decay)
    hist = []
    for step, (x, d) in enumerate(zip(batches, draws)):
        opt.zero_grad()
        lm, lg = mixed_loss(tb, T, p, x, n_num, K, d["t"], d["noise"], d.get("u"), exact_norm=exact_norm)
        (lm + lg).backward()
        opt.step()
        for g in opt.param_groups:                     # _anneal_lr, applied after the step (:24-28, :53)
            g["lr"] = lr * (1 - step / steps)
        hist.append((float(lm.detach()), float(lg.detach())))
        with torch.no_grad():                          # update_ema every step (:69)
            for k in names:
                ema[k].mul_(ema_rate).add_(p[k].detach(), alpha=1 - ema_rate)
    return hist, ema
