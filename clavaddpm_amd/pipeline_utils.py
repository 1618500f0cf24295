"""Cluster-guidance classifier and its fused input gradient on the HIP library.

Mirrors the hot-path pieces of the reference `pipeline_utils.py`: `Classifier`
(`:584-630`) and the `cond_fn` closures of `conditional_sampling_by_group_size`
(`:853-864`) / `conditional_sampling` (`:982-992`), packaged as a callable object
(`ClassifierGuidance`) that the diffusion sampler recognises and runs inside the
fused chain instead of calling back into Python every step.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, _native
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, c_float, c_int, check, lib, ptr, require_cuda, stream_ptr
from .tab_ddpm.modules import run_stack, time_embed_forward, timestep_embedding  # noqa: F401  (re-exported like the reference)


class Classifier(nn.Module):
    """`Classifier(d_in, d_out, dim_t, hidden_sizes, dropout_prob=0.5, num_heads=2, num_layers=1)` (`:584-623`).

    state_dict names match the reference (`proj`, `time_embed.{0,2}`, `model.{j}`).  The reference also builds an
    `nn.Transformer` it never calls (`:591-595`, call commented out at `:628`; 4.5 M dead parameters); pass
    `keep_unused_transformer=True` to allocate it too (identical RNG consumption / strict state_dict loading).
    """

    def __init__(self, d_in, d_out, dim_t, hidden_sizes, dropout_prob=0.5, num_heads=2, num_layers=1,
                 keep_unused_transformer=False):
        super().__init__()
        self.dim_t = dim_t
        self.proj = nn.Linear(d_in, dim_t)
        if keep_unused_transformer:
            self.transformer_layer = nn.Transformer(d_model=dim_t, nhead=num_heads, num_encoder_layers=num_layers)
        self.time_embed = nn.Sequential(nn.Linear(dim_t, dim_t), nn.SiLU(), nn.Linear(dim_t, dim_t))
        layers = [nn.Linear(dim_t, hidden_sizes[0]), nn.ReLU(), nn.BatchNorm1d(hidden_sizes[0]), nn.Dropout(p=dropout_prob)]
        for i in range(len(hidden_sizes) - 1):
            layers += [nn.Linear(hidden_sizes[i], hidden_sizes[i + 1]), nn.LeakyReLU(),
                       nn.BatchNorm1d(hidden_sizes[i + 1]), nn.Dropout(p=dropout_prob)]
        layers.append(nn.Linear(hidden_sizes[-1], d_out))
        self.model = nn.Sequential(*layers)

    def stack(self):
        """The eval-mode network as one Linear stack with BatchNorm folded to per-channel scale/shift."""
        ws, bs, acts, sc, sh = [self.proj.weight], [self.proj.bias], [ACT_NONE], [None], [None]
        mods = list(self.model)
        j = 0
        while j < len(mods):
            lin = mods[j]
            ws.append(lin.weight)
            bs.append(lin.bias)
            if j + 2 < len(mods):
                acts.append(ACT_RELU if isinstance(mods[j + 1], nn.ReLU) else ACT_LEAKY)
                bn = mods[j + 2]
                s = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
                sc.append(s)
                sh.append((bn.bias.detach() - bn.running_mean * s).float().contiguous())
                j += 4
            else:
                acts.append(ACT_NONE)
                sc.append(None)
                sh.append(None)
                j += 1
        return ws, bs, acts, sc, sh

    def forward(self, x, timesteps):
        require_cuda(x, timesteps)
        if self.training:
            raise NotImplementedError("Classifier.forward runs in eval mode (guidance); train it with "
                                      "clavaddpm_amd.classifier_training.ClassifierTrainer, which runs the train-mode "
                                      "forward, the backward and AdamW on the HIP library")
        emb = time_embed_forward(self.time_embed, timesteps, self.dim_t)
        ws, bs, acts, sc, sh = self.stack()
        if torch.is_grad_enabled() and x.requires_grad:
            # the reference's guidance closure differentiates the logits w.r.t. the input (pipeline_utils.py:856-864:
            # x_in.requires_grad_(True), torch.autograd.grad(selected.sum(), x_in)); weights are constants in eval mode
            return _ClassifierInputGradFn.apply(x, emb, [w.detach() for w in ws], [b.detach() for b in bs], acts, sc, sh)
        return run_stack(ws, bs, acts, x.float(), rowadd0=emb, bn_scale=sc, bn_shift=sh)[-1]


class _ClassifierInputGradFn(torch.autograd.Function):
    """logits = classifier(x, t) in eval mode with d(logits)/d(x) for autograd: forward = clava_net_forward_f32 (keeps the
    per-layer outputs), backward = clava_net_input_grad_f32.  Only the input is differentiable (guidance never updates the
    classifier); this is what lets the reference's `cond_fn` closure run unchanged on this class."""

    @staticmethod
    def forward(ctx, x, emb, ws, bs, acts, sc, sh):
        xin = x.detach().float().contiguous()
        outs = run_stack(ws, bs, acts, xin, rowadd0=emb, bn_scale=sc, bn_shift=sh)
        ctx.net_args = (ws, bs, acts, sc, sh)
        ctx.outs = outs
        ctx.x_dtype = x.dtype
        return outs[-1]

    @staticmethod
    def backward(ctx, dlogits):
        ws, bs, acts, sc, sh = ctx.net_args
        outs = ctx.outs
        B = dlogits.shape[0]
        dl = dlogits.detach().float().contiguous()
        net, keep = _native.make_net(ws, bs, acts, sc, sh)
        grads = [torch.empty((B, w.shape[1]), device=dl.device, dtype=torch.float32) for w in ws]
        if B:
            check(lib().clava_net_input_grad_f32(ctypes.byref(net), c_int(B), _native.void_array(outs), ptr(dl),
                                                 _native.void_array(grads), c_float(1.0), stream_ptr()),
                  "clava_net_input_grad_f32")
        del keep
        return grads[0].to(ctx.x_dtype), None, None, None, None, None, None


def logsoftmax_select_grad(logits, y):
    """onehot(y) - softmax(logits): gradient of sum_b log_softmax(logits)[b, y_b] w.r.t. logits."""
    require_cuda(logits, y)
    B, C = logits.shape
    y32 = y.reshape(-1).to(torch.int32).contiguous()
    out = torch.empty((B, C), device=logits.device, dtype=torch.float32)
    check(lib().clava_logsoftmax_select_grad_f32(ptr(logits), c_int(logits.stride(0)), ptr(y32), ptr(out), c_int(B), c_int(C),
                                                 stream_ptr()), "clava_logsoftmax_select_grad_f32")
    return out


class ClassifierGuidance:
    """Callable with the `cond_fn(x, t, y=None, remove_first_col=False)` contract of the reference closures.

    Returns grad_x sum_b log_softmax(classifier(x, t))[b, y_b] * classifier_scale as float32 [B, n_num].
    """

    def __init__(self, classifier, classifier_scale=1.0):
        self.classifier = classifier
        self.classifier_scale = float(classifier_scale)

    def __call__(self, x, t, y=None, remove_first_col=False):
        assert y is not None
        clf = self.classifier
        if clf.training:
            raise _lib.ClavaHipError("guidance needs classifier.eval()")
        require_cuda(x, t)
        xin = (x[:, 1:] if remove_first_col else x).detach().float().contiguous()
        B = xin.shape[0]
        emb = time_embed_forward(clf.time_embed, t, clf.dim_t)
        ws, bs, acts, sc, sh = clf.stack()
        outs = run_stack(ws, bs, acts, xin, rowadd0=emb, bn_scale=sc, bn_shift=sh)
        dlog = logsoftmax_select_grad(outs[-1], y.to(xin.device))
        net, keep = _native.make_net([w.detach() for w in ws], [b.detach() for b in bs], acts, sc, sh)
        grads = [torch.empty((B, w.shape[1]), device=xin.device, dtype=torch.float32) for w in ws]
        check(lib().clava_net_input_grad_f32(ctypes.byref(net), c_int(B), _native.void_array(outs), ptr(dlog),
                                             _native.void_array(grads), c_float(self.classifier_scale), stream_ptr()),
              "clava_net_input_grad_f32")
        return grads[0]


def make_cond_fn(classifier, classifier_scale=1.0):
    """Build the guidance callable the reference defines inline at `pipeline_utils.py:853-864`."""
    return ClassifierGuidance(classifier, classifier_scale)
