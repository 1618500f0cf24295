"""Denoiser network with the reference's constructor and state_dict layout, running on the HIP library.

Mirrors `tab_ddpm/modules.py` of the reference: `timestep_embedding` (`:20-38`),
`MLP` / `MLP.Block` (`:119-243`) and `MLPDiffusion` (`:425-460`).  `ResNetDiffusion`
is not provided: its forward is broken upstream (`:488` vs `:417`) and every config
uses `"model_type": "mlp"`.

Parameters are ordinary `nn.Parameter`s created in the reference's construction
order (mlp blocks, head, proj, time_embed), so a seeded construction yields the
same initial weights and checkpoints load either way.  There is no CPU compute
path: tensors must live on the GPU.
"""
import ctypes
import math

import torch
import torch.nn as nn

from .. import _lib, _native, ops
from .._lib import ACT_NONE, ACT_RELU, c_int, check, lib, ptr, require_cuda, stream_ptr

_FREQS = {}


def _freqs(dim, device, max_period=10000):
    """exp(-ln(max_period) * i / half) as float32, cached per device.

    The exponent is formed in float32 exactly as the reference forms it (`:31-33`); the exponential itself is
    evaluated in float64 and rounded once.  torch's float32 `exp` is only 1-ulp accurate and its result depends on the
    host CPU's vector ISA (the build container and the GPU box disagree in one of 64 entries), which moves
    cos/sin(t * freq) by up to t * 2^-24; the correctly rounded value is machine-independent."""
    key = (dim, str(device), max_period)
    if key not in _FREQS:
        half = dim // 2
        arg = -math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half
        _FREQS[key] = torch.exp(arg.double()).float().to(device)
    return _FREQS[key]


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embedding [N, dim] of integer (or float) timesteps; same contract as the reference (`:20-38`)."""
    require_cuda(timesteps)
    half = dim // 2
    t = timesteps
    if t.dtype != torch.int64:
        if t.is_floating_point() and bool((t != t.round()).any()):
            raise _lib.ClavaHipError("timestep_embedding: fractional timesteps are not supported by the HIP kernel")
        t = t.long()
    t = t.contiguous()
    emb = torch.empty((t.shape[0], 2 * half), device=t.device, dtype=torch.float32)
    if t.shape[0] > 0:
        check(lib().clava_timestep_embedding_f32(ptr(t), c_int(t.shape[0]), ptr(_freqs(dim, t.device, max_period)),
                                                 c_int(2 * half), ptr(emb), stream_ptr()), "clava_timestep_embedding_f32")
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def time_embed_forward(time_embed, timesteps, dim_t):
    """`time_embed(timestep_embedding(t))` = Linear -> SiLU -> Linear (`:445-449`, `:452`) on the HIP GEMM."""
    te = timestep_embedding(timesteps, dim_t)
    h = ops.linear(te, time_embed[0].weight, time_embed[0].bias, act="silu")
    return ops.linear(h, time_embed[2].weight, time_embed[2].bias)


class MLP(nn.Module):
    """Linear->ReLU->Dropout blocks followed by a Linear head (the `make_baseline` MLP of the reference, `:119-243`)."""

    class Block(nn.Module):
        def __init__(self, *, d_in, d_out, bias, activation, dropout):
            super().__init__()
            if activation != "ReLU":
                raise ValueError("only the ReLU baseline MLP is supported")
            self.linear = nn.Linear(d_in, d_out, bias)
            self.activation = nn.ReLU()
            self.dropout = nn.Dropout(dropout)

    def __init__(self, *, d_in, d_layers, dropouts, activation, d_out):
        super().__init__()
        if isinstance(dropouts, float):
            dropouts = [dropouts] * len(d_layers)
        assert len(d_layers) == len(dropouts)
        self.blocks = nn.ModuleList([
            MLP.Block(d_in=d_layers[i - 1] if i else d_in, d_out=d, bias=True, activation=activation, dropout=p)
            for i, (d, p) in enumerate(zip(d_layers, dropouts))])
        self.head = nn.Linear(d_layers[-1] if d_layers else d_in, d_out)

    @classmethod
    def make_baseline(cls, d_in, d_layers, dropout, d_out):
        assert isinstance(dropout, float)
        if len(d_layers) > 2:
            assert len(set(d_layers[1:-1])) == 1, \
                "if d_layers contains more than two elements, all but the first and last must be equal"
        return cls(d_in=d_in, d_layers=d_layers, dropouts=dropout, activation="ReLU", d_out=d_out)

    def layers(self):
        """(weights, biases, activation codes) of the stack, in order."""
        ws = [b.linear.weight for b in self.blocks] + [self.head.weight]
        bs = [b.linear.bias for b in self.blocks] + [self.head.bias]
        return ws, bs, [ACT_RELU] * len(self.blocks) + [ACT_NONE]

    def forward(self, x):
        ws, bs, acts = self.layers()
        return run_stack(ws, bs, acts, x.float())[-1]


def run_stack(weights, biases, acts, x, rowadd0=None, bn_scale=None, bn_shift=None):
    """Forward a Linear stack through clava_net_forward_f32; returns the list of per-layer outputs."""
    require_cuda(x)
    if x.dtype != torch.float32 or x.stride(-1) != 1:
        x = x.float().contiguous()
    M = x.shape[0]
    ws = [w.detach() for w in weights]
    bs = [b.detach() for b in biases]
    net, keep = _native.make_net(ws, bs, acts, bn_scale, bn_shift)
    outs = [torch.empty((M, w.shape[0]), device=x.device, dtype=torch.float32) for w in ws]
    if M == 0:
        return outs
    ldr = 0
    if rowadd0 is not None:
        if rowadd0.dtype != torch.float32 or not rowadd0.is_contiguous():
            rowadd0 = rowadd0.float().contiguous()
        ldr = rowadd0.shape[-1] if (rowadd0.dim() == 2 and rowadd0.shape[0] == M and M > 1) else 0
    out_arr = _native.void_array(outs)
    check(lib().clava_net_forward_f32(ctypes.byref(net), ptr(x), c_int(x.stride(0)), c_int(M), ptr(rowadd0), c_int(ldr),
                                      out_arr, stream_ptr()), "clava_net_forward_f32")
    return outs


class MLPDiffusion(nn.Module):
    """`MLPDiffusion(d_in, num_classes, is_y_cond, rtdl_params, dim_t=128)`; `forward(x, timesteps, y=None)` (`:425-460`)."""

    def __init__(self, d_in, num_classes, is_y_cond, rtdl_params, dim_t=128):
        super().__init__()
        self.dim_t = dim_t
        self.num_classes = num_classes
        self.is_y_cond = is_y_cond
        rtdl_params["d_in"] = dim_t          # the reference mutates the caller's dict the same way (:434-435)
        rtdl_params["d_out"] = d_in
        self.mlp = MLP.make_baseline(**rtdl_params)
        if self.num_classes > 0 and is_y_cond == "embedding":
            self.label_emb = nn.Embedding(self.num_classes, dim_t)
        elif self.num_classes == 0 and is_y_cond == "embedding":
            self.label_emb = nn.Linear(1, dim_t)
        self.proj = nn.Linear(d_in, dim_t)
        self.time_embed = nn.Sequential(nn.Linear(dim_t, dim_t), nn.SiLU(), nn.Linear(dim_t, dim_t))

    def stack(self):
        """proj + mlp as one Linear stack: (weights, biases, activation codes)."""
        ws, bs, acts = self.mlp.layers()
        return [self.proj.weight] + ws, [self.proj.bias] + bs, [ACT_NONE] + acts

    def forward(self, x, timesteps, y=None):
        require_cuda(x, timesteps)
        if self.training and any(b.dropout.p > 0 for b in self.mlp.blocks):
            raise NotImplementedError("dropout > 0 is not implemented on the HIP path (every ClavaDDPM config uses 0.0)")
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            from ..training import mlp_diffusion_autograd
            return mlp_diffusion_autograd(self, x, timesteps, y)
        emb = time_embed_forward(self.time_embed, timesteps, self.dim_t)
        if self.is_y_cond == "embedding" and y is not None:
            emb = emb + self._label_term(y)
        ws, bs, acts = self.stack()
        return run_stack(ws, bs, acts, x.float(), rowadd0=emb)[-1]

    def _label_term(self, y):
        # `emb += F.silu(self.label_emb(y))` (:453-458).  ClavaDDPM always builds the denoiser with is_y_cond='none'
        # (pipeline_modules.py:52); the embedding branch is kept for API completeness on small torch ops.
        if self.num_classes > 0:
            return torch.nn.functional.silu(self.label_emb(y.squeeze()))
        return torch.nn.functional.silu(self.label_emb(y.reshape(y.size(0), 1).float()))
