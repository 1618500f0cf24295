"""Thin torch-tensor wrappers over the C ABI (one function per `clava_*` entry point).

Only plumbing lives here: argument checking, output allocation through torch's
caching allocator, and the raw-pointer call.  All arithmetic is in the HIP
library.
"""
import torch

from . import _lib
from ._lib import c_float, c_int, check, lib, ptr, require_cuda, stream_ptr

_ACT = {None: 0, "none": 0, "relu": 1, "leaky_relu": 2, "silu": 3}


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise _lib.ClavaHipError(f"{name}: expected float32, got {t.dtype}")
    if t.stride(-1) != 1:
        raise _lib.ClavaHipError(f"{name}: innermost dimension must be contiguous")
    return t


def _ld(t):
    return t.stride(0) if t.dim() == 2 else 0


def linear(x, w, bias=None, act=None, rowadd=None, in_scale=None, in_shift=None, out=None):
    """act((x*in_scale+in_shift) @ w.T + bias + rowadd) via clava_gemm_bias_act_f32.

    `rowadd` is [M, N] or a single row [N] / [1, N] broadcast over the batch."""
    require_cuda(x, w, bias, rowadd, in_scale, in_shift)
    x, w = _f32c(x, "x"), _f32c(w, "w")
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise _lib.ClavaHipError(f"linear: x is [{M},{K}] but w is {tuple(w.shape)}")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    ldr = 0
    if rowadd is not None:
        rowadd = _f32c(rowadd, "rowadd")
        if rowadd.dim() == 2 and rowadd.shape[0] == M and M > 1:
            ldr = rowadd.stride(0)
        if rowadd.shape[-1] != N:
            raise _lib.ClavaHipError("linear: rowadd width mismatch")
    if M == 0:
        return out
    check(lib().clava_gemm_bias_act_f32(ptr(x), c_int(x.stride(0)), ptr(w), c_int(w.stride(0)), ptr(_f32c(bias, "bias")),
                                        ptr(rowadd), c_int(ldr), ptr(_f32c(in_scale, "in_scale")),
                                        ptr(_f32c(in_shift, "in_shift")), ptr(out), c_int(out.stride(0)), c_int(M),
                                        c_int(N), c_int(K), c_int(_ACT[act]), stream_ptr()), "clava_gemm_bias_act_f32")
    return out


def linear_dgrad(dy, w, act_out=None, slope=0.0, col_scale=None, alpha=1.0, out=None):
    """alpha * col_scale * act'(act_out) * (dy @ w) via clava_gemm_dgrad_f32."""
    require_cuda(dy, w, act_out, col_scale)
    dy, w = _f32c(dy, "dy"), _f32c(w, "w")
    M, N = dy.shape
    K = w.shape[1]
    if w.shape[0] != N:
        raise _lib.ClavaHipError("linear_dgrad: dy/w shape mismatch")
    if out is None:
        out = torch.empty((M, K), device=dy.device, dtype=torch.float32)
    if M == 0:
        return out
    check(lib().clava_gemm_dgrad_f32(ptr(dy), c_int(dy.stride(0)), ptr(w), c_int(w.stride(0)), ptr(_f32c(act_out, "act_out")),
                                     c_int(act_out.stride(0) if act_out is not None else 0), c_float(slope),
                                     ptr(_f32c(col_scale, "col_scale")), c_float(alpha), ptr(out), c_int(out.stride(0)),
                                     c_int(M), c_int(N), c_int(K), stream_ptr()), "clava_gemm_dgrad_f32")
    return out


def linear_wgrad(dy, x, splits=1, out=None, slab_stride=None):
    """Partial products of dy.T @ x over `splits` batch ranges: returns [splits, N_out, K_in] (or fills `out`, whose
    slabs are `slab_stride` floats apart)."""
    require_cuda(dy, x)
    dy, x = _f32c(dy, "dy"), _f32c(x, "x")
    M, N = dy.shape
    K = x.shape[1]
    if x.shape[0] != M:
        raise _lib.ClavaHipError("linear_wgrad: batch mismatch")
    if out is None:
        out = torch.empty((splits, N, K), device=dy.device, dtype=torch.float32)
    if slab_stride is None:
        slab_stride = N * K
    check(lib().clava_gemm_wgrad_f32(ptr(dy), c_int(dy.stride(0)), ptr(x), c_int(x.stride(0)), ptr(out), c_int(K),
                                     _lib.c_long(slab_stride), c_int(M), c_int(N), c_int(K), c_int(splits), stream_ptr()),
          "clava_gemm_wgrad_f32")
    return out
