"""Thin torch-tensor wrappers over the C ABI (one function per `clava_*` entry point).

Only plumbing lives here: argument checking, output allocation through torch's
caching allocator, and the raw-pointer call.  All arithmetic is in the HIP
library.
"""
import ctypes

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


def _rs(t):
    """Row stride of a 2-D tensor as a leading dimension.  torch reports an arbitrary stride(0) for a one-row tensor (a
    [1, n] transpose of an [n, 1] weight has stride 1): any value >= the row length is equivalent there."""
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[-1])


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
    check(lib().clava_gemm_bias_act_f32(ptr(x), c_int(_rs(x)), ptr(w), c_int(_rs(w)), ptr(_f32c(bias, "bias")),
                                        ptr(rowadd), c_int(ldr), ptr(_f32c(in_scale, "in_scale")),
                                        ptr(_f32c(in_shift, "in_shift")), ptr(out), c_int(_rs(out)), c_int(M),
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
    check(lib().clava_gemm_dgrad_f32(ptr(dy), c_int(_rs(dy)), ptr(w), c_int(_rs(w)), ptr(_f32c(act_out, "act_out")),
                                     c_int(_rs(act_out) if act_out is not None else 0), c_float(slope),
                                     ptr(_f32c(col_scale, "col_scale")), c_float(alpha), ptr(out), c_int(_rs(out)),
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
    check(lib().clava_gemm_wgrad_f32(ptr(dy), c_int(_rs(dy)), ptr(x), c_int(_rs(x)), ptr(out), c_int(K),
                                     _lib.c_long(slab_stride), c_int(M), c_int(N), c_int(K), c_int(splits), stream_ptr()),
          "clava_gemm_wgrad_f32")
    return out


# ---- split-f16 family (csrc/split.h): float32 values as two float16 planes, products on the f16 matrix cores ----------
class SplitTensor:
    """A [rows, cols] matrix stored as split rows (see include/clava_hip.h): `buf` is a uint8 device tensor of
    rows * pitch bytes, `shift` the power-of-two scale of the planes."""
    __slots__ = ("buf", "rows", "cols", "shift", "pitch")

    def __init__(self, buf, rows, cols, shift, pitch):
        self.buf, self.rows, self.cols, self.shift, self.pitch = buf, rows, cols, shift, pitch


def split_pitch(cols):
    return 4 * ((cols + 31) // 32 * 32)


def split_empty(rows, cols, shift, device):
    pitch = split_pitch(cols)
    return SplitTensor(torch.zeros(max(rows * pitch, 1), device=device, dtype=torch.uint8), rows, cols, shift, pitch)


def overflow_flag(device):
    return torch.zeros(1, device=device, dtype=torch.int32)


def split_pack(x, shift=0, overflow=None, out=None):
    """float32 [rows, cols] -> SplitTensor (clava_split_pack_f32); `out` reuses a SplitTensor of the same shape."""
    require_cuda(x)
    x = _f32c(x, "x")
    rows, cols = x.shape
    if out is None:
        out = split_empty(rows, cols, shift, x.device)
    else:
        if (out.rows, out.cols) != (rows, cols):
            raise _lib.ClavaHipError("split_pack: `out` has another shape")
        out.shift = shift
    if rows:
        check(lib().clava_split_pack_f32(ptr(x), _lib.c_long(_rs(x)), _lib.c_long(rows), c_int(cols), c_int(shift),
                                         ptr(out.buf), _lib.c_long(out.pitch), ptr(overflow), stream_ptr()), "clava_split_pack_f32")
    return out


def split_pack_transposed(x, shift=0, overflow=None, out=None):
    """SplitTensor of x.T for a float32 [cols, rows] matrix x, without materialising the transpose."""
    require_cuda(x)
    x = _f32c(x, "x")
    cols, rows = x.shape
    if out is None:
        out = split_empty(rows, cols, shift, x.device)
    else:
        if (out.rows, out.cols) != (rows, cols):
            raise _lib.ClavaHipError("split_pack_transposed: `out` has another shape")
        out.shift = shift
    if rows:
        check(lib().clava_split_pack_transposed_f32(ptr(x), _lib.c_long(_rs(x)), _lib.c_long(rows), c_int(cols), c_int(shift),
                                                    ptr(out.buf), _lib.c_long(out.pitch), ptr(overflow), stream_ptr()),
              "clava_split_pack_transposed_f32")
    return out


class _PackDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("ld", ctypes.c_longlong), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("scale", ctypes.c_float), ("dst", ctypes.c_void_p), ("pitch", ctypes.c_longlong), ("dst_t", ctypes.c_void_p),
                ("pitch_t", ctypes.c_longlong)]


class PackPlan:
    """A fixed list of (float32 matrix, shift, SplitTensor or None, SplitTensor of the transpose or None) packed by ONE
    launch (clava_split_pack_multi_f32).  The descriptors live in device memory and hold raw pointers: the plan is valid
    while every tensor named keeps its storage (`key` lets the owner notice when that changed)."""

    def __init__(self, items, device):
        arr = (_PackDesc * len(items))()
        self.keep = []
        self.max_tiles = 1
        for d, (src, shift, dst, dst_t) in zip(arr, items):
            require_cuda(src)
            if src.dtype != torch.float32 or src.dim() != 2 or src.stride(1) != 1:
                raise _lib.ClavaHipError("PackPlan: float32 row-major matrices expected")
            rows, cols = src.shape
            d.src, d.ld, d.rows, d.cols, d.scale = src.data_ptr(), _rs(src), rows, cols, 2.0 ** shift
            if dst is not None:
                if (dst.rows, dst.cols) != (rows, cols):
                    raise _lib.ClavaHipError("PackPlan: destination has another shape")
                dst.shift = shift
                d.dst, d.pitch = dst.buf.data_ptr(), dst.pitch
            if dst_t is not None:
                if (dst_t.rows, dst_t.cols) != (cols, rows):
                    raise _lib.ClavaHipError("PackPlan: transposed destination has another shape")
                dst_t.shift = shift
                d.dst_t, d.pitch_t = dst_t.buf.data_ptr(), dst_t.pitch
            self.max_tiles = max(self.max_tiles, ((rows + 63) // 64) * ((cols + 63) // 64))
            self.keep.append((src, dst, dst_t, shift))
        self.n = len(items)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.dev = raw.to(device)
        self.key = tuple((s.data_ptr(), sh) for s, sh, _, _ in items)

    def run(self, overflow=None):
        for _, dst, dst_t, shift in self.keep:       # a cached plan may run again after another plan re-scaled its destinations
            if dst is not None:
                dst.shift = shift
            if dst_t is not None:
                dst_t.shift = shift
        check(lib().clava_split_pack_multi_f32(ptr(self.dev), c_int(self.n), c_int(self.max_tiles), ptr(overflow), stream_ptr()),
              "clava_split_pack_multi_f32")


def split_transpose(s, out=None):
    """SplitTensor of the transpose (clava_split_transpose): planes moved as they are, same shift."""
    if out is None:
        out = split_empty(s.cols, s.rows, s.shift, s.buf.device)
    elif (out.rows, out.cols) != (s.cols, s.rows):
        raise _lib.ClavaHipError("split_transpose: `out` has another shape")
    out.shift = s.shift
    if s.rows and s.cols:
        check(lib().clava_split_transpose(ptr(s.buf), _lib.c_long(s.pitch), _lib.c_long(s.rows), c_int(s.cols), ptr(out.buf),
                                          _lib.c_long(out.pitch), stream_ptr()), "clava_split_transpose")
    return out


def linear_wgrad_f16x3(dyt, xt, splits, out, slab_stride, tile_config=-1):
    """Partial products of dy.T @ x over `splits` batch ranges on the f16 matrix cores (clava_gemm_wgrad_f16x3):
    dyt = dy^T [N_out, M], xt = x^T [K_in, M] as SplitTensors; fills `out` (slabs `slab_stride` floats apart, rows K_in apart)."""
    n_out, M = dyt.rows, dyt.cols
    k_in = xt.rows
    if xt.cols != M:
        raise _lib.ClavaHipError("linear_wgrad_f16x3: batch mismatch")
    nkb = (M + 31) // 32
    while splits > 1 and (splits - 1) * ((nkb + splits - 1) // splits) >= nkb:
        splits -= 1
    check(lib().clava_gemm_wgrad_f16x3(ptr(dyt.buf), _lib.c_long(dyt.pitch), c_int(dyt.shift), ptr(xt.buf), _lib.c_long(xt.pitch),
                                       c_int(xt.shift), ptr(out), c_int(k_in), _lib.c_long(slab_stride), c_int(M), c_int(n_out),
                                       c_int(k_in), c_int(splits), c_int(tile_config), stream_ptr()), "clava_gemm_wgrad_f16x3")
    return splits


def linear_wgrad_f16x3_tn(dy, x, splits, out, slab_stride, bias_out=None, ldw=None):
    """`linear_wgrad_f16x3` from the batch-major SplitTensors themselves (clava_gemm_wgrad_f16x3_tn): dy [M, N_out],
    x [M, K_in], M a multiple of 32; no transposed copies.  `bias_out` (slabs, same stride) also receives the column sums
    of dy (the bias gradient).  `ldw`: row stride of the gradient in `out` (default K_in; larger when x holds only the
    first columns of the layer's input)."""
    M, n_out = dy.rows, dy.cols
    k_in = x.cols
    if x.rows != M or M % 32:
        raise _lib.ClavaHipError("linear_wgrad_f16x3_tn: batch mismatch or batch not a multiple of 32")
    nkb = M // 32
    while splits > 1 and (splits - 1) * ((nkb + splits - 1) // splits) >= nkb:
        splits -= 1
    check(lib().clava_gemm_wgrad_f16x3_tn(ptr(dy.buf), _lib.c_long(dy.pitch), c_int(dy.shift), ptr(x.buf), _lib.c_long(x.pitch),
                                          c_int(x.shift), ptr(out), c_int(k_in if ldw is None else ldw), _lib.c_long(slab_stride), ptr(bias_out), c_int(M),
                                          c_int(n_out), c_int(k_in), c_int(splits), stream_ptr()), "clava_gemm_wgrad_f16x3_tn")
    return splits


def split_unpack(s):
    out = torch.empty((s.rows, s.cols), device=s.buf.device, dtype=torch.float32)
    if s.rows:
        check(lib().clava_split_unpack_f32(ptr(s.buf), _lib.c_long(s.pitch), _lib.c_long(s.rows), c_int(s.cols), c_int(s.shift),
                                           ptr(out), _lib.c_long(_rs(out)), stream_ptr()), "clava_split_unpack_f32")
    return out


def linear_f16x3(x, w, bias=None, act=None, rowadd=None, act_out=None, slope=0.0, alpha=1.0, out_shift=None, out_f32=False,
                 overflow=None, tile_config=-1, out=None, both=False, out2=None):
    """alpha * act'(act_out) * act(x @ w.T + bias + rowadd) with x, w SplitTensors (clava_gemm_f16x3).
    Returns a SplitTensor with shift `out_shift`, or a float32 tensor when out_f32, or (SplitTensor, float32) when `both`
    (`out` = the SplitTensor to fill, `out2` = the float32 tensor to fill)."""
    M, K = x.rows, x.cols
    N = w.rows
    if w.cols != K:
        raise _lib.ClavaHipError(f"linear_f16x3: x is [{M},{K}] but w is [{w.rows},{w.cols}]")
    y_split = y_f32 = None
    if both:
        y_split = out if out is not None else split_empty(M, N, 0 if out_shift is None else out_shift, x.buf.device)
        if out is not None and out_shift is not None:
            y_split.shift = out_shift
        y_f32 = out2 if out2 is not None else torch.empty((M, N), device=x.buf.device, dtype=torch.float32)
    elif out_f32:
        y_f32 = out if out is not None else torch.empty((M, N), device=x.buf.device, dtype=torch.float32)
    else:
        y_split = out if out is not None else split_empty(M, N, 0 if out_shift is None else out_shift, x.buf.device)
        if out is not None and out_shift is not None:
            y_split.shift = out_shift
    check(lib().clava_gemm_f16x3(ptr(x.buf), _lib.c_long(x.pitch), c_int(x.shift), ptr(w.buf), _lib.c_long(w.pitch), c_int(w.shift),
                                 ptr(_f32c(bias, "bias")), ptr(_f32c(rowadd, "rowadd")), c_int(_ACT[act]),
                                 ptr(act_out.buf) if act_out is not None else None,
                                 _lib.c_long(act_out.pitch if act_out is not None else 0), c_float(slope), c_float(alpha),
                                 ptr(y_split.buf) if y_split is not None else None,
                                 _lib.c_long(y_split.pitch if y_split is not None else 0),
                                 c_int(y_split.shift if y_split is not None else 0), ptr(y_f32),
                                 c_int(_rs(y_f32) if y_f32 is not None else 0), c_int(M), c_int(N), c_int(K), ptr(overflow),
                                 c_int(tile_config), stream_ptr()), "clava_gemm_f16x3")
    if both:
        return y_split, y_f32
    return y_f32 if out_f32 else y_split
