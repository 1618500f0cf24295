"""ctypes mirrors of the structs in include/clava_hip.h and helpers that fill them from torch tensors.

Keeping every tensor whose pointer was handed to the library alive until the work is enqueued is the caller's job:
the builders below return `(struct, keepalive)` pairs for that reason.
"""
import ctypes

import torch

from ._lib import ClavaHipError

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)


class CSchedule(ctypes.Structure):
    _fields_ = [("T", ctypes.c_int),
                ("sqrt_recip_alphas_cumprod", ctypes.c_void_p),
                ("sqrt_recipm1_alphas_cumprod", ctypes.c_void_p),
                ("posterior_mean_coef1", ctypes.c_void_p),
                ("posterior_mean_coef2", ctypes.c_void_p),
                ("model_variance", ctypes.c_void_p),
                ("model_sigma", ctypes.c_void_p),
                ("log_alpha", ctypes.c_void_p),
                ("log_1_min_alpha", ctypes.c_void_p),
                ("log_cumprod_alpha", ctypes.c_void_p),
                ("log_1_min_cumprod_alpha", ctypes.c_void_p),
                ("sqrt_alphas_cumprod", ctypes.c_void_p),
                ("sqrt_one_minus_alphas_cumprod", ctypes.c_void_p)]


class CNet(ctypes.Structure):
    _fields_ = [("n_layers", ctypes.c_int),
                ("dims", c_int_p),
                ("w", ctypes.POINTER(ctypes.c_void_p)),
                ("b", ctypes.POINTER(ctypes.c_void_p)),
                ("act", c_int_p),
                ("bn_scale", ctypes.POINTER(ctypes.c_void_p)),
                ("bn_shift", ctypes.POINTER(ctypes.c_void_p))]


class CChain(ctypes.Structure):
    _fields_ = [("sched", CSchedule),
                ("t_start", ctypes.c_int),
                ("n_steps", ctypes.c_int),
                ("n_num", ctypes.c_int), ("m", ctypes.c_int), ("sumK", ctypes.c_int),
                ("offsets", ctypes.c_void_p),
                ("log_k", ctypes.c_void_p),
                ("denoiser", ctypes.POINTER(CNet)),
                ("denoiser_emb", ctypes.c_void_p),
                ("classifier", ctypes.POINTER(CNet)),
                ("classifier_emb", ctypes.c_void_p),
                ("y", ctypes.c_void_p),
                ("guidance_scale", ctypes.c_float),
                ("seed", ctypes.c_ulonglong), ("row_offset", ctypes.c_ulonglong),
                ("noise", ctypes.c_void_p),
                ("uniforms", ctypes.c_void_p),
                ("first_step_f32", ctypes.c_int)]


def _dev_f32(t, name):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ClavaHipError(f"{name}: expected a contiguous float32 device tensor")
    return t


def void_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def make_net(weights, biases, acts, bn_scale=None, bn_shift=None):
    """weights[l]: [d_out, d_in] float32 device tensors.  Returns (CNet, keepalive)."""
    n = len(weights)
    dims = (ctypes.c_int * (n + 1))()
    dims[0] = weights[0].shape[1]
    for l, w in enumerate(weights):
        _dev_f32(w, f"w[{l}]")
        _dev_f32(biases[l], f"b[{l}]")
        if w.shape[1] != dims[l]:
            raise ClavaHipError(f"layer {l}: input width {w.shape[1]} != previous output width {dims[l]}")
        dims[l + 1] = w.shape[0]
    w_arr, b_arr = void_array(weights), void_array(biases)
    act_arr = (ctypes.c_int * n)(*acts)
    keep = [dims, w_arr, b_arr, act_arr, list(weights), list(biases)]
    net = CNet()
    net.n_layers = n
    net.dims = ctypes.cast(dims, c_int_p)
    net.w = ctypes.cast(w_arr, ctypes.POINTER(ctypes.c_void_p))
    net.b = ctypes.cast(b_arr, ctypes.POINTER(ctypes.c_void_p))
    net.act = ctypes.cast(act_arr, c_int_p)
    if bn_scale is not None:
        for t in list(bn_scale) + list(bn_shift):
            _dev_f32(t, "bn")
        s_arr, h_arr = void_array(bn_scale), void_array(bn_shift)
        net.bn_scale = ctypes.cast(s_arr, ctypes.POINTER(ctypes.c_void_p))
        net.bn_shift = ctypes.cast(h_arr, ctypes.POINTER(ctypes.c_void_p))
        keep += [s_arr, h_arr, list(bn_scale), list(bn_shift)]
    return net, keep


def make_schedule(T, tabs):
    """tabs: dict name -> device tensor (float32, or float64 for model_variance / model_sigma)."""
    s = CSchedule()
    s.T = int(T)
    for name, _ in CSchedule._fields_[1:]:
        t = tabs[name]
        want = torch.float64 if name in ("model_variance", "model_sigma") else torch.float32
        if not t.is_cuda or t.dtype != want or t.numel() != T or not t.is_contiguous():
            raise ClavaHipError(f"schedule table {name}: expected {T} contiguous {want} values on the device")
        setattr(s, name, t.data_ptr())
    return s
