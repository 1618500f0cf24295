"""Post-sampling decoding of the numeric columns on the HIP library.

Mirrors what `conditional_sampling_by_group_size` / `sample_from_diffusion` do to a sampled table right after the chain
(`pipeline_utils.py:912-934`): `dataset.num_transform.inverse_transform(X_gen[:, :n])` for the quantile normaliser the
reference builds at `lib/data.py:217-223`, then `lib.round_columns` (`lib/data.py:627-632`) on the integer-valued columns.
Both run in float64 on the device, so a sampled table can stay in HBM from the last reverse step to its decoded form.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_int, check, lib, ptr, stream_ptr

c_ll = ctypes.c_longlong


def quantile_inverse(X, quantiles, references, out=None):
    """`QuantileTransformer(output_distribution='normal').inverse_transform` for float64 device tensors:
    X [N, d], quantiles [n_quantiles, d] (`quantiles_`), references [n_quantiles] (`references_`)."""
    _lib.require_cuda(X, quantiles, references)
    for t in (X, quantiles, references):
        if t.dtype != torch.float64:
            raise _lib.ClavaHipError("quantile_inverse: float64 tensors expected")
    if X.stride(1) != 1:
        X = X.contiguous()
    quantiles, references = quantiles.contiguous(), references.contiguous()
    N, d = X.shape
    if quantiles.shape != (references.shape[0], d):
        raise _lib.ClavaHipError(f"quantile_inverse: quantiles {tuple(quantiles.shape)} do not match X [{N},{d}] / references")
    if out is None:
        out = torch.empty((N, d), device=X.device, dtype=torch.float64)
    if N:
        check(lib().clava_quantile_inverse_f64(ptr(X), c_ll(X.stride(0)), c_ll(N), c_int(d), ptr(quantiles), ptr(references),
                                               c_int(references.shape[0]), ptr(out), c_ll(out.stride(0)), stream_ptr()),
              "clava_quantile_inverse_f64")
    return out


def inverse_transform(normalizer, X, device="cuda:0"):
    """`normalizer.inverse_transform(X)` for a fitted scikit-learn `QuantileTransformer(output_distribution='normal')`.
    X: numpy array or tensor [N, d]; returns a float64 tensor on `device`."""
    if getattr(normalizer, "output_distribution", None) != "normal":
        raise NotImplementedError("only QuantileTransformer(output_distribution='normal') (what ClavaDDPM fits) is implemented")
    f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(device)
    Xd = X.to(device).double() if torch.is_tensor(X) else f(X)
    return quantile_inverse(Xd, f(normalizer.quantiles_), f(normalizer.references_))


def round_columns(X_real, X_synth, columns):
    """`lib.round_columns(X_real, X_synth, columns)`: every listed column of the float64 device tensor X_synth is
    snapped (in place) to the nearest unique value of the same column of X_real (numpy array or tensor)."""
    _lib.require_cuda(X_synth)
    if X_synth.dtype != torch.float64 or X_synth.stride(1) != 1:
        raise _lib.ClavaHipError("round_columns: X_synth must be a row-major float64 device tensor")
    N = X_synth.shape[0]
    for col in columns:
        real = X_real[:, col]
        real = real.detach().cpu().numpy() if torch.is_tensor(real) else np.asarray(real)
        uniq = torch.from_numpy(np.unique(real.astype(np.float64))).to(X_synth.device)
        if N:
            check(lib().clava_round_to_values_f64(ptr(X_synth), c_ll(X_synth.stride(0)), c_ll(N), c_int(int(col)), ptr(uniq),
                                                  c_int(uniq.shape[0]), stream_ptr()), "clava_round_to_values_f64")
    return X_synth


def discrete_columns(X_num_real):
    """Columns the reference rounds (`pipeline_utils.py:924-928`): at most 32 unique values, all integers."""
    cols = []
    for col in range(X_num_real.shape[1]):
        u = np.unique(X_num_real[:, col])
        if len(u) <= 32 and ((u - np.round(u)) == 0).all():
            cols.append(col)
    return cols
