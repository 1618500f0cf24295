"""Post-sampling decoding of numeric columns (oracle; test infrastructure only).

Restates the inverse branch of scikit-learn's `QuantileTransformer._transform_col`
(`sklearn/preprocessing/_data.py`; called through `dataset.num_transform.inverse_transform` at
`pipeline_utils.py:913`, transformer built at `lib/data.py:217-223`) and `lib.round_columns` (`lib/data.py:627-632`).
Pinned by tests/golden/g12_postprocess.npz (scikit-learn's own `inverse_transform` and the reference's `round_columns`).
"""
import numpy as np
from scipy import stats

BOUNDS_THRESHOLD = 1e-7


def quantile_inverse(X, quantiles, references):
    """X [N,d] float64 -> inverse-transformed copy.  quantiles [n_q, d], references [n_q]."""
    out = np.array(X, dtype=np.float64, copy=True)
    for c in range(out.shape[1]):
        q = quantiles[:, c]
        with np.errstate(invalid="ignore"):
            u = stats.norm.cdf(out[:, c])
            lower = u - BOUNDS_THRESHOLD < 0
            upper = u + BOUNDS_THRESHOLD > 1
        ok = ~np.isnan(u)
        u[ok] = np.interp(u[ok], references, q)
        u[upper] = q[-1]
        u[lower] = q[0]
        out[:, c] = u
    return out


def round_columns(X_real, X_synth, columns):
    """Snap each listed column to the nearest unique training value; ties go to the smaller value (first argmin)."""
    out = np.array(X_synth, dtype=np.float64, copy=True)
    for col in columns:
        uniq = np.unique(X_real[:, col]).astype(float)
        dist = np.sqrt((out[:, col][:, None] - uniq[None, :]) ** 2)
        out[:, col] = uniq[dist.argmin(axis=1)]
    return out
