"""GPU parity of the post-sampling decoding: quantile inverse-transform and round_columns against
scikit-learn's / the reference's own outputs (tests/golden/g12_postprocess.npz) and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden

from clavaddpm_amd import postprocess as pp
from oracle import postprocess as opp

DEV = "cuda"


def _close(got, ref):
    """float64 results: scipy's cephes erf/erfc vs the device library's differ by an ulp or two before the linear
    interpolation, so 1e-12 relative to the column's scale; NaNs must sit in the same places."""
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    scale = np.nanmax(np.abs(ref), axis=0, keepdims=True)
    err = np.abs(got - ref) / np.maximum(scale, 1e-300)
    assert np.nanmax(err) < 1e-12, np.nanmax(err)


def test_g12_quantile_inverse_golden():
    g = golden("g12_postprocess.npz")
    q, r, x = g["quantiles"].to(DEV), g["references"].to(DEV), g["x"].to(DEV)
    got = pp.quantile_inverse(x, q, r).cpu().numpy()
    ref = g["inverse"].numpy()
    _close(got, ref)
    # the 1e-7 clamps return the end quantiles themselves: exact
    qn = g["quantiles"].numpy()
    for c in range(ref.shape[1]):
        for end in (qn[0, c], qn[-1, c]):
            m = ref[:, c] == end
            assert np.array_equal(got[m, c], ref[m, c])
    # in place, and on rows embedded in a wider buffer
    wide = torch.full((x.shape[0], 9), 7.0, device=DEV, dtype=torch.float64)
    wide[:, 2:7] = x
    view = wide[:, 2:7]
    pp.quantile_inverse(view, q, r, out=view)
    _close(wide[:, 2:7].cpu().numpy(), ref)
    assert bool((wide[:, :2] == 7.0).all()) and bool((wide[:, 7:] == 7.0).all())


def test_g12_round_columns_golden_exact():
    g = golden("g12_postprocess.npz")
    x = g["inverse"].to(DEV).clone()
    out = pp.round_columns(g["real"].numpy(), x, g["cols"].tolist())
    assert out is x
    got, ref = x.cpu().numpy(), g["rounded"].numpy()
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok], ref[ok]) and np.array_equal(np.isnan(got), ~ok)      # value snapping: bit-exact
    assert pp.discrete_columns(g["real"].numpy()) == [1, 3, 4]  # the integer-valued columns with <= 32 distinct values


@pytest.mark.parametrize("N,d,nq", [(1, 1, 10), (257, 3, 10), (200000, 7, 1000), (5000, 15, 166)])
def test_quantile_inverse_vs_oracle(N, d, nq):
    rng = np.random.RandomState(N + d)
    real = rng.standard_normal((max(nq * 30, 300), d)) * rng.uniform(0.1, 100, (1, d)) + rng.uniform(-5, 5, (1, d))
    real[:, 0] = np.round(real[:, 0])                            # repeated values -> flat quantile segments
    refs = np.linspace(0, 1, nq, endpoint=True)
    quant = np.nanpercentile(real, refs * 100, axis=0)
    quant = np.maximum.accumulate(quant)                         # what QuantileTransformer.fit does
    x = rng.standard_normal((N, d)) * 2.5
    ref = opp.quantile_inverse(x, quant, refs)
    f = lambda a: torch.from_numpy(a).to(DEV)
    _close(pp.quantile_inverse(f(x), f(quant), f(refs)).cpu().numpy(), ref)


def test_round_columns_ties_and_edges():
    real = np.array([[1.0, 10.0], [2.0, 10.0], [4.0, 30.0], [2.0, -5.0]])
    x = np.array([[1.5, 20.0], [3.0, 12.5], [-9.0, 1e9], [2.0, np.nan], [3.999999, 2.5]])
    ref = opp.round_columns(real, x, [0, 1])
    got = pp.round_columns(real, torch.from_numpy(x.copy()).to(DEV), [0, 1]).cpu().numpy()
    assert np.array_equal(got, ref)                              # ties go to the smaller value; NaN takes the first value
    assert got[0, 0] == 1.0 and got[1, 0] == 2.0 and got[0, 1] == 10.0 and got[3, 1] == -5.0
    rng = np.random.RandomState(1)
    real = rng.randint(-20, 20, (5000, 3)).astype(float)
    x = rng.standard_normal((100000, 3)) * 15
    x[::3] = np.round(x[::3] * 2) / 2                            # many exact half-way points
    ref = opp.round_columns(real, x, [0, 2])
    got = pp.round_columns(real, torch.from_numpy(x.copy()).to(DEV), [0, 2]).cpu().numpy()
    assert np.array_equal(got, ref)
    # idempotence: rounding a rounded table changes nothing
    again = pp.round_columns(real, torch.from_numpy(got.copy()).to(DEV), [0, 2]).cpu().numpy()
    assert np.array_equal(again, got)
