"""GPU parity of the exact row matching against the brute-force oracle.  Parity against the reference's faiss
IVF search is UNPINNED (third-party approximate search, not installed, no recorded outputs): what is pinned here is the
reference's loop logic around an exact search."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from clavaddpm_amd import _lib, matching
from oracle import matching as om


def _tables(na, nb, d, seed, integer=False):
    rng = np.random.RandomState(seed)
    if integer:                                   # many exact ties and duplicate rows
        return rng.randint(0, 4, (na, d)).astype(np.float64), rng.randint(0, 4, (nb, d)).astype(np.float64)
    return rng.standard_normal((na, d)), rng.standard_normal((nb, d)) * 1.1 + 0.05


@pytest.mark.parametrize("na,nb,d,integer", [(300, 500, 5, False), (1, 1, 1, False), (700, 700, 1, False), (2000, 2000, 8, False),
                                             (513, 640, 3, True), (257, 4099, 20, False), (1200, 1300, 64, False),
                                             (400, 450, 65, False), (300, 300, 150, True)])
def test_unique_matching_equals_the_sequential_oracle(na, nb, d, integer):
    """Greedy nearest neighbour with removal, queries in row order: indices exact (ties to the lower row), squared float32
    distances bit-equal (same accumulation order).  na = nb drives most candidate lists to exhaustion near the end."""
    A, B = _tables(na, nb, d, na + nb + d, integer)
    idx, dist = matching.match_tables(A, B, n_clusters=25, unique_matching=True)
    ridx, rdist = om.match_tables(A, B, unique_matching=True)
    assert idx == ridx
    assert np.array_equal(np.asarray(dist, dtype=np.float32), np.asarray(rdist, dtype=np.float32))
    assert len(set(idx)) == na


@pytest.mark.parametrize("na,nb,d,integer", [(300, 500, 5, False), (640, 640, 3, True), (9000, 9500, 6, False)])
def test_non_unique_matching_and_index_repair(na, nb, d, integer):
    A, B = _tables(na, nb, d, 7 * na + d, integer)
    idx, dist = matching.match_tables(A, B, unique_matching=False, batch_size=100)
    ridx, rdist = om.match_tables(A, B, unique_matching=False)
    assert idx == ridx and len(set(idx)) == na
    assert np.array_equal(np.asarray(dist, dtype=np.float32), np.asarray(rdist, dtype=np.float32))
    ma, mb = matching.match_rows(A[:200], B[:400])
    assert sorted(ma) == list(range(200)) and len(mb) == 200


def test_matching_limits_and_errors():
    A, B = _tables(10, 5, 3, 1)
    with pytest.raises(ValueError):
        matching.match_tables(A, B, unique_matching=True)              # fewer candidates than queries
    with pytest.raises(_lib.ClavaHipError):
        matching.match_tables(A[:2], B, device="cpu")


def test_duplicate_queries_do_not_stall_the_unique_matching(monkeypatch):
    """when every query is the same row, each search resolves only the 4 candidates one query can hold;
    the batch size must then shrink to one query tile instead of re-searching 8192 queries per 4 matches.  5000 identical
    queries against 6000 rows: exact against the sequential oracle, and the searches stay small."""
    rng = np.random.RandomState(3)
    B = rng.standard_normal((6000, 4))
    A = np.repeat(rng.standard_normal((1, 4)), 5000, axis=0)
    calls = []
    real = matching._topk
    monkeypatch.setattr(matching, "_topk", lambda Ad, Bd, removed: (calls.append(Ad.shape[0]), real(Ad, Bd, removed))[1])
    idx, dist = matching.match_tables(A, B, unique_matching=True)
    ridx, rdist = om.match_tables(A, B, unique_matching=True)
    assert idx == ridx and len(set(idx)) == 5000
    assert np.array_equal(np.asarray(dist, dtype=np.float32), np.asarray(rdist, dtype=np.float32))
    assert len(calls) <= 5000 // 4 + 8
    assert sum(calls) <= 64 * len(calls) + 2 * 8192, (len(calls), sum(calls))     # one query tile per search after the first
