"""Multi-parent row matching (oracle; test infrastructure only).  PARITY UNPINNED against the reference's search.

`match_tables` (`pipeline_utils.py:1147-1205`) finds, for every row of A in order, a near row of B and - with
`unique_matching` - removes it from the index before the next query; without it, duplicates are replaced afterwards by
`convert_to_unique_indices` (`:1131-1145`).  The reference searches a `faiss.IndexIVFFlat` (n_clusters lists, default
nprobe = 1): approximate, and dependent on faiss's k-means of the coarse quantiser.  faiss (third-party, un-vendored,
`requirements.txt`) is not installed here and the reference holds no output of this step, so nothing pins the search
itself; this oracle restates the loop logic of the reference with an EXACT brute-force search (squared L2 in float32,
columns accumulated in order, first minimum) - what IVF approximates.  `convert_to_unique_indices` and `match_rows` are
plain Python / numpy in the reference and are restated statement by statement.
"""
import numpy as np


def sqdist_rows(a, B):
    """Squared L2 distance of one row to every row of B, float32, columns accumulated in order."""
    acc = np.zeros(B.shape[0], dtype=np.float32)
    for c in range(B.shape[1]):
        diff = B[:, c] - a[c]
        acc += diff * diff
    return acc


def convert_to_unique_indices(indices):
    """`:1131-1145`: every repeated index is replaced by the smallest index in range(len(indices)) not used so far."""
    indices = list(indices)
    occurrence = set()
    candidates = sorted(set(range(len(indices))) - set(indices))
    pos = 0
    for i, num in enumerate(indices):
        if num in occurrence:
            indices[i] = candidates[pos]
            pos += 1
        else:
            occurrence.add(num)
    return indices


def match_tables(A, B, unique_matching=True):
    """Returns (indices, squared distances) like the reference (`:1147-1205`), exact search."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    removed = np.zeros(B.shape[0], dtype=bool)
    idx, dist = [], []
    for a in A:
        dd = sqdist_rows(a, B)
        if unique_matching:
            dd = np.where(removed, np.float32(np.inf), dd)
        j = int(np.argmin(dd))                      # first minimum = lowest row index among ties
        idx.append(j)
        dist.append(float(dd[j]))
        if unique_matching:
            removed[j] = True
    if not unique_matching:
        idx = convert_to_unique_indices(idx)
        assert len(idx) == len(set(idx))
    return idx, dist
