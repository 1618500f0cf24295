"""Multi-parent row matching on the GPU: `match_tables`, `convert_to_unique_indices`, `match_rows`
with the reference's signatures (`pipeline_utils.py:1131-1239`).

What differs, on purpose and visibly: the reference searches a `faiss.IndexIVFFlat` with `n_clusters` lists and the default
nprobe = 1 - an approximate nearest neighbour that depends on faiss's own k-means.  Here the search is EXACT (brute force
on the device, `clava_nn_topk_f32`), so `n_clusters` is accepted and ignored and results can differ from a faiss run where
IVF misses the true neighbour.  The greedy semantics are the reference's: queries in row order, each match removed from
the candidates of the later ones (`unique_matching=True`), or duplicates replaced afterwards by the smallest unused index
(`unique_matching=False`).  Parity against faiss itself is unpinned (faiss is not installed in the build image and the
reference ships no output of this step); the GPU path is exact against the brute-force oracle.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_int, check, lib, ptr, stream_ptr

NK = 4                   # candidates per query returned by clava_nn_topk_f32
QUERY_BATCH = 8192       # queries searched per launch; the greedy pass over a batch runs on the host
MIN_BATCH = 64           # one query tile of the search kernel: the batch size while a run of duplicate queries lasts


def _topk(Ad, Bd, removed):
    L = lib()
    L.clava_nn_topk_workspace_bytes.restype = ctypes.c_longlong
    nq, d = Ad.shape
    nb = Bd.shape[0]
    ws = torch.empty(int(L.clava_nn_topk_workspace_bytes(c_int(nq), c_int(nb))), device=Ad.device, dtype=torch.uint8)
    idx = torch.empty((nq, NK), device=Ad.device, dtype=torch.int32)
    dist = torch.empty((nq, NK), device=Ad.device, dtype=torch.float32)
    check(L.clava_nn_topk_f32(ptr(Ad), ctypes.c_longlong(Ad.stride(0)), c_int(nq), ptr(Bd), ctypes.c_longlong(Bd.stride(0)), c_int(nb),
                              c_int(d), ptr(removed), ptr(ws), ptr(idx), ptr(dist), stream_ptr()), "clava_nn_topk_f32")
    return idx.cpu().numpy(), dist.cpu().numpy()


def convert_to_unique_indices(indices):
    """`pipeline_utils.py:1131-1145`: every repeated index becomes the smallest index of range(len(indices)) not used yet
    (the reference's `min(set)` per duplicate, here one sorted list walked once)."""
    indices = list(indices)
    seen = set()
    free = sorted(set(range(len(indices))) - set(indices))
    pos = 0
    for i, num in enumerate(indices):
        if num in seen:
            indices[i] = free[pos]
            pos += 1
        else:
            seen.add(num)
    return indices


def match_tables(A, B, n_clusters=25, unique_matching=True, batch_size=100, device=None):
    """(indices, distances) as `pipeline_utils.py:1147-1205`: indices[i] = the row of B matched to row i of A, distances[i]
    its squared L2 distance (float32, like faiss's METRIC_L2).  `n_clusters` / `batch_size` belong to the IVF search and
    the GPU-faiss batching of the reference and are ignored."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    if A.ndim != 2 or B.ndim != 2 or A.shape[1] != B.shape[1]:
        raise ValueError("match_tables: A [n, d] and B [m, d] expected")
    if unique_matching and B.shape[0] < A.shape[0]:
        raise ValueError("match_tables: unique matching needs at least as many rows in B as in A")
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise _lib.ClavaHipError("match_tables: the search runs on the GPU (no CPU fallback)")
    Ad, Bd = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    n = A.shape[0]
    indices, distances = [], []
    if not unique_matching:
        for s in range(0, n, QUERY_BATCH):
            idx, dist = _topk(Ad[s:s + QUERY_BATCH], Bd, None)
            indices += idx[:, 0].tolist()
            distances += dist[:, 0].tolist()
        indices = convert_to_unique_indices(indices)
        assert len(indices) == len(set(indices))
        return indices, distances
    removed = torch.zeros(B.shape[0], device=dev, dtype=torch.uint8)
    taken = np.zeros(B.shape[0], dtype=bool)
    start = 0
    batch = QUERY_BATCH
    while start < n:
        stop = min(n, start + batch)
        idx, dist = _topk(Ad[start:stop], Bd, removed)
        done = 0
        newly = []
        for q in range(stop - start):
            # the nearest row still free = the first candidate (ascending distance) no earlier query took; when all NK are
            # gone the list no longer says which row is next: end the batch here and search again with the updated mask
            for k in range(NK):
                j = int(idx[q, k])
                if j >= 0 and not taken[j]:
                    break
            else:
                break
            taken[j] = True
            newly.append(j)
            indices.append(j)
            distances.append(float(dist[q, k]))
            done += 1
        if done == 0:
            raise _lib.ClavaHipError("match_tables: no candidate left for a query (fewer free rows in B than queries)")
        removed[torch.as_tensor(newly, device=dev, dtype=torch.long)] = 1
        start += done
        # A batch that ended early (a query ran out of candidates: near-duplicate queries competing for the same rows) is
        # searched again from that query.  Only ~NK queries resolve per search while such a run lasts, so the next search
        # takes few queries instead of QUERY_BATCH it would mostly throw away; the size doubles again after every batch
        # that completes (the fixed batch made runs of duplicates quadratic).
        batch = max(MIN_BATCH, min(QUERY_BATCH, 2 * done)) if done < stop - (start - done) else min(QUERY_BATCH, 2 * batch)
    return indices, distances


def match_rows(A, B):
    """`pipeline_utils.py:1207-1239`: repeat the non-unique matching, keep the pairs matched exactly once, reconsider the
    rest; returns (matched rows of A, matched rows of B) as lists.  NOTE: like the reference, the second list holds indices
    into the CURRENT (shrunken) B of each round, not original row numbers (`:1226` appends `nearest_neighbors_indices`
    themselves)."""
    A = np.asarray(A)
    B = np.asarray(B)
    original_indices_A = np.arange(A.shape[0])
    matched_A, matched_B = [], []
    while A.shape[0] > 0:
        nn, _ = match_tables(A, B, n_clusters=25, unique_matching=False, batch_size=100)
        nn = np.asarray(nn)
        counts = np.bincount(nn, minlength=B.shape[0])
        unique = counts[nn] == 1
        matched_A.extend(original_indices_A[unique])
        matched_B.extend(nn[unique])
        reconsider_B = np.unique(nn[~unique])
        A = A[~unique]
        original_indices_A = original_indices_A[~unique]
        B = B[reconsider_B]
    return matched_A, matched_B
