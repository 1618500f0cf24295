// Exact nearest-neighbour candidates for the multi-parent row matching.
//
// Reference: match_tables (pipeline_utils.py:1147-1205) searches a faiss.IndexIVFFlat (25 lists, nprobe = 1: approximate,
// dependent on faiss's own k-means) one query at a time and removes every match from the index.  faiss is a third-party
// library that is not part of the reference tree; this is an EXACT search with the same greedy semantics (parity unpinned
// against faiss IVF, exact against the brute-force oracle oracle/matching.py).
//
// clava_nn_topk_f32: for every query row the NK = 4 nearest rows of B that are not flagged in `removed`, squared L2
// distance accumulated column by column in float32 (the oracle's order), ties to the lower row index.  The host walks the
// queries in order and gives each its first candidate not taken by an earlier query (clavaddpm_amd/matching.py): the
// nearest row among those left IS the first surviving entry of the sorted candidate list, so the greedy result is exact
// as long as a list is not exhausted; where it is, the host ends the batch there and searches again.
//   kernel 1: grid (B chunks, query tiles of 64), 256 threads = 64 queries x 4 row slices; the tile of queries and 128-row
//   tiles of B go through LDS; every thread keeps a sorted top-4 in registers;
//   kernel 2: one thread per query merges the 4 x chunks partial lists.
#include "common.h"
#include "clava_hip.h"

namespace {

constexpr int NK = 4, QT = 64, BT = 128, DMAX = 64;

struct Top4 {
    float d[NK];
    int i[NK];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < NK; ++k) { d[k] = INFINITY; i[k] = 0x7fffffff; }
    }
    // insert (dist, idx) keeping (dist, idx) lexicographic order
    __device__ __forceinline__ void push(float dist, int idx) {
        if (!(dist < d[NK - 1] || (dist == d[NK - 1] && idx < i[NK - 1]))) return;
        d[NK - 1] = dist; i[NK - 1] = idx;
#pragma unroll
        for (int k = NK - 1; k > 0; --k) {
            const bool sw = d[k] < d[k - 1] || (d[k] == d[k - 1] && i[k] < i[k - 1]);
            const float td = sw ? d[k - 1] : d[k]; const int ti = sw ? i[k - 1] : i[k];
            d[k - 1] = sw ? d[k] : d[k - 1]; i[k - 1] = sw ? i[k] : i[k - 1];
            d[k] = td; i[k] = ti;
        }
    }
};

__global__ void __launch_bounds__(256) nn_partial_kernel(const float* __restrict__ A, long long lda, int nq, const float* __restrict__ B,
                                                         long long ldb, int nb, int d, const unsigned char* __restrict__ removed,
                                                         int rows_per_chunk, float* __restrict__ pd, int* __restrict__ pi) {
    __shared__ float sq[QT][DMAX + 1];
    __shared__ float sb[BT][DMAX + 1];
    const int q0 = blockIdx.y * QT, chunk = blockIdx.x, n_lists = gridDim.x * 4;
    const int qi = threadIdx.x & (QT - 1), slice = threadIdx.x >> 6;
    // tables wider than DMAX columns go through LDS in column chunks; a (query, row) pair's running sum stays in a register
    // across the chunks, so the float32 accumulation order (column 0, 1, 2, ...) is the same for every width
    const bool one_chunk = d <= DMAX;
    auto load_queries = [&](int c0, int dc) {
        for (int e = threadIdx.x; e < QT * dc; e += 256) {
            const int r = e / dc, c = e - r * dc;
            sq[r][c] = (q0 + r < nq) ? A[(long long)(q0 + r) * lda + c0 + c] : 0.f;
        }
    };
    if (one_chunk) load_queries(0, d);
    Top4 top;
    top.init();
    const int b_begin = chunk * rows_per_chunk, b_end = min(nb, b_begin + rows_per_chunk);
    for (int t0 = b_begin; t0 < b_end; t0 += BT) {
        float acc[BT / 4];
#pragma unroll
        for (int j = 0; j < BT / 4; ++j) acc[j] = 0.f;
        for (int c0 = 0; c0 < d; c0 += DMAX) {
            const int dc = min(DMAX, d - c0);
            __syncthreads();
            if (!one_chunk) load_queries(c0, dc);
            for (int e = threadIdx.x; e < BT * dc; e += 256) {
                const int r = e / dc, c = e - r * dc;
                sb[r][c] = (t0 + r < b_end) ? B[(long long)(t0 + r) * ldb + c0 + c] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < BT / 4; ++j) {
                const int r = slice + 4 * j;
                float a = acc[j];
                for (int c = 0; c < dc; ++c) {
                    const float df = sb[r][c] - sq[qi][c];
                    a += df * df;
                }
                acc[j] = a;
            }
        }
#pragma unroll
        for (int j = 0; j < BT / 4; ++j) {
            const int row = t0 + slice + 4 * j;
            if (row < b_end && !(removed && removed[row])) top.push(acc[j], row);
        }
    }
    if (q0 + qi < nq) {
        const long long o = ((long long)(q0 + qi) * n_lists + chunk * 4 + slice) * NK;
#pragma unroll
        for (int k = 0; k < NK; ++k) { pd[o + k] = top.d[k]; pi[o + k] = top.i[k]; }
    }
}

__global__ void __launch_bounds__(256) nn_merge_kernel(const float* __restrict__ pd, const int* __restrict__ pi, int nq, int n_lists,
                                                       float* __restrict__ out_d, int* __restrict__ out_i) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    Top4 top;
    top.init();
    const long long o = (long long)q * n_lists * NK;
    for (int e = 0; e < n_lists * NK; ++e) {
        const int idx = pi[o + e];
        if (idx != 0x7fffffff) top.push(pd[o + e], idx);
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        out_d[(long long)q * NK + k] = top.d[k];
        out_i[(long long)q * NK + k] = (top.i[k] == 0x7fffffff) ? -1 : top.i[k];
    }
}

int chunks_for(int nq, int nb) {
    const int q_tiles = (nq + QT - 1) / QT;
    int chunks = (1024 + q_tiles - 1) / q_tiles;                 // ~4 workgroups per CU in total
    const int max_chunks = (nb + BT - 1) / BT;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > 256) chunks = 256;
    return chunks < 1 ? 1 : chunks;
}

}  // namespace

extern "C" {

long long clava_nn_topk_workspace_bytes(int nq, int nb) {
    return (long long)nq * chunks_for(nq, nb) * 4 * NK * 8;
}

int clava_nn_topk_f32(const float* A, long long lda, int nq, const float* B, long long ldb, int nb, int d,
                      const unsigned char* removed, void* workspace, int* out_idx, float* out_dist, void* stream) {
    CLAVA_CHECK_ARG(A && B && out_idx && out_dist && workspace, "nn_topk: null argument");
    CLAVA_CHECK_ARG(nq > 0 && nb > 0 && d > 0 && lda >= d && ldb >= d, "nn_topk: bad shape nq=%d nb=%d d=%d", nq, nb, d);
    hipStream_t st = (hipStream_t)stream;
    const int chunks = chunks_for(nq, nb);
    const int per = ((nb + chunks - 1) / chunks + BT - 1) / BT * BT;
    float* pd = (float*)workspace;
    int* pi = (int*)(pd + (long long)nq * chunks * 4 * NK);
    ProfScope prof(PROF_MISC, 3.0 * nq * (double)nb * d, 4.0 * ((double)nq * d + (double)nb * d), st);
    hipLaunchKernelGGL(nn_partial_kernel, dim3(chunks, (nq + QT - 1) / QT), dim3(256), 0, st, A, lda, nq, B, ldb, nb, d, removed, per, pd, pi);
    CLAVA_CHECK_LAUNCH("clava_nn_topk_f32(partial)");
    hipLaunchKernelGGL(nn_merge_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, pd, pi, nq, chunks * 4, out_dist, out_idx);
    CLAVA_CHECK_LAUNCH("clava_nn_topk_f32(merge)");
    return CLAVA_OK;
}

}  // extern "C"
