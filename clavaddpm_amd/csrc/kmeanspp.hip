// k-means++ seeding of the clustering stage on the device: the `kmeans_plusplus` call behind
// GaussianMixture(init_params='k-means++') (pipeline_modules.py:269-277 -> sklearn/mixture/_base.py
// `_initialize_parameters` -> sklearn/cluster/_kmeans.py `_kmeans_plusplus`).  The random draws stay on the host (NumPy's
// generator, so a seeded run picks the same centres as scikit-learn); everything of size N runs here, in float64:
//
//   row_norms      xsq[n] = sum_j X[n,j]^2                                                     (row_norms(X, squared=True))
//   candidates     for T candidate rows c_t:  dist = max(0, (-2 X[n].X[c_t] + xsq[c_t]) + xsq[n])   (_euclidean_distances)
//                  newd[t][n] = min(closest[n], dist);  pot[t] = sum_n newd[t][n]              (np.minimum, `@ sample_weight`)
//   pick           ids[t] = searchsorted(cumsum(closest), rand[t] * sum(closest)), clipped to N-1  (stable_cumsum, np.clip)
//
// All three are HBM-bound streams (candidates: 8*N*(d + 2 + T) bytes per new centre).  A row is read by 16 lanes (two
// doubles each per 256-byte step: whole 128-byte lines per row), the T dot products are reduced over those lanes with
// shuffles; sums over N go through per-workgroup partials added in a fixed order.
#include "common.h"
#include "clava_hip.h"

namespace {

constexpr int KMPP_MAX_T = 16;
constexpr int KMPP_BLOCKS = 512;     // partial sums per candidate
constexpr int PICK_BLOCKS = 1024;    // block sums of the cumulative distribution

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) row_norms_kernel(const double* __restrict__ X, long long ldx, long long N, int d,
                                                        double* __restrict__ xsq) {
    const int l16 = threadIdx.x & 15;
    for (long long n = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); n < N; n += (long long)gridDim.x * 16) {
        const double* row = X + (size_t)n * ldx;
        double acc = 0.0;
        for (int j = l16; j < d; j += 16) acc += row[j] * row[j];
        for (int m = 8; m > 0; m >>= 1) acc += shfl_xor_f64(acc, m);
        if (l16 == 0) xsq[n] = acc;
    }
}

struct CandArgs {
    const double* X; long long ldx; long long N; int d;
    const double* xsq;
    const double* closest;          // [N] or null (first centre)
    const long long* cand;          // [T] row indices (device)
    int T;
    double* newd;                   // [T][N]
    double* part;                   // [T][gridDim.x]
};

template <int T>
__global__ void __launch_bounds__(256) kmpp_candidates_kernel(CandArgs g) {
    extern __shared__ double C[];                 // [T][d] candidate rows, then [T] their squared norms
    double* csq = C + (size_t)T * g.d;
    __shared__ double dots[KMPP_MAX_T][16];
    for (int i = threadIdx.x; i < T * g.d; i += 256) {
        const int t = i / g.d, j = i - t * g.d;
        C[i] = g.X[(size_t)g.cand[t] * g.ldx + j];
    }
    if (threadIdx.x < T) csq[threadIdx.x] = g.xsq[g.cand[threadIdx.x]];
    __syncthreads();
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int wt = threadIdx.x >> 4, wi = threadIdx.x & 15;   // writer role: candidate wt (< T), row n0 + wi
    double pot = 0.0;
    // a workgroup walks blocks of 16 consecutive rows: 16 lanes per row for the dot products, then thread (t, i) finishes
    // candidate t of row i so that every new_closest[t] is written in whole 128-byte runs
    for (long long n0 = (long long)blockIdx.x * 16; n0 < g.N; n0 += (long long)gridDim.x * 16) {
        const long long n = n0 + grp;
        double acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = 0.0;
        if (n < g.N) {
            const double* row = g.X + (size_t)n * g.ldx;
            for (int j = l16; j < g.d; j += 16) {
                const double x = row[j];
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] += x * C[t * g.d + j];
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
            for (int m = 8; m > 0; m >>= 1) acc[t] += shfl_xor_f64(acc[t], m);
        double mine = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) mine = l16 == t ? acc[t] : mine;
        if (l16 < T) dots[l16][grp] = mine;
        __syncthreads();
        if (wt < T && n0 + wi < g.N) {
            const long long m = n0 + wi;
            double dist = (-2.0 * dots[wt][wi] + csq[wt]) + g.xsq[m];
            dist = fmax(dist, 0.0);
            const double nd = g.closest ? fmin(g.closest[m], dist) : dist;
            g.newd[(size_t)wt * g.N + m] = nd;
            pot += nd;
        }
        __syncthreads();
    }
    if (wt < T) dots[wt][wi] = pot;
    __syncthreads();
    if (threadIdx.x < T) {
        double s = 0.0;
        for (int i = 0; i < 16; ++i) s += dots[threadIdx.x][i];
        g.part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}

// pot[t] = sum_b part[t][b] in a fixed order (one wavefront per candidate)
__global__ void __launch_bounds__(64) kmpp_pot_kernel(const double* __restrict__ part, int nb, double* __restrict__ pot) {
    const int t = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) s += part[(size_t)t * nb + b];
    for (int m = 32; m > 0; m >>= 1) s += shfl_xor_f64(s, m);
    if (threadIdx.x == 0) pot[t] = s;
}

// block sums of `closest` over contiguous ranges of `per` elements
__global__ void __launch_bounds__(256) pick_blocksum_kernel(const double* __restrict__ v, long long N, long long per,
                                                            double* __restrict__ bsum) {
    __shared__ double sh[256];
    const long long a = (long long)blockIdx.x * per, b = min(N, a + per);
    double s = 0.0;
    for (long long i = a + threadIdx.x; i < b; i += 256) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}

// one workgroup; wavefront t (< T) finds ids[t] = first i with cumsum(v)[i] >= rand[t] * total, clipped to N - 1
__global__ void __launch_bounds__(1024) pick_search_kernel(const double* __restrict__ v, long long N, long long per, int nb,
                                                           const double* __restrict__ bsum, const double* __restrict__ rand01,
                                                           int T, long long* __restrict__ ids, double* __restrict__ total_out) {
    __shared__ double prefix[PICK_BLOCKS + 1];     // exclusive prefix of the block sums
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) { prefix[b] = s; s += bsum[b]; }
        prefix[nb] = s;
        if (total_out) total_out[0] = s;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= T) return;
    const double target = rand01[wave] * prefix[nb];
    int b = 0;                                      // first block whose inclusive prefix reaches the target
    while (b < nb - 1 && prefix[b + 1] < target) ++b;
    const long long a = (long long)b * per, e = N;   // normally ends inside block b; rounding may push it a little further
    double base = prefix[b];
    long long found = -1;
    for (long long i0 = a; i0 < e && found < 0; i0 += 64) {
        const long long i = i0 + lane;
        double x = i < e ? v[i] : 0.0;
        for (int m = 1; m < 64; m <<= 1) {          // inclusive scan over the wavefront
            int lo = __double2loint(x), hi = __double2hiint(x);
            lo = __shfl_up(lo, m);
            hi = __shfl_up(hi, m);
            const double y = __hiloint2double(hi, lo);
            if (lane >= m) x += y;
        }
        const double cum = base + x;
        const unsigned long long hit = __ballot(i < e && cum >= target);
        if (hit) found = i0 + (__ffsll((long long)hit) - 1);
        base += __hiloint2double(__shfl(__double2hiint(x), 63), __shfl(__double2loint(x), 63));   // the chunk's total
    }
    if (found < 0) found = N - 1;                   // np.clip(candidate_ids, None, n - 1)
    if (lane == 0) ids[wave] = found;
}

template <int T>
void launch_candidates(const CandArgs& g, int blocks, hipStream_t st) {
    const size_t lds = ((size_t)T * g.d + T) * sizeof(double);
    hipLaunchKernelGGL((kmpp_candidates_kernel<T>), dim3(blocks), dim3(256), lds, st, g);
}

}  // namespace

extern "C" {

int clava_kmpp_row_norms_f64(const double* X, long long ldx, long long N, int d, double* xsq, void* stream) {
    CLAVA_CHECK_ARG(X && xsq && N > 0 && d > 0 && ldx >= d, "kmpp_row_norms: bad arguments");
    const long long blocks = (N + 15) / 16;
    ProfScope prof(PROF_GMM, 2.0 * N * d, 8.0 * N * (d + 1), (hipStream_t)stream);
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, X, ldx, N, d, xsq);
    CLAVA_CHECK_LAUNCH("clava_kmpp_row_norms_f64");
    return CLAVA_OK;
}

long long clava_kmpp_workspace_doubles(void) { return (long long)KMPP_MAX_T * KMPP_BLOCKS + PICK_BLOCKS; }

int clava_kmpp_candidates_f64(const double* X, long long ldx, long long N, int d, const double* xsq, const double* closest,
                              const long long* candidates, int T, double* workspace, double* new_closest, double* potentials,
                              void* stream) {
    CLAVA_CHECK_ARG(X && xsq && candidates && workspace && new_closest && potentials && N > 0 && d > 0 && ldx >= d,
                    "kmpp_candidates: bad arguments");
    if (T < 1 || T > KMPP_MAX_T) {
        clava_set_error("kmpp_candidates: %d candidates, supported 1..%d", T, KMPP_MAX_T);
        return CLAVA_ERR_LIMIT;
    }
    if (((size_t)T * d + T) * sizeof(double) > 60000) {
        clava_set_error("kmpp_candidates: %d candidates x %d features exceed the staging buffer", T, d);
        return CLAVA_ERR_LIMIT;
    }
    hipStream_t st = (hipStream_t)stream;
    CandArgs g;
    g.X = X; g.ldx = ldx; g.N = N; g.d = d; g.xsq = xsq; g.closest = closest; g.cand = candidates; g.T = T; g.newd = new_closest;
    g.part = workspace;
    const long long want = (N + 15) / 16;
    const int blocks = (int)(want < KMPP_BLOCKS ? want : KMPP_BLOCKS);
    {
        ProfScope prof(PROF_GMM, 2.0 * N * d * T, 8.0 * N * (d + 2 + T), st);
        switch (T) {
#define CLAVA_T(n) case n: launch_candidates<n>(g, blocks, st); break;
            CLAVA_T(1) CLAVA_T(2) CLAVA_T(3) CLAVA_T(4) CLAVA_T(5) CLAVA_T(6) CLAVA_T(7) CLAVA_T(8)
            CLAVA_T(9) CLAVA_T(10) CLAVA_T(11) CLAVA_T(12) CLAVA_T(13) CLAVA_T(14) CLAVA_T(15) CLAVA_T(16)
#undef CLAVA_T
        }
        CLAVA_CHECK_LAUNCH("clava_kmpp_candidates_f64");
    }
    hipLaunchKernelGGL(kmpp_pot_kernel, dim3(T), dim3(64), 0, st, workspace, blocks, potentials);
    CLAVA_CHECK_LAUNCH("clava_kmpp_candidates_f64(potentials)");
    return CLAVA_OK;
}

int clava_kmpp_pick_f64(const double* closest, long long N, const double* rand01, int T, double* workspace, long long* ids,
                        double* total, void* stream) {
    CLAVA_CHECK_ARG(closest && rand01 && workspace && ids && N > 0, "kmpp_pick: bad arguments");
    if (T < 1 || T > KMPP_MAX_T) {
        clava_set_error("kmpp_pick: %d draws, supported 1..%d", T, KMPP_MAX_T);
        return CLAVA_ERR_LIMIT;
    }
    hipStream_t st = (hipStream_t)stream;
    long long per = (N + PICK_BLOCKS - 1) / PICK_BLOCKS;
    per = (per + 255) / 256 * 256;
    const int nb = (int)((N + per - 1) / per);
    double* bsum = workspace + (size_t)KMPP_MAX_T * KMPP_BLOCKS;     // behind the candidate partials
    ProfScope prof(PROF_GMM, 0.0, 8.0 * N, st);
    hipLaunchKernelGGL(pick_blocksum_kernel, dim3(nb), dim3(256), 0, st, closest, N, per, bsum);
    CLAVA_CHECK_LAUNCH("clava_kmpp_pick_f64(block sums)");
    hipLaunchKernelGGL(pick_search_kernel, dim3(1), dim3(1024), 0, st, closest, N, per, nb, bsum, rand01, T, ids, total);
    CLAVA_CHECK_LAUNCH("clava_kmpp_pick_f64");
    return CLAVA_OK;
}

}  // extern "C"
