// Diagonal-covariance Gaussian-mixture E-step in float64 on the gfx950 matrix cores.
//
// Replaces the arithmetic behind `GaussianMixture(covariance_type='diag').fit(X).predict(X)` at
// pipeline_modules.py:268-277, i.e. scikit-learn's `_estimate_log_gaussian_prob` ('diag' branch: two N x d x k
// float64 products), `_estimate_weighted_log_prob`, `_estimate_log_prob_resp` (log-sum-exp over components) and
// `predict` (arg-max), called once per EM iteration (<= 100) plus the final E-step and predict.
//
//   lp[n,c]  = sum_j mu^2 P - 2 * X[n,:] . (mu P)[c,:] + X[n,:]^2 . P[c,:]            (P = prec_chol^2)
//   wlp[n,c] = -0.5 * (d ln 2pi + lp) + sum_j ln prec_chol[c,j] + ln w[c]
//   log_prob_norm[n] = logsumexp_c wlp ;  log_resp = wlp - log_prob_norm ;  label = argmax_c wlp
//
// Structure: a 256-thread workgroup owns 64 consecutive rows (one 16-row MFMA tile per wave); the feature axis is
// walked in chunks of 32 columns: the X chunk (coalesced 16-B loads of whole row segments) and the two k x 32 table
// chunks are staged in LDS, then v_mfma_f64_16x16x4_f64 accumulates both products for up to four 16-component column
// tiles (k <= 64).  X is read from HBM exactly once; the kernel is HBM-bound for k*d small and f64-MFMA-bound
// otherwise.  The two products keep separate accumulators and are combined in scikit-learn's order.
#include "common.h"
#include <mutex>
#include "clava_hip.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DC = 32;           // feature columns per chunk
constexpr int LDT = DC + 1;      // LDS row stride in doubles
constexpr int ROWS = 64;         // rows per workgroup
constexpr int KMAX = 64;         // components supported

struct GmmArgs {
    const double* X; long long ldx; long long N; int d, k;
    const double* mp;        // [k, d] means * precisions
    const double* P;         // [k, d] precisions = prec_chol^2
    const double* s_mu2p;    // [k] sum_j means^2 * precisions
    const double* logdet;    // [k] sum_j log prec_chol
    const double* logw;      // [k] log weights
    double dln2pi;           // d * log(2 pi)
    double* log_prob_norm;   // [N]
    double* log_resp;        // [N, k] or null
    int* labels;             // [N] or null
};

// tables from the mixture parameters (k*d elements: negligible work, one workgroup per component)
__global__ void __launch_bounds__(256) gmm_prep_kernel(const double* __restrict__ means, const double* __restrict__ pc,
                                                       const double* __restrict__ w, int k, int d, double* __restrict__ mp,
                                                       double* __restrict__ P, double* __restrict__ s_mu2p,
                                                       double* __restrict__ logdet, double* __restrict__ logw) {
    const int c = blockIdx.x;
    __shared__ double part[2][256];
    double a = 0.0, b = 0.0;
    for (int j = threadIdx.x; j < d; j += 256) {
        const double m = means[(size_t)c * d + j], q = pc[(size_t)c * d + j];
        const double p = q * q;
        P[(size_t)c * d + j] = p;
        mp[(size_t)c * d + j] = m * p;
        a += (m * m) * p;
        b += log(q);
    }
    part[0][threadIdx.x] = a; part[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { part[0][threadIdx.x] += part[0][threadIdx.x + s]; part[1][threadIdx.x] += part[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { s_mu2p[c] = part[0][0]; logdet[c] = part[1][0]; logw[c] = log(w[c]); }
}

// exp(x) for x <= 0 in the log-sum-exp of the responsibilities: round-to-nearest range reduction x = n ln2 + r,
// |r| <= ln2 / 2, degree-11 Taylor polynomial in Horner form (truncation 0.347^12 / 12! = 6e-15 relative), exact scaling
// by 2^n.  Branch-free, ~17 float64 instructions; the device library's exp() is ~3x that with branches, and with
// 16 * ceil(k/16) of them per row the exponentials, not the matrix products, would set the kernel's speed.
__device__ __forceinline__ double exp_nonpos(double x) {
    x = fmax(x, -745.0);                                     // below this exp underflows to 0 anyway; keeps n in int range
    const double n = rint(x * 1.4426950408889634);
    double r = fma(n, -6.93147180369123816490e-01, x);       // ln2 split hi / lo
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;                   // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);               // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);               // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);               // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);               // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);               // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);               // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);               // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);               // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);               // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

template <int CT>   // number of 16-component column tiles (1..4)
__global__ void __launch_bounds__(256) gmm_estep_kernel(GmmArgs g) {
    __shared__ double Xs[ROWS * LDT];
    __shared__ double Ms[CT * 16 * LDT];
    __shared__ double Ps[CT * 16 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (long long row0 = (long long)blockIdx.x * ROWS; row0 < g.N; row0 += (long long)gridDim.x * ROWS) {
        f64x4 acc1[CT], acc2[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { acc1[ct] = {0.0, 0.0, 0.0, 0.0}; acc2[ct] = {0.0, 0.0, 0.0, 0.0}; }
        for (int j0 = 0; j0 < g.d; j0 += DC) {
            __syncthreads();     // previous chunk fully consumed
            // stage the X chunk: thread -> (row, column pair); 16 threads cover one 32-column row segment
            for (int s = tid; s < ROWS * (DC / 2); s += 256) {
                const int rr = s / (DC / 2), cc = (s % (DC / 2)) * 2;
                const long long row = row0 + rr;
                double v0 = 0.0, v1 = 0.0;
                if (row < g.N) {
                    const double* p = g.X + row * g.ldx + j0 + cc;
                    if (j0 + cc < g.d) v0 = p[0];
                    if (j0 + cc + 1 < g.d) v1 = p[1];
                }
                Xs[rr * LDT + cc] = v0;
                Xs[rr * LDT + cc + 1] = v1;
            }
            for (int s = tid; s < CT * 16 * DC; s += 256) {
                const int c = s / DC, cc = s % DC;
                const bool ok = c < g.k && j0 + cc < g.d;
                Ms[c * LDT + cc] = ok ? g.mp[(size_t)c * g.d + j0 + cc] : 0.0;
                Ps[c * LDT + cc] = ok ? g.P[(size_t)c * g.d + j0 + cc] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < DC / 4; ++ks) {
                // A operand of v_mfma_f64_16x16x4_f64: lane holds A[row = lane & 15][k = lane >> 4]
                const double a1 = Xs[(wave * 16 + r) * LDT + ks * 4 + q];
                const double a2 = a1 * a1;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const double b1 = Ms[(ct * 16 + r) * LDT + ks * 4 + q];   // B[k = lane >> 4][col = lane & 15]
                    const double b2 = Ps[(ct * 16 + r) * LDT + ks * 4 + q];
                    acc1[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1[ct], 0, 0, 0);
                    acc2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc2[ct], 0, 0, 0);
                }
            }
        }
        // epilogue.  f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const long long row = row0 + wave * 16 + q + 4 * reg;
            double wlp[CT];
            double mx = -INFINITY;
            int arg = 0x7fffffff;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 16 + r;
                if (c < g.k) {
                    const double lp = (g.s_mu2p[c] - 2.0 * acc1[ct][reg]) + acc2[ct][reg];
                    wlp[ct] = ((-0.5 * (g.dln2pi + lp)) + g.logdet[c]) + g.logw[c];
                } else {
                    wlp[ct] = -INFINITY;
                }
                if (wlp[ct] > mx) { mx = wlp[ct]; arg = c; }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {       // the 16 lanes of a row group hold the row's components
                const double om = __shfl_xor(mx, o, 64);
                const int oa = __shfl_xor(arg, o, 64);
                if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
            }
            double sm = 0.0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) sm += (ct * 16 + r < g.k) ? exp_nonpos(wlp[ct] - mx) : 0.0;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
            const double lse = mx + log(sm);
            if (row < g.N) {
                if (r == 0) {
                    g.log_prob_norm[row] = lse;
                    if (g.labels) g.labels[row] = arg;
                }
                if (g.log_resp) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int c = ct * 16 + r;
                        if (c < g.k) g.log_resp[row * g.k + c] = wlp[ct] - lse;
                    }
                }
            }
        }
    }
}


// ---- resident-table kernel (the fast path) ---------------------------------------------------------------------------
// Both k x d tables stay in LDS for the whole launch (staged once per workgroup, workgroups are persistent), each wave
// owns 16-row tiles on its own (no workgroup barrier in the row loop) and loads its X operand straight from HBM into
// MFMA registers: the contraction slot (ks, q) of a 32-column chunk is mapped to column 8*q + ks, so lane (row r,
// q = lane >> 4) owns the 64 contiguous bytes X[r][j0 + 8q .. 8q + 7] (four 16-byte loads; the four q lanes of a row
// cover its whole 256-byte segment).  The tables are staged in the matching order, piece-major
// [chunk][tile][table][piece i][lane][2 doubles] so that a lane's B operands are conflict-free ds_read_b128s.
// With X read once and nothing else streamed, the kernel is bound by the f64 matrix pipe for k > ~16 components
// (4*N*d*16*ceil(k/16) FLOP) and by HBM below that.
template <int CT, int NCH>   // 16-component column tiles (1..4), 32-column chunks held in registers per row tile (1, 2, 4)
__global__ void __launch_bounds__(256) gmm_estep_resident_kernel(GmmArgs g, int vec_ok) {
    extern __shared__ __attribute__((aligned(16))) double tab[];      // NCH * CT * 2 * 4 * 64 * 2 doubles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    // ---- stage the tables: element (chunk, ct, table, piece i, lane l, e) = table[ct*16 + (l & 15)][chunk*32 + 8*(l >> 4) + 2*i + e]
    constexpr int TOTAL = NCH * CT * 2 * 4 * 64 * 2;
    for (int s = tid; s < TOTAL; s += 256) {
        const int e = s & 1, l = (s >> 1) & 63, i = (s >> 7) & 3, tb = (s >> 9) & 1;
        const int ct = (s >> 10) % CT, ch = (s >> 10) / CT;
        const int c = ct * 16 + (l & 15), col = ch * 32 + 8 * (l >> 4) + 2 * i + e;
        double v = 0.0;
        if (c < g.k && col < g.d) v = (tb ? g.P : g.mp)[(size_t)c * g.d + col];
        tab[s] = v;
    }
    // per-component constants of this lane's columns, kept in registers: a load inside the row loop would queue behind
    // the previous tile's log_resp stores on the in-order memory counter and expose their HBM latency once per use
    double k_mu2p[CT], k_logdet[CT], k_logw[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = min(ct * 16 + r, g.k - 1);
        k_mu2p[ct] = g.s_mu2p[c]; k_logdet[ct] = g.logdet[c]; k_logw[ct] = g.logw[c];
    }
    __syncthreads();

    const long long n_tiles = (g.N + 15) / 16;
    const long long stride = (long long)gridDim.x * 4;
    double xa[NCH][8], xb[NCH][8];
    auto load_tile = [&](long long tile, double (&dst)[NCH][8]) {
        const long long row_ld = min(tile * 16 + r, g.N - 1);
        const double* xrow = g.X + row_ld * g.ldx + 8 * q;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c0 = ch * 32 + 8 * q;
            if (vec_ok && c0 + 8 <= g.d) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double2 v = *reinterpret_cast<const double2*>(xrow + ch * 32 + 2 * i);
                    dst[ch][2 * i] = v.x; dst[ch][2 * i + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) dst[ch][i] = (c0 + i < g.d) ? xrow[ch * 32 + i] : 0.0;
            }
        }
    };
    long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile < n_tiles) load_tile(tile, xa);
    for (; tile < n_tiles; tile += stride) {
        // the next tile's X is requested before this tile's matrix products and has landed by the time this tile's stores
        // are issued: the wait for it (below) never includes store completions, and the stores drain under the next
        // tile's products
        load_tile(min(tile + stride, n_tiles - 1), xb);
        f64x4 acc1[CT], acc2[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { acc1[ct] = {0.0, 0.0, 0.0, 0.0}; acc2[ct] = {0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const double* tch = tab + (size_t)ch * CT * 2 * 4 * 64 * 2 + lane * 2;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                double b1[8], b2[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double2 m = *reinterpret_cast<const double2*>(tch + ((ct * 2 + 0) * 4 + i) * 128);
                    const double2 p = *reinterpret_cast<const double2*>(tch + ((ct * 2 + 1) * 4 + i) * 128);
                    b1[2 * i] = m.x; b1[2 * i + 1] = m.y;
                    b2[2 * i] = p.x; b2[2 * i + 1] = p.y;
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    acc1[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ch][ks], b1[ks], acc1[ct], 0, 0, 0);
                    acc2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ch][ks] * xa[ch][ks], b2[ks], acc2[ct], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // xb has landed; nothing else is in flight but old stores
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int i = 0; i < 8; ++i) xa[ch][i] = xb[ch][i];
        // epilogue.  f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const long long row = tile * 16 + q + 4 * reg;
            double wlp[CT];
            double mx = -INFINITY;
            int arg = 0x7fffffff;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 16 + r;
                if (c < g.k) {
                    const double lp = (k_mu2p[ct] - 2.0 * acc1[ct][reg]) + acc2[ct][reg];
                    wlp[ct] = ((-0.5 * (g.dln2pi + lp)) + k_logdet[ct]) + k_logw[ct];
                } else {
                    wlp[ct] = -INFINITY;
                }
                if (wlp[ct] > mx) { mx = wlp[ct]; arg = c; }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const double om = __shfl_xor(mx, o, 64);
                const int oa = __shfl_xor(arg, o, 64);
                if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
            }
            double sm = 0.0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) sm += (ct * 16 + r < g.k) ? exp_nonpos(wlp[ct] - mx) : 0.0;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
            const double lse = mx + log(sm);
            if (row < g.N) {
                if (r == 0) {
                    g.log_prob_norm[row] = lse;
                    if (g.labels) g.labels[row] = arg;
                }
                if (g.log_resp) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int c = ct * 16 + r;
                        if (c < g.k) g.log_resp[row * g.k + c] = wlp[ct] - lse;
                    }
                }
            }
        }
    }
}

template <int CT, int NCH>
int launch_resident(const GmmArgs& g, hipStream_t st) {
    static bool attr_set[64] = {};   // per device
    constexpr int lds_bytes = NCH * CT * 2 * 4 * 64 * 2 * 8;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_estep_resident_kernel<CT, NCH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        attr_set[dev] = true;
    }
    const long long n_tiles = (g.N + 15) / 16;
    const int per_cu = lds_bytes <= 40 * 1024 ? 4 : (lds_bytes <= 80 * 1024 ? 2 : 1);     // co-resident workgroups per CU
    long long grid = (n_tiles + 3) / 4;
    if (grid > 256LL * per_cu) grid = 256LL * per_cu;
    const int vec_ok = ((g.ldx & 1) == 0) && ((reinterpret_cast<uintptr_t>(g.X) & 15) == 0);
    hipLaunchKernelGGL((gmm_estep_resident_kernel<CT, NCH>), dim3((unsigned)grid), dim3(256), lds_bytes, st, g, vec_ok);
    return 0;
}

template <int CT>
int launch_resident_ct(const GmmArgs& g, int n_chunks, hipStream_t st) {
    if (n_chunks <= 1) return launch_resident<CT, 1>(g, st);
    if (n_chunks <= 2) return launch_resident<CT, 2>(g, st);
    return launch_resident<CT, 4>(g, st);
}

// ---------------------------------------------------------------------------------------------------- M-step
// scikit-learn's `_m_step` for 'diag' (`_estimate_gaussian_parameters`, `_estimate_gaussian_covariances_diag`,
// `_compute_precision_cholesky`): with resp = exp(log_resp),
//   nk[c] = sum_n resp + 10 eps;  means = resp^T X / nk;  cov = resp^T X^2 / nk - means^2 + reg_covar;
//   weights = nk / sum(nk);  precisions_chol = 1 / sqrt(cov);  and the EM lower bound mean_n log_prob_norm.
// The two k x N x d products run on v_mfma_f64_16x16x4_f64 with the SAMPLE axis as the contraction: a K-slice is four
// samples, A[c][n] = resp (lane: component lane & 15, sample lane >> 4), B[n][j] = X or X^2 (lane: sample lane >> 4,
// feature lane & 15), so both operands are read straight from HBM in 128-byte row segments, exponentiated in registers,
// and never staged.  A wavefront owns 16-row groups (4 K-slices; the next group's loads are issued before the current
// group's products) and keeps the 2 x CT x 2 accumulator tiles of a 32-feature chunk in registers for the whole launch;
// workgroups reduce their four wavefronts through LDS tile by tile and write one partial set each; two small kernels
// add the partial sets in a fixed order and turn the totals into parameters.  Traffic: log_resp once per 32-feature
// chunk, X once.
struct GmmMArgs {
    const double* X; long long ldx; long long N; int d, k;
    const double* log_resp;         // [N, k]
    const double* log_prob_norm;    // [N] or null
    double* part;                   // [G][KP*DP*2 + KP + 1] partial sets
    int KP, DP;                     // padded k (CT*16), padded d (32 * chunks)
    long long set_stride;
};

template <int CT, int RS>   // 16-component tiles, K-slices (of four samples) per register-staged row group
__global__ void __launch_bounds__(256) gmm_mstep_kernel(GmmMArgs g) {
    __shared__ double red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.y * 32;
    const bool first_chunk = blockIdx.y == 0;
    f64x4 s1[CT][2], s2[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) { s1[ct][jt] = {0.0, 0.0, 0.0, 0.0}; s2[ct][jt] = {0.0, 0.0, 0.0, 0.0}; }
    double nk[CT], lpn = 0.0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) nk[ct] = 0.0;
    bool cv[CT], jv[2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) cv[ct] = ct * 16 + c16 < g.k;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) jv[jt] = j0 + jt * 16 + c16 < g.d;

    constexpr int GR = 4 * RS;                      // rows per group
    const long long groups = (g.N + GR - 1) / GR;
    const long long stride = (long long)gridDim.x * 4;
    double la[RS][CT], xb[RS][2], lp[RS];
    auto load = [&](long long gi) {
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const long long n = gi * GR + q + 4 * r;
            const bool ok = n < g.N;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                la[r][ct] = (ok && cv[ct]) ? g.log_resp[(size_t)n * g.k + ct * 16 + c16] : -INFINITY;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) xb[r][jt] = (ok && jv[jt]) ? g.X[(size_t)n * g.ldx + j0 + jt * 16 + c16] : 0.0;
            lp[r] = (ok && first_chunk && c16 == 0 && g.log_prob_norm) ? g.log_prob_norm[n] : 0.0;
        }
    };
    long long gi = (long long)blockIdx.x * 4 + wave;
    if (gi < groups) load(gi);
    for (; gi < groups; gi += stride) {
        double a[RS][CT], x[RS][2];
#pragma unroll
        for (int r = 0; r < RS; ++r) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) a[r][ct] = la[r][ct] == -INFINITY ? 0.0 : exp_nonpos(la[r][ct]);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) x[r][jt] = xb[r][jt];
            lpn += lp[r];
        }
        if (gi + stride < groups) load(gi + stride);
#pragma unroll
        for (int r = 0; r < RS; ++r)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                nk[ct] += a[r][ct];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    s1[ct][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][ct], x[r][jt], s1[ct][jt], 0, 0, 0);
                    s2[ct][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][ct], x[r][jt] * x[r][jt], s2[ct][jt], 0, 0, 0);
                }
            }
    }
    // ---- workgroup reduction, one accumulator tile at a time; D map: component = (lane >> 4) + 4 * reg, feature = lane & 15
    double* set = g.part + (size_t)blockIdx.x * g.set_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const f64x4 v = t == 0 ? s1[ct][jt] : s2[ct][jt];
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave][e * 64 + lane] = v[e];
                __syncthreads();
                {
                    const int e = tid >> 6, l = tid & 63;
                    const double sum = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
                    const int comp = ct * 16 + (l >> 4) + 4 * e, feat = j0 + jt * 16 + (l & 15);
                    set[((size_t)t * g.KP + comp) * g.DP + feat] = sum;
                }
                __syncthreads();
            }
    if (first_chunk) {
        // nk: lane (q, c16) holds the partial of component ct*16 + c16 over its samples; lpn: lanes with c16 == 0
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            red[wave][lane] = nk[ct];
            __syncthreads();
            if (tid < 16) {
                double sum = 0.0;
                for (int w = 0; w < 4; ++w)
                    for (int qq = 0; qq < 4; ++qq) sum += red[w][qq * 16 + tid];
                set[(size_t)2 * g.KP * g.DP + ct * 16 + tid] = sum;
            }
            __syncthreads();
        }
        red[wave][lane] = lpn;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < 4; ++w)
                for (int qq = 0; qq < 4; ++qq) sum += red[w][qq * 16];
            set[(size_t)2 * g.KP * g.DP + g.KP] = sum;
        }
    }
}

// totals[e] = sum over the G partial sets, fixed order: 64 elements per workgroup, 4 interleaved set ranges per element
__global__ void __launch_bounds__(256) gmm_mstep_reduce_kernel(const double* __restrict__ part, long long set_stride, int G,
                                                               long long E, double* __restrict__ totals) {
    __shared__ double sh[4][64];
    const int el = threadIdx.x & 63, pr = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + el;
    double acc = 0.0;
    if (e < E)
        for (int s0 = pr; s0 < G; s0 += 32) {
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (s0 + 4 * i < G) ? part[(size_t)(s0 + 4 * i) * set_stride + e] : 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
        }
    sh[pr][el] = acc;
    __syncthreads();
    if (pr == 0 && e < E) totals[e] = ((sh[0][el] + sh[1][el]) + sh[2][el]) + sh[3][el];
}

// one workgroup per component: parameters from the totals
__global__ void __launch_bounds__(128) gmm_mstep_params_kernel(const double* __restrict__ totals, int KP, int DP, int k, int d,
                                                               long long N, double reg_covar, double* __restrict__ weights,
                                                               double* __restrict__ means, double* __restrict__ covariances,
                                                               double* __restrict__ precisions_chol, double* __restrict__ stats) {
    const int c = blockIdx.x;
    const double* nkp = totals + (size_t)2 * KP * DP;
    const double eps10 = 10.0 * 2.220446049250313e-16;
    double wsum = 0.0;
    for (int i = 0; i < k; ++i) wsum += nkp[i] + eps10;
    const double nk = nkp[c] + eps10;
    int bad = 0;
    for (int j = threadIdx.x; j < d; j += 128) {
        const double m = totals[(size_t)c * DP + j] / nk;
        const double cov = totals[((size_t)KP + c) * DP + j] / nk - m * m + reg_covar;
        means[(size_t)c * d + j] = m;
        covariances[(size_t)c * d + j] = cov;
        precisions_chol[(size_t)c * d + j] = 1.0 / sqrt(cov);
        bad |= !(cov > 0.0);
    }
    if (threadIdx.x == 0) weights[c] = nk / wsum;
    if (c == 0 && threadIdx.x == 0) stats[0] = nkp[KP] / (double)N;       // lower bound: mean log_prob_norm
    if (bad) stats[1] = 1.0;                                              // benign race: every writer stores the same value
}

template <int CT>
void launch_mstep(const GmmMArgs& g, int G, int chunks, hipStream_t st) {
    // four tiles of components leave room for two K-slices of staged operands only (four spill)
    hipLaunchKernelGGL((gmm_mstep_kernel<CT, (CT == 4 ? 2 : 4)>), dim3(G, chunks), dim3(256), 0, st, g);
}

// ---- scratch of the entry point that allocates nothing (clava_gmm_estep_noalloc_f64) -----------------------------------
// The E-step's derived tables (mu*P, P: k x d each; three k-vectors) for up to KMAX components x NOALLOC_DMAX features
// live in this file-scope device array instead of a caller workspace.  Every GPU the library runs on holds its own copy
// (NOALLOC_DOUBLES * 8 = 513.5 KB) until the process ends.  The array's NAME is only a host-side stand-in: the device
// address comes from hipGetSymbolAddress, once per device, and only that pointer is handed to kernels.
constexpr int NOALLOC_DMAX = 512;
constexpr long long NOALLOC_DOUBLES = 2LL * KMAX * NOALLOC_DMAX + 3LL * KMAX;
static __device__ __align__(256) double scratch[NOALLOC_DOUBLES];

constexpr int NOALLOC_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
double* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// device pointer of `scratch` on the current device (looked up once per device, with that device current)
int scratch_for_current_device(double** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
        clava_set_error("gmm_estep_noalloc: hipGetDevice failed or device %d beyond %d: %s", dev, NOALLOC_MAX_DEVICES,
                        hipGetErrorString(e));
        return CLAVA_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        double* p = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            (void)hipGetLastError();
            clava_set_error("gmm_estep_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
            return CLAVA_ERR_LAUNCH;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return CLAVA_OK;
}

}  // namespace

extern "C" {

long long clava_gmm_workspace_doubles(int k, int d) { return 2LL * k * d + 3LL * k; }

long long clava_gmm_noalloc_max_doubles(void) { return NOALLOC_DOUBLES; }

int clava_gmm_estep_noalloc_f64(const double* X, long long ldx, long long N, int d, int k, const double* weights,
                                const double* means, const double* precisions_chol, double* log_prob_norm, double* log_resp,
                                int* labels, void* stream) {
    CLAVA_CHECK_ARG(d > 0 && k > 0, "gmm_estep_noalloc: bad shape d=%d k=%d", d, k);
    if (k > KMAX || clava_gmm_workspace_doubles(k, d) > NOALLOC_DOUBLES) {
        clava_set_error("gmm_estep_noalloc: k=%d, d=%d needs %lld doubles of tables, the static scratch holds %lld "
                        "(k <= %d, 2*k*d + 3*k <= %lld); use clava_gmm_estep_f64 with a caller workspace",
                        k, d, clava_gmm_workspace_doubles(k, d), NOALLOC_DOUBLES, KMAX, NOALLOC_DOUBLES);
        return CLAVA_ERR_LIMIT;
    }
    double* ws = nullptr;
    const int rc = scratch_for_current_device(&ws);
    if (rc != CLAVA_OK) return rc;
    // nothing in the scratch has to start at zero: gmm_prep_kernel rewrites every element the E-step kernel reads
    return clava_gmm_estep_f64(X, ldx, N, d, k, weights, means, precisions_chol, ws, log_prob_norm, log_resp, labels, stream);
}

int clava_gmm_estep_f64(const double* X, long long ldx, long long N, int d, int k, const double* weights,
                        const double* means, const double* precisions_chol, double* workspace, double* log_prob_norm,
                        double* log_resp, int* labels, void* stream) {
    CLAVA_CHECK_ARG(X && weights && means && precisions_chol && workspace && log_prob_norm, "gmm_estep: null argument");
    CLAVA_CHECK_ARG(N > 0 && d > 0 && k > 0 && ldx >= d, "gmm_estep: bad shape N=%lld d=%d k=%d", N, d, k);
    if (k > KMAX) {
        clava_set_error("gmm_estep: k=%d components exceeds the supported maximum of %d", k, KMAX);
        return CLAVA_ERR_LIMIT;
    }
    hipStream_t st = (hipStream_t)stream;
    GmmArgs g;
    g.X = X; g.ldx = ldx; g.N = N; g.d = d; g.k = k;
    double* w = workspace;
    double* mp = w;             w += (size_t)k * d;
    double* P = w;              w += (size_t)k * d;
    double* s_mu2p = w;         w += k;
    double* logdet = w;         w += k;
    double* logw = w;
    g.mp = mp; g.P = P; g.s_mu2p = s_mu2p; g.logdet = logdet; g.logw = logw;
    g.dln2pi = (double)d * 1.8378770664093453;     // log(2 pi)
    g.log_prob_norm = log_prob_norm; g.log_resp = log_resp; g.labels = labels;
    hipLaunchKernelGGL(gmm_prep_kernel, dim3(k), dim3(256), 0, st, means, precisions_chol, weights, k, d, mp, P, s_mu2p, logdet, logw);
    CLAVA_CHECK_LAUNCH("clava_gmm_estep_f64(prep)");
    const long long blocks = (N + ROWS - 1) / ROWS;
    const int grid = (int)(blocks < 65536 ? blocks : 65536);
    // algorithmic traffic: X once, log_resp + log_prob_norm (+ labels) once; flops: two N x d x k products
    ProfScope prof(PROF_GMM, 4.0 * N * d * k, 8.0 * N * (d + (log_resp ? k : 0) + 1) + (labels ? 4.0 * N : 0.0), st);
    const int ct = (k + 15) / 16;
    // resident-table fast path for d <= 128 (up to four 32-column chunks of a row tile held in registers, both tables
    // in <= 128 KB of LDS); wider tables take the chunk-staging kernel
    const int n_chunks = (d + 31) / 32;
    if (n_chunks <= 4) {
        int bad = 1;
        switch (ct) {
            case 1: bad = launch_resident_ct<1>(g, n_chunks, st); break;
            case 2: bad = launch_resident_ct<2>(g, n_chunks, st); break;
            case 3: bad = launch_resident_ct<3>(g, n_chunks, st); break;
            default: bad = launch_resident_ct<4>(g, n_chunks, st); break;
        }
        if (!bad) {
            CLAVA_CHECK_LAUNCH("clava_gmm_estep_f64");
            return CLAVA_OK;
        }
    }
    switch (ct) {
        case 1: hipLaunchKernelGGL((gmm_estep_kernel<1>), dim3(grid), dim3(256), 0, st, g); break;
        case 2: hipLaunchKernelGGL((gmm_estep_kernel<2>), dim3(grid), dim3(256), 0, st, g); break;
        case 3: hipLaunchKernelGGL((gmm_estep_kernel<3>), dim3(grid), dim3(256), 0, st, g); break;
        default: hipLaunchKernelGGL((gmm_estep_kernel<4>), dim3(grid), dim3(256), 0, st, g); break;
    }
    CLAVA_CHECK_LAUNCH("clava_gmm_estep_f64");
    return CLAVA_OK;
}

long long clava_gmm_mstep_workspace_doubles(int k, int d) {
    const long long KP = (k + 15) / 16 * 16, DP = (d + 31) / 32 * 32;
    return (512 + 1) * (2 * KP * DP + KP + 1) + 2;
}

long long clava_gmm_mstep_totals_doubles(int k, int d) {
    const long long KP = (k + 15) / 16 * 16, DP = (d + 31) / 32 * 32;
    return 2 * KP * DP + KP + 1;
}

int clava_gmm_mstep_partial_f64(const double* X, long long ldx, long long N, int d, int k, const double* log_resp,
                                const double* log_prob_norm, double* workspace, double* totals, void* stream) {
    CLAVA_CHECK_ARG(X && log_resp && workspace && totals, "gmm_mstep: null argument");
    CLAVA_CHECK_ARG(N > 0 && d > 0 && k > 0 && ldx >= d, "gmm_mstep: bad shape N=%lld d=%d k=%d", N, d, k);
    if (k > KMAX) {
        clava_set_error("gmm_mstep: k=%d components exceeds the supported maximum of %d", k, KMAX);
        return CLAVA_ERR_LIMIT;
    }
    hipStream_t st = (hipStream_t)stream;
    const int CT = (k + 15) / 16, chunks = (d + 31) / 32;
    GmmMArgs g;
    g.X = X; g.ldx = ldx; g.N = N; g.d = d; g.k = k; g.log_resp = log_resp; g.log_prob_norm = log_prob_norm;
    g.KP = CT * 16; g.DP = chunks * 32;
    g.set_stride = 2LL * g.KP * g.DP + g.KP + 1;
    g.part = workspace;
    const long long groups = (N + 15) / 16;          // at least: a wavefront walks whole row groups
    long long G = (groups + 3) / 4;
    if (G > 512) G = 512;
    if (G < 1) G = 1;
    {
        ProfScope prof(PROF_GMM, 4.0 * N * d * k, 8.0 * N * (d + (double)k * chunks + 1), st);
        switch (CT) {
            case 1: launch_mstep<1>(g, (int)G, chunks, st); break;
            case 2: launch_mstep<2>(g, (int)G, chunks, st); break;
            case 3: launch_mstep<3>(g, (int)G, chunks, st); break;
            default: launch_mstep<4>(g, (int)G, chunks, st); break;
        }
        CLAVA_CHECK_LAUNCH("clava_gmm_mstep_f64");
    }
    const long long E = g.set_stride;
    hipLaunchKernelGGL(gmm_mstep_reduce_kernel, dim3((unsigned)((E + 63) / 64)), dim3(256), 0, st, g.part, g.set_stride, (int)G, E, totals);
    CLAVA_CHECK_LAUNCH("clava_gmm_mstep_f64(reduce)");
    return CLAVA_OK;
}

int clava_gmm_mstep_finalize_f64(const double* totals, long long N, int d, int k, double reg_covar, double* weights,
                                 double* means, double* covariances, double* precisions_chol, double* stats, void* stream) {
    CLAVA_CHECK_ARG(totals && weights && means && covariances && precisions_chol && stats, "gmm_mstep: null argument");
    CLAVA_CHECK_ARG(N > 0 && d > 0 && k > 0, "gmm_mstep: bad shape N=%lld d=%d k=%d", N, d, k);
    if (k > KMAX) {
        clava_set_error("gmm_mstep: k=%d components exceeds the supported maximum of %d", k, KMAX);
        return CLAVA_ERR_LIMIT;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(stats, 0, 2 * sizeof(double), st) != hipSuccess) {
        clava_set_error("gmm_mstep: hipMemsetAsync failed");
        return CLAVA_ERR_LAUNCH;
    }
    const int KP = (k + 15) / 16 * 16, DP = (d + 31) / 32 * 32;
    hipLaunchKernelGGL(gmm_mstep_params_kernel, dim3(k), dim3(128), 0, st, totals, KP, DP, k, d, N, reg_covar, weights, means,
                       covariances, precisions_chol, stats);
    CLAVA_CHECK_LAUNCH("clava_gmm_mstep_f64(params)");
    return CLAVA_OK;
}

int clava_gmm_mstep_f64(const double* X, long long ldx, long long N, int d, int k, const double* log_resp,
                        const double* log_prob_norm, double reg_covar, double* workspace, double* weights, double* means,
                        double* covariances, double* precisions_chol, double* stats, void* stream) {
    CLAVA_CHECK_ARG(X && log_resp && workspace && weights && means && covariances && precisions_chol && stats,
                    "gmm_mstep: null argument");
    double* totals = workspace ? workspace + (size_t)512 * clava_gmm_mstep_totals_doubles(k, d) : nullptr;
    const int rc = clava_gmm_mstep_partial_f64(X, ldx, N, d, k, log_resp, log_prob_norm, workspace, totals, stream);
    if (rc != CLAVA_OK) return rc;
    return clava_gmm_mstep_finalize_f64(totals, N, d, k, reg_covar, weights, means, covariances, precisions_chol, stats, stream);
}

}  // extern "C"
