// Post-sampling decoding of the numeric columns: what runs on every sampled table right after the chain.
//
//   quantile_inverse_kernel   scikit-learn QuantileTransformer(output_distribution='normal').inverse_transform, the
//                             `dataset.num_transform.inverse_transform(X_gen[:, :n])` call of pipeline_utils.py:913
//                             (the transformer is built at lib/data.py:217-223):
//                               u = norm.cdf(x); y = np.interp(u, references_, quantiles_[:, col]);
//                               u + 1e-7 > 1 -> quantiles_[-1]; u - 1e-7 < 0 -> quantiles_[0]; NaN stays NaN
//                             (sklearn/preprocessing/_data.py `_transform_col`, inverse branch).
//   round_to_values_kernel    lib.round_columns (lib/data.py:627-632): snap a column to the nearest of the sorted unique
//                             training values, distance sqrt((x-u)^2) like scipy's cdist, first minimum wins.
//
// Both are elementwise float64 streams over [N, d] (HBM-bound: 16 bytes per element); the tables (<= 1000 quantiles per
// column, <= a few thousand unique values) stay in L1/L2.  One thread per element, consecutive threads on consecutive
// elements of the row-major matrix (coalesced), a 10-step binary search per element.
#include "common.h"
#include "clava_hip.h"

namespace {

constexpr double BOUNDS_THRESHOLD = 1e-7;   // sklearn/preprocessing/_data.py

// scipy.special.ndtr (cephes / xsf): the standard normal CDF
__device__ __forceinline__ double ndtr(double a) {
    const double x = a * 0.70710678118654752440;
    const double z = fabs(x);
    if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(x);
    double y = 0.5 * erfc(z);
    if (x > 0) y = 1.0 - y;
    return y;
}

// numpy's np.interp for one value (numpy/_core/src/multiarray/compiled_base.c arr_interp): xp increasing, n >= 1
__device__ __forceinline__ double interp1(double x, const double* __restrict__ xp, const double* __restrict__ fp, int fstride, int n) {
    if (isnan(x)) return x;
    if (x > xp[n - 1]) return fp[(size_t)(n - 1) * fstride];
    if (x < xp[0]) return fp[0];
    int lo = 0, hi = n - 1;                 // invariant: xp[lo] <= x, and x < xp[hi] or hi == n - 1
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (x >= xp[mid]) lo = mid; else hi = mid;
    }
    int j = lo;
    if (x >= xp[n - 1]) j = n - 1;          // x == xp[n-1]
    if (j == n - 1) return fp[(size_t)j * fstride];
    const double x0 = xp[j], y0 = fp[(size_t)j * fstride];
    if (x0 == x) return y0;
    const double x1 = xp[j + 1], y1 = fp[(size_t)(j + 1) * fstride];
    const double slope = (y1 - y0) / (x1 - x0);
    double r = slope * (x - x0) + y0;
    if (isnan(r)) {
        r = slope * (x - x1) + y1;
        if (isnan(r) && y0 == y1) r = y0;
    }
    return r;
}

__global__ void __launch_bounds__(256) quantile_inverse_kernel(const double* __restrict__ x, long long ldx, long long N, int d,
                                                               const double* __restrict__ quantiles, const double* __restrict__ refs,
                                                               int nq, double* __restrict__ out, long long ldo) {
    const long long total = N * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / d;
        const int c = (int)(i - n * d);
        const double v = x[(size_t)n * ldx + c];
        const double u = ndtr(v);
        double y = interp1(u, refs, quantiles + c, d, nq);
        if (u + BOUNDS_THRESHOLD > 1.0) y = quantiles[(size_t)(nq - 1) * d + c];
        if (u - BOUNDS_THRESHOLD < 0.0) y = quantiles[c];
        out[(size_t)n * ldo + c] = y;
    }
}

__global__ void __launch_bounds__(256) round_to_values_kernel(double* __restrict__ x, long long ldx, long long N, int col,
                                                              const double* __restrict__ values, int U) {
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long long)gridDim.x * blockDim.x) {
        const double v = x[(size_t)n * ldx + col];
        double pick = values[0];
        const double d0 = v - pick;
        double best = sqrt(d0 * d0);
        if (!isnan(best))                                   // a NaN sample: argmin of all-NaN distances is index 0
            for (int j = 1; j < U; ++j) {
                const double dlt = v - values[j];
                const double dist = sqrt(dlt * dlt);
                if (dist < best) { best = dist; pick = values[j]; }   // strict: the first minimum wins, like argmin
            }
        x[(size_t)n * ldx + col] = pick;
    }
}

}  // namespace

extern "C" {

int clava_quantile_inverse_f64(const double* x, long long ldx, long long N, int d, const double* quantiles, const double* references,
                               int n_quantiles, double* out, long long ldo, void* stream) {
    CLAVA_CHECK_ARG(x && quantiles && references && out && N > 0 && d > 0 && n_quantiles >= 1 && ldx >= d && ldo >= d,
                    "quantile_inverse: bad arguments");
    const long long total = N * d;
    const long long blocks = (total + 255) / 256;
    ProfScope prof(PROF_MISC, 0.0, 16.0 * total, (hipStream_t)stream);
    hipLaunchKernelGGL(quantile_inverse_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, N, d, quantiles, references, n_quantiles, out, ldo);
    CLAVA_CHECK_LAUNCH("clava_quantile_inverse_f64");
    return CLAVA_OK;
}

int clava_round_to_values_f64(double* x, long long ldx, long long N, int col, const double* values, int n_values, void* stream) {
    CLAVA_CHECK_ARG(x && values && N > 0 && col >= 0 && col < ldx && n_values >= 1, "round_to_values: bad arguments");
    const long long blocks = (N + 255) / 256;
    ProfScope prof(PROF_MISC, 0.0, 16.0 * N, (hipStream_t)stream);
    hipLaunchKernelGGL(round_to_values_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, N, col, values, n_values);
    CLAVA_CHECK_LAUNCH("clava_round_to_values_f64");
    return CLAVA_OK;
}

}  // extern "C"
