// The denoiser's NARROW end layers in training (head: Linear(512, D); proj: Linear(D, 128); D = a handful of columns
// for numeric tables) as memory-bound kernels.  On the tiled float32 MFMA kernels these products cost 16-27 us each -- a
// 64 x 64 tile for 2 useful columns, 32 workgroups for the weight gradients -- on the critical path of a 0.7 ms step.
//
//   narrow_head     out[b][d] = bias[d] + sum_k h[b][k] W[d][k]                   (tab_ddpm/modules.py:460, mlp head)
//   narrow_dgrad    g[b][j] = relu'(act[b][j]) * sum_d dy[b][d] W[d][j], written as split rows (split.h) and / or float32:
//                   the input gradient of the head with the ReLU mask of the last block and the split pack folded in
//                   (the autograd backward of modules.py:460 / :174-180)
//   narrow_wgrad    P[d][j] = sum_b A[b][d] Bm[b][j] per batch range (gradient slabs, like clava_gemm_wgrad_f32), with the
//                   column sums of either operand (the bias gradient) alongside: head dW = dy^T h, proj dW^T = x^T g
//
// Fixed summation orders, no atomics: results are bitwise reproducible.  D <= 16.
#include "clava_hip.h"
#include "common.h"
#include "split.h"
#include "step.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DMAX = 16;

// ---------------------------------------------------------------------------------------------- head forward
// 16 lanes per row (each 16 bytes of every 256-byte piece of the row), 16 rows per workgroup; W in LDS.
template <int DM>
__global__ void __launch_bounds__(256) narrow_head_kernel(const float* __restrict__ h, long long ldh, int B, int K,
                                                          const float* __restrict__ W, const float* __restrict__ bias, int D,
                                                          float* __restrict__ out, int ld_out) {
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [D][K]
    for (int i = threadIdx.x; i < D * K; i += 256) wl[i] = W[i];
    __syncthreads();
    const int sub = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const float* hr = h + (row < B ? row : (long long)B - 1) * ldh;
    float acc[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) acc[d] = 0.f;
    for (int k = 4 * sub; k < K; k += 64) {
        const f4 hv = *reinterpret_cast<const f4*>(hr + k);
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            if (d < D) {
                const f4 wv = *reinterpret_cast<const f4*>(wl + d * K + k);
                acc[d] = fmaf(hv[0], wv[0], acc[d]);
                acc[d] = fmaf(hv[1], wv[1], acc[d]);
                acc[d] = fmaf(hv[2], wv[2], acc[d]);
                acc[d] = fmaf(hv[3], wv[3], acc[d]);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DM; ++d)
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc[d] += __shfl_xor(acc[d], o, 64);
    if (sub == 0 && row < B) {
#pragma unroll
        for (int d = 0; d < DM; ++d)
            if (d < D) out[row * ld_out + d] = acc[d] + bias[d];
    }
}

// ---------------------------------------------------------------------------------------------- head input gradient
// one thread per (row, 4 columns); W (D x J floats) and the dy row come through the caches
__global__ void __launch_bounds__(256) narrow_dgrad_kernel(const float* __restrict__ dy, int ld_dy, int B, int D,
                                                           const float* __restrict__ W, int J, const float* __restrict__ act,
                                                           long long ld_act, float slope, float* __restrict__ g32, long long ldg,
                                                           char* __restrict__ gs, long long pitch, float scale, int* overflow) {
    const int groups = ((J + 31) & ~31) >> 2;
    const long long total = (long long)B * groups;
    const bool vecw = (J & 3) == 0 && (reinterpret_cast<unsigned long long>(W) & 15) == 0;
    bool ovf = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / groups;
        const int c0 = (int)(i - row * groups) << 2;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c0 < J) {
            for (int d = 0; d < D; ++d) {
                const float a = dy[row * ld_dy + d];
                if (vecw) {
                    const f4 w = *reinterpret_cast<const f4*>(W + (long long)d * J + c0);
                    v[0] = fmaf(a, w[0], v[0]); v[1] = fmaf(a, w[1], v[1]); v[2] = fmaf(a, w[2], v[2]); v[3] = fmaf(a, w[3], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c0 + r < J) v[r] = fmaf(a, W[(long long)d * J + c0 + r], v[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (c0 + r >= J) v[r] = 0.f;
                else if (act) v[r] *= (act[row * ld_act + c0 + r] > 0.f) ? 1.f : slope;
            }
            if (g32) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + r < J) g32[row * ldg + c0 + r] = v[r];
            }
        }
        if (gs) {
            h4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = v[r] * scale;
                ovf |= !(fabsf(s) <= 65504.f);
                hi[r] = (_Float16)s;
                lo[r] = (_Float16)(s - (float)hi[r]);
            }
            char* op = gs + row * pitch + (long long)(c0 >> 5) * 128 + ((c0 & 31) << 1);
            *reinterpret_cast<h4*>(op) = hi;
            *reinterpret_cast<h4*>(op + 64) = lo;
        }
    }
    if (ovf && overflow) *overflow = 1;
}

// ---------------------------------------------------------------------------------------------- narrow weight gradient
// grid (ceil(J / 64), splits); thread (cq = tid & 15, rg = tid >> 4): columns j0 + 4 cq .. + 3 of Bm, rows rg, rg + 16, ...
// of the batch range, four 16-byte loads in flight per thread (the loop is latency bound: a first version with one 4-byte
// load per iteration and 128 iterations per thread took 50 us for 8 MB).  The narrow operand's rows go through LDS 256 at a
// time.  Row groups are combined in a fixed order: xor-shuffles over the four groups of a wave, then LDS over the waves.
template <int DM>
__global__ void __launch_bounds__(256) narrow_wgrad_kernel(const float* __restrict__ A, long long lda, int D,
                                                           const float* __restrict__ Bm, long long ldb, int J, int B, int splits,
                                                           float* __restrict__ dw, long long stride_d, long long stride_j,
                                                           float* __restrict__ db_narrow, float* __restrict__ db_wide,
                                                           long long slab_stride) {
    __shared__ float As[256 * DM];
    __shared__ __attribute__((aligned(16))) float part[4][DM + 1][64];
    __shared__ float parta[4][DM];
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + 4 * cq;
    const bool vec = (ldb & 3) == 0 && (reinterpret_cast<unsigned long long>(Bm) & 15) == 0 && j + 3 < J;
    const int rows_per = (B + splits - 1) / splits;
    const int r0 = blockIdx.y * rows_per, r1 = min(B, r0 + rows_per);
    f4 acc[DM], bsum = {0.f, 0.f, 0.f, 0.f};
    float asum[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) { acc[d] = f4{0.f, 0.f, 0.f, 0.f}; asum[d] = 0.f; }
    for (int rb = r0; rb < r1; rb += 256) {
        const int nr = min(256, r1 - rb);
        __syncthreads();
        for (int i = threadIdx.x; i < nr * D; i += 256) {
            const int r = i / D, d = i - r * D;
            As[r * DM + d] = A[(long long)(rb + r) * lda + d];
        }
        __syncthreads();
        for (int r = rg; r < nr; r += 64) {
            f4 bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + 16 * u;
                bv[u] = f4{0.f, 0.f, 0.f, 0.f};
                if (rr < nr) {
                    const float* p = Bm + (long long)(rb + rr) * ldb + j;
                    if (vec) bv[u] = *reinterpret_cast<const f4*>(p);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (j + e < J) bv[u][e] = p[e];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = min(r + 16 * u, nr - 1);          // rows beyond the range carry bv = 0; asum is guarded
                const bool live = r + 16 * u < nr;
                bsum += bv[u];
#pragma unroll
                for (int d = 0; d < DM; ++d) {
                    if (d < D) {
                        const float a = As[rr * DM + d];
                        acc[d] += a * bv[u];
                        asum[d] += live ? a : 0.f;
                    }
                }
            }
        }
    }
    // the four row groups of a wave sit 16 lanes apart
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
#pragma unroll
        for (int d = 0; d < DM; ++d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[d][e] += __shfl_xor(acc[d][e], o, 64);
            asum[d] += __shfl_xor(asum[d], o, 64);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[e] += __shfl_xor(bsum[e], o, 64);
    }
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int d = 0; d < DM; ++d) *reinterpret_cast<f4*>(&part[wave][d][4 * cq]) = acc[d];
        *reinterpret_cast<f4*>(&part[wave][DM][4 * cq]) = bsum;
        if (cq == 0) {
#pragma unroll
            for (int d = 0; d < DM; ++d) parta[wave][d] = asum[d];
        }
    }
    __syncthreads();
    float* slab = dw + (long long)blockIdx.y * slab_stride;
    if (threadIdx.x < 64) {
        const int c = threadIdx.x, jj = blockIdx.x * 64 + c;
        if (jj < J) {
#pragma unroll
            for (int d = 0; d < DM; ++d)
                if (d < D) slab[d * stride_d + jj * stride_j] = ((part[0][d][c] + part[1][d][c]) + part[2][d][c]) + part[3][d][c];
            if (db_wide) db_wide[(long long)blockIdx.y * slab_stride + jj] = ((part[0][DM][c] + part[1][DM][c]) + part[2][DM][c]) + part[3][DM][c];
        }
    }
    if (db_narrow && blockIdx.x == 0 && threadIdx.x < D)
        db_narrow[(long long)blockIdx.y * slab_stride + threadIdx.x] =
            ((parta[0][threadIdx.x] + parta[1][threadIdx.x]) + parta[2][threadIdx.x]) + parta[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------- narrow product from split rows
// out[b][d] = alpha * (scale_acc * sum_k A[b][k] W[d][k] + bias[d] + bias2[d]) for N <= 16 outputs, A and W as split rows
// (split.h): the denoiser head and the last input-gradient product of the guidance classifier in the SAMPLING chain.  On
// the tiled MFMA kernel these use a 64-column tile for 2 useful columns and 157 workgroups walking K in 32-deep blocks
// (26 us at 20000 rows); here 16 lanes read a row's planes once (4 * Kp bytes), hi + lo is exact in float32, plain FMAs.
// GAUSS: the product is the guidance gradient of a reverse step (its last input-gradient product) and the row's fused
// Gaussian posterior step (step.h: gauss_step_element, the body of gauss_step_kernel) follows in the same thread - one launch
// and one cross-stream hand-over less per reverse step (`ga.grad` = this product's output, `ga.eps` = the denoiser's).
template <int DM, bool GAUSS>
__global__ void __launch_bounds__(256) narrow_split_kernel(SplitGemmArgs g, GaussStepArgs ga) {
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [N][Kp], hi + lo
    const int Kp = g.Kp, D = g.N;
    for (int i = threadIdx.x; i < D * (Kp >> 2); i += 256) {
        const int d = i / (Kp >> 2), k = (i - d * (Kp >> 2)) << 2;
        const char* p = g.W + (long long)d * g.pitchW + (k >> 5) * 128 + ((k & 31) << 1);
        const h4 hi = *reinterpret_cast<const h4*>(p), lo = *reinterpret_cast<const h4*>(p + 64);
        *reinterpret_cast<f4*>(wl + d * Kp + k) = f4{(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2],
                                                    (float)hi[3] + (float)lo[3]};
    }
    __syncthreads();
    const int sub = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const char* ar = g.A + (row < g.M ? row : (long long)g.M - 1) * g.pitchA;
    float acc[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) acc[d] = 0.f;
    for (int k = 4 * sub; k < Kp; k += 64) {
        const char* p = ar + (k >> 5) * 128 + ((k & 31) << 1);
        const h4 hi = *reinterpret_cast<const h4*>(p), lo = *reinterpret_cast<const h4*>(p + 64);
        const float x0 = (float)hi[0] + (float)lo[0], x1 = (float)hi[1] + (float)lo[1], x2 = (float)hi[2] + (float)lo[2],
                    x3 = (float)hi[3] + (float)lo[3];
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            if (d < D) {
                const f4 wv = *reinterpret_cast<const f4*>(wl + d * Kp + k);
                acc[d] = fmaf(x0, wv[0], acc[d]);
                acc[d] = fmaf(x1, wv[1], acc[d]);
                acc[d] = fmaf(x2, wv[2], acc[d]);
                acc[d] = fmaf(x3, wv[3], acc[d]);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DM; ++d)
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc[d] += __shfl_xor(acc[d], o, 64);
    if (sub == 0 && row < g.M) {
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            if (d < D) {
                float b = 0.f;
                if (g.bias) b += g.bias[d];
                if (g.bias2) b += g.bias2[d];
                g.out_f32[row * g.ldc + d] = g.alpha * (acc[d] * g.scale_acc + b);
            }
        }
        if constexpr (GAUSS) {
#pragma unroll
            for (int d = 0; d < DM; ++d)
                if (d < D) gauss_step_element(ga, (int)row, d, ga.eps[(size_t)row * ga.ld_eps + d]);    // (reads the gradient this thread just stored)
        }
    }
}

template <typename F>
int by_width(int D, F&& f) {
    if (D <= 2) return f(std::integral_constant<int, 2>());
    if (D <= 4) return f(std::integral_constant<int, 4>());
    if (D <= 8) return f(std::integral_constant<int, 8>());
    return f(std::integral_constant<int, 16>());
}

}  // namespace

bool clava_narrow_split_ok(const SplitGemmArgs& g) {
    return g.N >= 1 && g.N <= DMAX && g.out_f32 && !g.out_split && !g.mask && g.slabs <= 1 && g.act == CLAVA_ACT_NONE &&
           (long long)g.N * g.Kp <= 16384 && g.pitchA % 8 == 0 && g.pitchW % 8 == 0 &&
           (reinterpret_cast<unsigned long long>(g.A) & 7) == 0 && (reinterpret_cast<unsigned long long>(g.W) & 7) == 0;
}

int clava_launch_narrow_split(const SplitGemmArgs& g, hipStream_t stream) {
    CLAVA_CHECK_ARG(clava_narrow_split_ok(g) && g.M > 0 && g.Kp > 0 && g.Kp % 32 == 0 && g.ldc >= g.N, "narrow_split: unsupported product");
    ProfScope prof(PROF_MISC, 2.0 * g.M * g.N * (double)(g.K > 0 ? g.K : g.Kp), 4.0 * g.M * ((double)g.Kp + g.N), stream);
    return by_width(g.N, [&](auto dm) {
        hipLaunchKernelGGL((narrow_split_kernel<decltype(dm)::value, false>), dim3((g.M + 15) / 16), dim3(256), (size_t)g.N * g.Kp * 4, stream, g,
                           GaussStepArgs{});
        CLAVA_CHECK_LAUNCH("narrow_split");
        return (int)CLAVA_OK;
    });
}

// the same product followed, row by row, by the Gaussian reverse step that consumes it as its guidance gradient
int clava_launch_narrow_split_gauss(const SplitGemmArgs& g, const GaussStepArgs& ga, hipStream_t stream) {
    CLAVA_CHECK_ARG(clava_narrow_split_ok(g) && g.M > 0 && g.Kp > 0 && g.Kp % 32 == 0 && g.ldc >= g.N, "narrow_split_gauss: unsupported product");
    CLAVA_CHECK_ARG(ga.B == g.M && ga.n_num == g.N && ga.grad == g.out_f32 && ga.ld_grad == g.ldc && ga.eps && ga.x,
                    "narrow_split_gauss: the step's gradient must be this product's output, one column per numeric column");
    ProfScope prof(PROF_MISC, 2.0 * g.M * g.N * (double)(g.K > 0 ? g.K : g.Kp), 4.0 * g.M * ((double)g.Kp + g.N) + 24.0 * g.M * g.N, stream);
    return by_width(g.N, [&](auto dm) {
        hipLaunchKernelGGL((narrow_split_kernel<decltype(dm)::value, true>), dim3((g.M + 15) / 16), dim3(256), (size_t)g.N * g.Kp * 4, stream, g, ga);
        CLAVA_CHECK_LAUNCH("narrow_split_gauss");
        return (int)CLAVA_OK;
    });
}

extern "C" {

int clava_narrow_head_f32(const float* h, long long ldh, int B, int K, const float* W, const float* bias, int D, float* out, int ld_out,
                          void* stream) {
    CLAVA_CHECK_ARG(h && W && bias && out && B > 0 && K > 0 && D > 0 && D <= DMAX && ld_out >= D && ldh >= K, "narrow_head: bad arguments");
    CLAVA_CHECK_ARG(K % 4 == 0 && ldh % 4 == 0 && (reinterpret_cast<unsigned long long>(h) & 15) == 0 && D * K * 4 <= 64 * 1024,
                    "narrow_head: K = %d must be a multiple of 4 with 16-byte aligned rows, D * K <= 16384", K);
    ProfScope prof(PROF_MISC, 2.0 * B * K * D, 4.0 * B * (K + D), (hipStream_t)stream);
    return by_width(D, [&](auto dm) {
        hipLaunchKernelGGL((narrow_head_kernel<decltype(dm)::value>), dim3((B + 15) / 16), dim3(256), (size_t)D * K * 4, (hipStream_t)stream,
                           h, ldh, B, K, W, bias, D, out, ld_out);
        CLAVA_CHECK_LAUNCH("clava_narrow_head_f32");
        return (int)CLAVA_OK;
    });
}

int clava_narrow_dgrad_f32(const float* dy, int ld_dy, int B, int D, const float* W, int J, const float* act, long long ld_act,
                           float slope, float* g_f32, long long ldg, void* g_split, long long pitch, int shift, int* overflow,
                           void* stream) {
    CLAVA_CHECK_ARG(dy && W && B > 0 && D > 0 && D <= DMAX && J > 0 && ld_dy >= D && (g_f32 || g_split), "narrow_dgrad: bad arguments");
    CLAVA_CHECK_ARG(!act || ld_act >= J, "narrow_dgrad: activation rows shorter than J");
    CLAVA_CHECK_ARG(!g_f32 || ldg >= J, "narrow_dgrad: ldg too small");
    const int Jp = (J + 31) & ~31;
    if (g_split) {
        CLAVA_CHECK_ARG(pitch >= 4LL * Jp && pitch % 8 == 0 && (reinterpret_cast<unsigned long long>(g_split) & 7) == 0,
                        "narrow_dgrad: split pitch %lld too small or misaligned", pitch);
        CLAVA_CHECK_ARG(shift >= -24 && shift <= 24, "narrow_dgrad: shift %d out of range", shift);
    }
    const long long total = (long long)B * (Jp / 4);
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    ProfScope prof(PROF_MISC, 2.0 * B * J * D, 4.0 * B * J * (1 + (act ? 1 : 0) + (g_f32 ? 1 : 0)), (hipStream_t)stream);
    hipLaunchKernelGGL(narrow_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, B, D, W, J, act, ld_act, slope,
                       g_f32, ldg, (char*)g_split, pitch, ldexpf(1.f, shift), overflow);
    CLAVA_CHECK_LAUNCH("clava_narrow_dgrad_f32");
    return CLAVA_OK;
}

int clava_narrow_wgrad_f32(const float* A, long long lda, int D, const float* Bm, long long ldb, int J, int B, int splits,
                           float* dw_slabs, long long stride_d, long long stride_j, float* db_narrow_slabs, float* db_wide_slabs,
                           long long slab_stride, void* stream) {
    CLAVA_CHECK_ARG(A && Bm && dw_slabs && B > 0 && D > 0 && D <= DMAX && J > 0 && splits >= 1 && splits <= 65535 && lda >= D && ldb >= J,
                    "narrow_wgrad: bad arguments");
    CLAVA_CHECK_ARG(stride_d > 0 && stride_j > 0, "narrow_wgrad: output strides must be positive");
    ProfScope prof(PROF_MISC, 2.0 * B * J * D, 4.0 * B * (J + D), (hipStream_t)stream);
    return by_width(D, [&](auto dm) {
        hipLaunchKernelGGL((narrow_wgrad_kernel<decltype(dm)::value>), dim3((J + 63) / 64, splits), dim3(256), 0, (hipStream_t)stream, A, lda,
                           D, Bm, ldb, J, B, splits, dw_slabs, stride_d, stride_j, db_narrow_slabs, db_wide_slabs, slab_stride);
        CLAVA_CHECK_LAUNCH("clava_narrow_wgrad_f32");
        return (int)CLAVA_OK;
    });
}

}  // extern "C"
