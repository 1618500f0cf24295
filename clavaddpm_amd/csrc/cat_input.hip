// First denoiser layer of a table with categorical columns, without the [B, sum K] log-one-hot image.
//
// The reference feeds MLPDiffusion.proj (tab_ddpm/modules.py:459) the concatenation of the numeric columns and the
// log-one-hot of every categorical feature (index_to_log_onehot, tab_ddpm/utils.py:127-134: 0 at the hot class,
// L = log(1e-30) elsewhere).  With S_j = sum_c W[j][n_num + c] the product collapses to
//     h_j = sum_i W[j][i] x_i + (b_j + L S_j) + emb_j - L sum_f W[j][n_num + off_f + idx_f]
// i.e. a K = n_num product plus m gathered weight columns per row instead of a K = n_num + sum K product: at the
// scale-sweep shape (100 numeric, 50 features, sum K = 8297) 150 column reads per row instead of 8397, and the step
// kernel no longer writes the 4 * sum K bytes per row of the image (last bullet; section 8d).
// Accumulation order is fixed (numeric columns in order, then features in order): results do not depend on the grid.
#include "split.h"
#include "step.h"

namespace {

struct f2 { float x, y; };
__device__ __forceinline__ f2 ld2(const float* p) { const float2 v = *reinterpret_cast<const float2*>(p); return f2{v.x, v.y}; }

constexpr int ROWS = 2;          // rows per wavefront pass: one read of a numeric weight column serves two rows

__global__ void __launch_bounds__(256) cat_input_kernel(CatInputArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = (a.B + ROWS - 1) / ROWS;
    const float slope = a.act == CLAVA_ACT_RELU ? 0.f : (a.act == CLAVA_ACT_LEAKY ? 0.01f : 1.f);
    unsigned amax = 0;
    for (int grp = blockIdx.x * 4 + wave; grp < groups; grp += gridDim.x * 4) {
        const int row0 = grp * ROWS;
        for (int c0 = 0; c0 < a.d1; c0 += 128) {
            const int j = c0 + 2 * lane;
            f2 base = ld2(a.cat_bias + j);
            if (a.emb && a.ld_emb == 0) { const f2 e = ld2(a.emb + j); base.x += e.x; base.y += e.y; }
            f2 num[ROWS], hot[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) num[r] = hot[r] = f2{0.f, 0.f};
            for (int i0 = 0; i0 < a.n_num; i0 += 64) {
                // 64 numeric columns of each row in one coalesced load; lane ii's value is broadcast per column below
                int xr[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
                    xr[r] = (i0 + lane < a.n_num) ? __float_as_int(a.xnum[(size_t)min(row0 + r, a.B - 1) * a.ld_x + i0 + lane]) : 0;
                const int ni = min(64, a.n_num - i0);
                // eight weight columns in flight per pass (a loop with a run-time trip count waits for every load in turn)
                for (int ii = 0; ii < ni; ii += 8) {
                    f2 w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = ld2(a.num_wt + (size_t)(i0 + min(ii + u, ni - 1)) * a.d1 + j);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (ii + u < ni) {
#pragma unroll
                            for (int r = 0; r < ROWS; ++r) {
                                const float xv = __int_as_float(__builtin_amdgcn_readlane(xr[r], ii + u));
                                num[r].x += xv * w[u].x; num[r].y += xv * w[u].y;
                            }
                        }
                    }
                }
            }
            for (int f0 = 0; f0 < a.m; f0 += 64) {
                const int f = f0 + lane;
                // a chain whose activations left the float16 range (the caller reruns it on the float32 kernels) can hand
                // over arg-max indices of all-NaN scores: clamp into the feature, never gather outside the weight matrix
                int col[ROWS];
                const int o0 = (f < a.m) ? a.offsets[f] : 0, kf = (f < a.m) ? a.offsets[f + 1] - o0 : 1;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int cls = (f < a.m) ? a.idx[(size_t)min(row0 + r, a.B - 1) * a.m + f] : 0;
                    col[r] = o0 + min(max(cls, 0), kf - 1);
                }
                const int nf = min(64, a.m - f0);
                for (int ff = 0; ff < nf; ff += 8) {
                    f2 w[ROWS][8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int fu = min(ff + u, nf - 1);
#pragma unroll
                        for (int r = 0; r < ROWS; ++r)
                            w[r][u] = ld2(a.cat_wt + (size_t)__builtin_amdgcn_readlane(col[r], fu) * a.d1 + j);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (ff + u < nf) {
#pragma unroll
                            for (int r = 0; r < ROWS; ++r) { hot[r].x += w[r][u].x; hot[r].y += w[r][u].y; }
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int row = row0 + r;
                if (row >= a.B) continue;
                float v[2] = {(base.x + num[r].x) - a.log_tiny * hot[r].x, (base.y + num[r].y) - a.log_tiny * hot[r].y};
                if (a.emb && a.ld_emb) { const f2 e = ld2(a.emb + (size_t)row * a.ld_emb + j); v[0] += e.x; v[1] += e.y; }
                if (a.out_f32) {
                    const float y0 = fmaxf(v[0], v[0] * slope), y1 = fmaxf(v[1], v[1] * slope);
                    *reinterpret_cast<float2*>(a.out_f32 + (size_t)row * a.ld_f32 + j) = make_float2(y0, y1);
                }
                _Float16 hi[2], lo[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float x = fmaxf(v[e], v[e] * slope) * a.scale_out;
                    amax = max(amax, __float_as_uint(x) << 1);
                    hi[e] = (_Float16)x;
                    lo[e] = (_Float16)(x - (float)hi[e]);
                }
                char* op = a.out + (size_t)row * a.pitch_out + (size_t)(j >> 5) * 128 + ((j & 31) << 1);
                *reinterpret_cast<unsigned*>(op) = (unsigned)__builtin_bit_cast(unsigned short, hi[0]) | ((unsigned)__builtin_bit_cast(unsigned short, hi[1]) << 16);
                *reinterpret_cast<unsigned*>(op + 64) = (unsigned)__builtin_bit_cast(unsigned short, lo[0]) | ((unsigned)__builtin_bit_cast(unsigned short, lo[1]) << 16);
            }
        }
    }
    if (amax > (0x477fe000u << 1) && a.overflow) *a.overflow = 1;
}

// The gather form's weight operands, rebuilt every training step (the optimizer moved the weights): wt[k][c] = w[c][k]
// (rows = input columns, 512 contiguous bytes each) and, per 64-column tile, the float64 sums of its categorical columns
// per output row; a second launch adds the tiles in order: bias[c] = b[c] + L * sum_k w[c][n_num + k].  Two launches for
// what took seven framework kernels (transpose copy, slice sum in float64, scale, add, casts): 50 -> ~10 us on the
// stream the first layer waits for.
constexpr int WT_TILE = 64;
__global__ void __launch_bounds__(256) cat_weights_kernel(const float* __restrict__ w, long long ldw, int D, int n_num,
                                                          float* __restrict__ wt, double* __restrict__ partial) {
    __shared__ float tile[128][WT_TILE + 1];
    const int k0 = blockIdx.x * WT_TILE;
    for (int i = threadIdx.x; i < 128 * WT_TILE; i += 256) {
        const int c = i / WT_TILE, k = i - c * WT_TILE;
        tile[c][k] = (k0 + k < D) ? w[(long long)c * ldw + k0 + k] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 128 * WT_TILE; i += 256) {
        const int k = i >> 7, c = i & 127;
        if (k0 + k < D) wt[(long long)(k0 + k) * 128 + c] = tile[c][k];
    }
    if (threadIdx.x < 128) {
        double acc = 0.0;
        for (int k = 0; k < WT_TILE; ++k)
            if (k0 + k >= n_num) acc += (double)tile[threadIdx.x][k];          // columns beyond D were staged as zeros
        partial[(long long)blockIdx.x * 128 + threadIdx.x] = acc;
    }
}
// 8 x 128 threads: part p adds the tiles p, p + 8, ... of its column (a single chain of 132 dependent loads took 34 us),
// then the eight partial sums are added in order
__global__ void __launch_bounds__(1024) cat_bias_kernel(const double* __restrict__ partial, int tiles, const float* __restrict__ b,
                                                        double log_tiny, float* __restrict__ bias) {
    __shared__ double part[8][128];
    const int c = threadIdx.x & 127, p = threadIdx.x >> 7;
    double acc = 0.0;
    for (int t = p; t < tiles; t += 8) acc += partial[(long long)t * 128 + c];
    part[p][c] = acc;
    __syncthreads();
    if (p == 0) {
        double sum = part[0][c];
#pragma unroll
        for (int q = 1; q < 8; ++q) sum += part[q][c];
        bias[c] = (float)((double)b[c] + log_tiny * sum);
    }
}

}  // namespace

extern "C" int clava_cat_input_weights_tiles(int D) { return D > 0 ? (D + WT_TILE - 1) / WT_TILE : 0; }

extern "C" int clava_cat_input_weights_f32(const float* w, long long ldw, int d1, int D, int n_num, const float* b, double log_tiny,
                                           float* wt, double* partial, float* cat_bias, void* stream) {
    CLAVA_CHECK_ARG(d1 == 128, "cat_input_weights: layer width %d not supported (built for 128, dim_t of every config)", d1);
    CLAVA_CHECK_ARG(w && b && wt && partial && cat_bias && D > 0 && n_num >= 0 && n_num <= D && ldw >= D, "cat_input_weights: bad arguments");
    const int tiles = (D + WT_TILE - 1) / WT_TILE;
    hipLaunchKernelGGL(cat_weights_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, w, ldw, D, n_num, wt, partial);
    CLAVA_CHECK_LAUNCH("clava_cat_input_weights_f32");
    hipLaunchKernelGGL(cat_bias_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, tiles, b, log_tiny, cat_bias);
    CLAVA_CHECK_LAUNCH("clava_cat_input_weights_f32(bias)");
    return CLAVA_OK;
}

extern "C" int clava_cat_input_f32(const float* xnum, int ld_x, const int* idx, int B, int n_num, int m, int d1, const int* offsets,
                                   const float* num_wt, const float* cat_wt, const float* cat_bias, const float* emb, long long ld_emb,
                                   float log_tiny, int act, void* out_split, long long pitch, int shift, float* out_f32,
                                   long long ld_f32, int* overflow, void* stream) {
    CLAVA_CHECK_ARG(out_split && pitch >= 4LL * d1 && pitch % 8 == 0 && shift >= -24 && shift <= 24, "cat_input: bad split output");
    CLAVA_CHECK_ARG(!emb || ld_emb == 0 || (ld_emb >= d1 && ld_emb % 2 == 0), "cat_input: embedding rows must be >= d1 floats, 8-byte aligned");
    CLAVA_CHECK_ARG(!out_f32 || (ld_f32 >= d1 && ld_f32 % 2 == 0), "cat_input: float32 output rows must be >= d1 floats, 8-byte aligned");
    CatInputArgs a = {};
    a.xnum = xnum; a.ld_x = ld_x; a.idx = idx; a.B = B; a.n_num = n_num; a.m = m; a.d1 = d1; a.offsets = offsets;
    a.num_wt = num_wt; a.cat_wt = cat_wt; a.cat_bias = cat_bias; a.emb = emb; a.ld_emb = ld_emb; a.log_tiny = log_tiny; a.act = act;
    a.out = (char*)out_split; a.pitch_out = pitch; a.scale_out = ldexpf(1.f, shift); a.out_f32 = out_f32; a.ld_f32 = ld_f32;
    a.overflow = overflow;
    return clava_launch_cat_input(a, (hipStream_t)stream);
}

int clava_launch_cat_input(const CatInputArgs& a, hipStream_t stream) {
    CLAVA_CHECK_ARG(a.B > 0 && a.m > 0 && a.d1 > 0 && a.d1 % 128 == 0 && a.n_num >= 0, "cat_input: bad shape B=%d m=%d d1=%d", a.B, a.m, a.d1);
    CLAVA_CHECK_ARG(a.idx && a.offsets && a.cat_wt && a.cat_bias && a.out && (a.n_num == 0 || (a.xnum && a.num_wt && a.ld_x >= a.n_num)),
                    "cat_input: null operand");
    const int groups = (a.B + ROWS - 1) / ROWS;
    const int blocks = (groups + 3) / 4 < 4096 ? (groups + 3) / 4 : 4096;
    // algorithmic traffic: x and the indices in, the layer's split-row output out (the weight columns are L2 hits)
    ProfScope prof(PROF_CAT_INPUT, 2.0 * a.B * (double)a.d1 * (a.n_num + a.m), 4.0 * a.B * ((double)a.n_num + a.m + a.d1), stream);
    hipLaunchKernelGGL(cat_input_kernel, dim3(blocks), dim3(256), 0, stream, a);
    CLAVA_CHECK_LAUNCH("clava_cat_input");
    return CLAVA_OK;
}
