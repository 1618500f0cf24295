// Split-f16 ("f16x3") GEMM: f32-accurate products on the f16 matrix cores (gemm_f16x3.hip).
//
// A float32 value v is carried as two float16 planes  hi = f16(v * 2^shift),  lo = f16(v * 2^shift - hi),
// i.e. 22 significand bits (hi + lo reproduces v * 2^shift to 2^-22 relative while lo stays a normal float16,
// and to 2^-25 / 2^shift absolute below that).  A product a*w is evaluated as
//     a_hi*w_hi + a_hi*w_lo + a_lo*w_hi
// on v_mfma_f32_16x16x32_f16 with a float32 accumulator; the dropped a_lo*w_lo term is <= 2^-22 of the product.
// The shifts are per-tensor powers of two chosen by the host (so that the largest magnitude sits near 2^6, a factor
// 1000 below the float16 limit 65504; float16 subnormals are honoured by the conversions and by the MFMA, so small
// low planes degrade gracefully); the epilogue undoes them exactly.  A kernel that sees an
// output beyond the float16 range raises the caller's overflow flag: the host then reruns on the float32 kernels.
//
// Storage ("split rows"): row-major, the column count padded to a multiple of 32 (Kp); each 32-column block of a
// row is 128 bytes: 32 hi halves then 32 lo halves.  Row pitch >= 4*Kp bytes (what float32 would take).  Pad
// columns hold zeros.
#pragma once
#include "common.h"

struct SplitGemmArgs {
    const char* A;      // activations, split rows [M][Kp]
    const char* W;      // weights, split rows [N][Kp] (an nn.Linear weight, or its transpose for the input gradient)
    int M, N, Kp;       // Kp % 32 == 0
    int K;              // true contraction length (<= Kp), for the algorithmic FLOP count only
    long long pitchA, pitchW;   // bytes per row
    const float* bias;  // [N] or null
    const float* bias2; // [N] or null (the broadcast time-embedding row)
    int act;            // CLAVA_ACT_NONE / RELU / LEAKY
    const char* mask;   // split rows [M][Np]: multiply by (mask value > 0 ? 1 : mask_slope); or null
    long long pitchMask;
    float mask_slope;
    float scale_acc;    // 2^-(shift of A + shift of W): accumulator -> true value
    float alpha;        // final scale of the true value
    float scale_out;    // 2^(shift of the output planes)
    char* out_split;    // split rows [M][Np], Np = round_up(N, 32); or null
    long long pitchOut;
    float* out_f32;     // [M][N] (ldc) or null
    int ldc;
    int* overflow;      // set to 1 when an output magnitude exceeds the float16 range; or null
    int tiles_m, tiles_n;
    // weight-gradient products: the contraction (batch) axis is cut into `slabs` ranges of kb_per_slab 32-deep blocks;
    // blockIdx.y = slab, whose partial product goes to out_f32 + slab * slab_stride floats.  slabs <= 1: one product.
    int slabs = 1, kb_per_slab = 0;
    long long slab_stride = 0;
    // batch-major weight-gradient kernel only: column sums of the A operand (the bias gradient sum_b dy[b][n]) go to
    // out_colsum[slab * slab_stride + n], scaled by scale_colsum (2^-shift of dy); or null
    float* out_colsum = nullptr;
    float scale_colsum = 0.f;
    // Activation-derivative masks as BITS (the guidance chain): a forward product with `bits_out` also writes, per row, one
    // byte per 8 output columns whose bit r says "stored hi-plane value of column 8*g + r is > 0"; the input-gradient
    // product reads `mask_bits` instead of the activation's planes (`mask`): 1/32 of the bytes (reading the hi plane alone
    // still pulls whole 128-byte lines, hi | lo, from memory).  Rows pitchBits bytes apart, >= round_up(N, 32) / 8.
    // Only tile configurations whose lanes hold 8 consecutive columns take these (the launcher picks among those).
    unsigned char* bits_out = nullptr;
    const unsigned char* mask_bits = nullptr;
    long long pitchBits = 0;
};

// cfg < 0: pick the tile configuration for the shape; otherwise force configuration `cfg` (tools/bench_f16x3.py)
int clava_launch_gemm_f16x3(SplitGemmArgs g, int cfg, hipStream_t stream);
int clava_gemm_f16x3_config_count();
// products with at most 16 outputs and a float32 result (heads, the last input-gradient product): a memory-bound kernel that
// reads the split rows once (narrow.hip); clava_launch_gemm_f16x3 routes to it when the shape qualifies (CLAVA_NARROW_SPLIT=0:
// never)
bool clava_narrow_split_ok(const SplitGemmArgs& g);
int clava_launch_narrow_split(const SplitGemmArgs& g, hipStream_t stream);
struct GaussStepArgs;
int clava_launch_narrow_split_gauss(const SplitGemmArgs& g, const GaussStepArgs& ga, hipStream_t stream);
// weight-gradient product from batch-major operands (gemm_f16x3_tn.hip): A = dy [Kp rows = batch][M = n_out columns],
// W = x [Kp][N = k_in], out_f32[n_out][k_in] (slabs over the batch axis as above)
int clava_launch_gemm_f16x3_tn(SplitGemmArgs g, hipStream_t stream);
// transposed: the packed matrix [rows, cols] is the transpose of the stored one (element (r, c) = src[c * ld + r])
int clava_launch_split_pack(const float* src, long long ld, long long rows, int cols, int shift, char* dst,
                            long long pitch, int* overflow, hipStream_t stream, bool transposed = false);
// dst[cols][rows] = src[rows][cols]^T, both split rows (planes are moved, not re-rounded; pad rows/columns become zeros)
int clava_launch_split_transpose(const char* src, long long pitch_src, long long rows, int cols, char* dst, long long pitch_dst,
                                 hipStream_t stream);
int clava_launch_split_unpack(const char* src, long long pitch, long long rows, int cols, int shift, float* dst,
                              long long ld, hipStream_t stream);
