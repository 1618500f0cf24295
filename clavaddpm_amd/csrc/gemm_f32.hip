// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32) with fused prologue/epilogue.
//
// Replaces every nn.Linear of the reference hot path: MLPDiffusion (tab_ddpm/modules.py:444-449,
// 153-158, 191), Classifier (pipeline_utils.py:589, 597-623) and their autograd backward
// (scripts/train.py:35-38; cond_fn at pipeline_utils.py:853-864).
//
// Numerics: the f32-input MFMA is a k-ordered fmaf chain (exact f32, no reduced-precision path),
// so results differ from a CPU sgemm only by summation order (~1e-7 relative).
//
// Structure: 256-thread workgroup = 4 waves (2x2), workgroup tile BMxBN (128x128 or 64x64), BK = 32,
// register-staged global->LDS copy into a double-buffered LDS image (one barrier per K tile; the next
// tile's global loads are in flight while the current tile is multiplied).  K-major operands are kept
// as [row][BK+4] so that each lane fetches four consecutive k of its row with one ds_read_b128 (the +4
// pad makes the 16 rows of a b128 lane group fall on distinct 16-B slots); MN-major operands are kept
// as [k][rows+4] and read one dword per MFMA (consecutive lanes -> consecutive banks).
// The k order inside an 8-wide chunk is permuted (lane half h supplies k = 4h + j for MFMA j); both
// operands use the same permutation, so the product is unchanged.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int PAD = 4;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    int lda, ldb, ldc;
    int k_per_split;
    long slab;
    int tiles_m, tiles_n;
    GemmEpi e;
};

// ---- staging: global -> registers -> LDS ------------------------------------------------------------
// K-major source S[rows, K] (ld): LDS image T[ROWS][BK+PAD].
template <int ROWS, bool VEC>
__device__ __forceinline__ void load_kmajor(f32x4 (&reg)[ROWS / 32], const float* __restrict__ S, int ld, int R,
                                            int r0, int k0, int k_end, const float* __restrict__ sc,
                                            const float* __restrict__ sh, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int s = tid + 256 * i;
        const int row = r0 + (s >> 3);
        const int k = k0 + ((s & 7) << 2);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < R) {
            const float* p = S + (size_t)row * ld + k;
            if (VEC && k + 3 < k_end) {
                v = *reinterpret_cast<const f32x4*>(p);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < k_end) v[e] = p[e];
            }
            if (sc != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < k_end) v[e] = v[e] * sc[k + e] + sh[k + e];
            }
        }
        reg[i] = v;
    }
}

template <int ROWS>
__device__ __forceinline__ void store_kmajor(float* T, const f32x4 (&reg)[ROWS / 32], int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int s = tid + 256 * i;
        *reinterpret_cast<f32x4*>(T + (s >> 3) * (BK + PAD) + ((s & 7) << 2)) = reg[i];
    }
}

// MN-major source S[K, rows] (ld): LDS image T[BK][ROWS+PAD].
template <int ROWS, bool VEC>
__device__ __forceinline__ void load_mnmajor(f32x4 (&reg)[ROWS / 32], const float* __restrict__ S, int ld, int R,
                                             int r0, int k0, int k_end, int tid) {
    constexpr int Q = ROWS / 4;  // float4 slots per k row
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int s = tid + 256 * i;
        const int k = k0 + s / Q;
        const int c = r0 + ((s % Q) << 2);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < k_end) {
            const float* p = S + (size_t)k * ld + c;
            if (VEC && c + 3 < R) {
                v = *reinterpret_cast<const f32x4*>(p);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < R) v[e] = p[e];
            }
        }
        reg[i] = v;
    }
}

template <int ROWS>
__device__ __forceinline__ void store_mnmajor(float* T, const f32x4 (&reg)[ROWS / 32], int tid) {
    constexpr int Q = ROWS / 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int s = tid + 256 * i;
        *reinterpret_cast<f32x4*>(T + (s / Q) * (ROWS + PAD) + ((s % Q) << 2)) = reg[i];
    }
}

// ---- fragments: LDS -> one f32x4 per 32-row MFMA tile = the operand of 4 consecutive MFMAs -------------
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ f32x4 frag(const float* T, int row, int kc, int h) {
    if (KMAJOR) {
        return *reinterpret_cast<const f32x4*>(T + row * (BK + PAD) + kc * 8 + 4 * h);
    } else {
        const float* p = T + (kc * 8 + 4 * h) * (ROWS + PAD) + row;
        f32x4 v;
        v[0] = p[0];
        v[1] = p[ROWS + PAD];
        v[2] = p[2 * (ROWS + PAD)];
        v[3] = p[3 * (ROWS + PAD)];
        return v;
    }
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case CLAVA_ACT_RELU: return v > 0.f ? v : 0.f;
        case CLAVA_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        case CLAVA_ACT_SILU: return v / (1.f + expf(-v));
        default: return v;
    }
}

template <bool AK, bool BKM, int BM, int BN, bool VEC>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64;        // 32x32 MFMA tiles per wave, wave tile (BM/2)x(BN/2)
    constexpr int A_SZ = AK ? BM * (BK + PAD) : BK * (BM + PAD);
    constexpr int B_SZ = BKM ? BN * (BK + PAD) : BK * (BN + PAD);
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: blocks that share an XCD (equal blockIdx % 8) take a contiguous run of tiles,
    // i.e. neighbouring column tiles of the same A row panel (bijective for any tile count).
    const int nt = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nt >> 3, rem = nt & 7;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int m0 = (tile / g.tiles_n) * BM;
    const int n0 = (tile % g.tiles_n) * BN;

    const int k_begin = blockIdx.y * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    const int nk = (k_end - k_begin + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[BM / 32], rb[BN / 32];
    auto gload = [&](int kt) {
        const int k0 = k_begin + kt * BK;
        if (AK) load_kmajor<BM, VEC>(ra, g.A, g.lda, g.M, m0, k0, k_end, g.e.a_scale, g.e.a_shift, tid);
        else load_mnmajor<BM, VEC>(ra, g.A, g.lda, g.M, m0, k0, k_end, tid);
        if (BKM) load_kmajor<BN, VEC>(rb, g.B, g.ldb, g.N, n0, k0, k_end, nullptr, nullptr, tid);
        else load_mnmajor<BN, VEC>(rb, g.B, g.ldb, g.N, n0, k0, k_end, tid);
    };
    auto sstore = [&](int buf) {
        float* As = smem + buf * (A_SZ + B_SZ);
        float* Bs = As + A_SZ;
        if (AK) store_kmajor<BM>(As, ra, tid); else store_mnmajor<BM>(As, ra, tid);
        if (BKM) store_kmajor<BN>(Bs, rb, tid); else store_mnmajor<BN>(Bs, rb, tid);
    };

    if (nk > 0) {
        gload(0);
        sstore(0);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const float* As = smem + cur * (A_SZ + B_SZ);
        const float* Bs = As + A_SZ;
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = frag<AK, BM>(As, wr * (BM / 2) + i * 32 + r, kc, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = frag<BKM, BN>(Bs, wc * (BN / 2) + j * 32 + r, kc, h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const GemmEpi& e = g.e;
    float* C = g.C + (size_t)blockIdx.y * g.slab;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * (BN / 2) + j * 32 + r;
        if (col >= g.N) continue;
        const float bias = e.bias ? e.bias[col] : 0.f;
        const float cs = e.col_scale ? e.col_scale[col] : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m0 + wr * (BM / 2) + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (row >= g.M) continue;
                float v = acc[i][j][reg] + bias;
                if (e.rowadd) v += e.rowadd[(size_t)row * e.ldr + col];
                v = apply_act(v, e.act);
                if (e.mask_src) v *= (e.mask_src[(size_t)row * e.ldm + col] > 0.f ? 1.f : e.mask_slope);
                v *= cs;
                v *= e.alpha;
                C[(size_t)row * g.ldc + col] = v;
            }
        }
    }
}

template <bool AK, bool BKM, int BM, int BN>
int launch_tiles(const GemmArgs& g, bool vec, int splitk, hipStream_t stream) {
    dim3 grid(g.tiles_m * g.tiles_n, splitk, 1);
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, BM, BN, true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, BM, BN, false>), grid, dim3(256), 0, stream, g);
    CLAVA_CHECK_LAUNCH("clava_gemm_f32");
    return CLAVA_OK;
}

template <bool AK, bool BKM>
int launch_layout(GemmArgs& g, bool vec, int splitk, hipStream_t stream) {
    const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * splitk;
    if (g.N > 64 && g.M > 64 && t128 >= 256) {
        g.tiles_m = (g.M + 127) / 128;
        g.tiles_n = (g.N + 127) / 128;
        return launch_tiles<AK, BKM, 128, 128>(g, vec, splitk, stream);
    }
    g.tiles_m = (g.M + 63) / 64;
    g.tiles_n = (g.N + 63) / 64;
    return launch_tiles<AK, BKM, 64, 64>(g, vec, splitk, stream);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int clava_launch_gemm_f32(const float* A, int lda, bool a_kmajor, const float* B, int ldb, bool b_kmajor, float* C,
                          int ldc, int M, int N, int K, const GemmEpi& epi, int splitk, long slab,
                          hipStream_t stream) {
    CLAVA_CHECK_ARG(A && B && C, "gemm: null operand");
    CLAVA_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty shape M=%d N=%d K=%d", M, N, K);
    CLAVA_CHECK_ARG(splitk >= 1 && splitk <= 64, "gemm: splitk=%d out of range", splitk);
    CLAVA_CHECK_ARG(lda >= (a_kmajor ? K : M) && ldb >= (b_kmajor ? K : N) && ldc >= N, "gemm: leading dimension too small");
    CLAVA_CHECK_ARG(!(epi.a_scale && !a_kmajor), "gemm: A prologue needs a K-major A");
    CLAVA_CHECK_ARG(!(splitk > 1 && (epi.bias || epi.rowadd || epi.act || epi.mask_src || epi.col_scale)),
                    "gemm: split-K writes plain partial products");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    int kps = (K + splitk - 1) / splitk;
    g.k_per_split = (kps + BK - 1) / BK * BK;   // split boundaries stay multiples of BK (vector loads stay aligned)
    g.slab = slab;
    g.e = epi;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    if (a_kmajor && b_kmajor) return launch_layout<true, true>(g, vec, splitk, stream);
    if (a_kmajor && !b_kmajor) return launch_layout<true, false>(g, vec, splitk, stream);
    if (!a_kmajor && !b_kmajor) return launch_layout<false, false>(g, vec, splitk, stream);
    clava_set_error("gemm: layout (A MN-major, B K-major) is not used by any caller");
    return CLAVA_ERR_ARG;
}
