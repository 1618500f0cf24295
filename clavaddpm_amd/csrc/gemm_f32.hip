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
// All loads are branch-free: out-of-range slots read a clamped (valid) address and are zeroed by a select,
// so the compiler can issue every load of a tile back to back and interleave them with the MFMAs.
//
// K-major source S[rows, K] (ld): LDS image T[ROWS][BK+PAD].  Slot s = tid + 256*i covers row s/8, k 4*(s%8)..+3.
template <int ROWS>
struct KMajorSrc {
    const float* p[ROWS / 32];   // row base (clamped row)
    bool rv[ROWS / 32];          // row in range
    int kq;                      // this thread's k offset inside a tile
    f32x4 s4, h4;                // A-prologue scale / shift of the tile in flight (AFF only)
    __device__ __forceinline__ void init(const float* S, int ld, int R, int r0, int tid) {
        kq = (tid & 7) << 2;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int row = r0 + ((tid + 256 * i) >> 3);
            rv[i] = row < R;
            p[i] = S + (size_t)min(row, R - 1) * ld;
        }
    }
    // raw loads only (clamped addresses); validity and the affine prologue are applied in store(), so nothing
    // consumes the loaded registers until the data is needed and the loads stay in flight behind the MFMAs
    template <bool VEC, bool AFF>
    __device__ __forceinline__ void load(f32x4 (&reg)[ROWS / 32], int k0, int k_end, const float* __restrict__ sc,
                                         const float* __restrict__ sh) {
        const int k = k0 + kq;
        if (AFF) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = min(k + e, k_end - 1);
                s4[e] = sc[kk];
                h4[e] = sh[kk];
            }
        }
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            if (VEC) {   // K % 4 == 0: a float4 is entirely inside or entirely outside
                reg[i] = *reinterpret_cast<const f32x4*>(p[i] + (k < k_end ? k : 0));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) reg[i][e] = p[i][k + e < k_end ? k + e : 0];
            }
        }
    }
    template <bool AFF>
    __device__ __forceinline__ void store(float* T, const f32x4 (&reg)[ROWS / 32], int k0, int k_end, int tid) const {
        const int k = k0 + kq;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int s = tid + 256 * i;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = rv[i] && (k + e < k_end);
                v[e] = ok ? (AFF ? reg[i][e] * s4[e] + h4[e] : reg[i][e]) : 0.f;
            }
            *reinterpret_cast<f32x4*>(T + (s >> 3) * (BK + PAD) + ((s & 7) << 2)) = v;
        }
    }
};

// MN-major source S[K, rows] (ld): LDS image T[BK][ROWS+PAD].  Slot s covers k s/Q, columns 4*(s%Q)..+3.
template <int ROWS>
struct MNMajorSrc {
    static constexpr int Q = ROWS / 4;
    const float* S;
    int ld, R;
    int c[ROWS / 32], kk[ROWS / 32];
    __device__ __forceinline__ void init(const float* S_, int ld_, int R_, int r0, int tid) {
        S = S_; ld = ld_; R = R_;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int s = tid + 256 * i;
            kk[i] = s / Q;
            c[i] = r0 + ((s % Q) << 2);
        }
    }
    template <bool VEC, bool AFF>
    __device__ __forceinline__ void load(f32x4 (&reg)[ROWS / 32], int k0, int k_end, const float*, const float*) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int k = k0 + kk[i];
            const float* row = S + (size_t)(k < k_end ? k : 0) * ld;
            if (VEC) {   // R % 4 == 0
                reg[i] = *reinterpret_cast<const f32x4*>(row + (c[i] < R ? c[i] : 0));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) reg[i][e] = row[c[i] + e < R ? c[i] + e : 0];
            }
        }
    }
    template <bool AFF>
    __device__ __forceinline__ void store(float* T, const f32x4 (&reg)[ROWS / 32], int k0, int k_end, int tid) const {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int s = tid + 256 * i;
            const bool kok = k0 + kk[i] < k_end;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (kok && c[i] + e < R) ? reg[i][e] : 0.f;
            *reinterpret_cast<f32x4*>(T + (s / Q) * (ROWS + PAD) + ((s % Q) << 2)) = v;
        }
    }
};

template <bool KMAJOR, int ROWS>
struct SrcOf { using type = KMajorSrc<ROWS>; };
template <int ROWS>
struct SrcOf<false, ROWS> { using type = MNMajorSrc<ROWS>; };

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

// Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); the
// per-register row offsets are compile-time, so their products with the leading dimensions stay scalar.
template <int ACT, int TM, int TN>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[TM][TN], const GemmArgs& g, float* C, int row_base,
                                         int col_base) {
    const GemmEpi& e = g.e;
    const bool has_rowadd = e.rowadd != nullptr, has_mask = e.mask_src != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col_base + j * 32;
        const bool cok = col < g.N;
        const int cc = min(col, g.N - 1);
        const float bias = e.bias ? e.bias[cc] : 0.f;
        const float cs = (e.col_scale ? e.col_scale[cc] : 1.f) * e.alpha;
        float* Cp = C + (size_t)row_base * g.ldc + col;
        const float* Rp = has_rowadd ? e.rowadd + (size_t)row_base * e.ldr + cc : nullptr;
        const float* Mp = has_mask ? e.mask_src + (size_t)row_base * e.ldm + cc : nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int ro = i * 32 + (reg & 3) + 8 * (reg >> 2);
                const bool ok = cok && (row_base + ro < g.M);
                float v = acc[i][j][reg] + bias;
                if (has_rowadd) v += ok ? Rp[(size_t)ro * e.ldr] : 0.f;
                if (ACT == CLAVA_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (ACT == CLAVA_ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
                else if (ACT == CLAVA_ACT_SILU) v = v / (1.f + expf(-v));
                if (has_mask) v *= ((ok ? Mp[(size_t)ro * e.ldm] : 0.f) > 0.f ? 1.f : e.mask_slope);
                v *= cs;
                if (ok) Cp[(size_t)ro * g.ldc] = v;
            }
        }
    }
}

template <bool AK, bool BKM, int BM, int BN, bool VEC, bool AFF>
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

    typename SrcOf<AK, BM>::type srcA;
    typename SrcOf<BKM, BN>::type srcB;
    srcA.init(g.A, g.lda, g.M, m0, tid);
    srcB.init(g.B, g.ldb, g.N, n0, tid);
    f32x4 ra[BM / 32], rb[BN / 32];

    // Pipeline: registers hold tile kt+1 while tile kt is multiplied out of LDS buffer `cur`.  Half way through
    // the tile the registers are written to the other LDS buffer (free since the barrier that ended tile kt-1)
    // and the loads of tile kt+2 are issued, so both sit behind MFMAs; one barrier per tile.
    if (nk > 0) {
        srcA.template load<VEC, AFF>(ra, k_begin, k_end, g.e.a_scale, g.e.a_shift);
        srcB.template load<VEC, false>(rb, k_begin, k_end, nullptr, nullptr);
        srcA.template store<AFF>(smem, ra, k_begin, k_end, tid);
        srcB.template store<false>(smem + A_SZ, rb, k_begin, k_end, tid);
        if (nk > 1) {
            srcA.template load<VEC, AFF>(ra, k_begin + BK, k_end, g.e.a_scale, g.e.a_shift);
            srcB.template load<VEC, false>(rb, k_begin + BK, k_end, nullptr, nullptr);
        }
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
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
            if (kc == BK / 16 - 1 && kt + 1 < nk) {
                float* An = smem + (cur ^ 1) * (A_SZ + B_SZ);
                srcA.template store<AFF>(An, ra, k_begin + (kt + 1) * BK, k_end, tid);
                srcB.template store<false>(An + A_SZ, rb, k_begin + (kt + 1) * BK, k_end, tid);
                if (kt + 2 < nk) {
                    srcA.template load<VEC, AFF>(ra, k_begin + (kt + 2) * BK, k_end, g.e.a_scale, g.e.a_shift);
                    srcB.template load<VEC, false>(rb, k_begin + (kt + 2) * BK, k_end, nullptr, nullptr);
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    float* C = g.C + (size_t)blockIdx.y * g.slab;
    const int row_base = m0 + wr * (BM / 2) + 4 * h;
    const int col_base = n0 + wc * (BN / 2) + r;
    switch (g.e.act) {
        case CLAVA_ACT_RELU: epilogue<CLAVA_ACT_RELU, TM, TN>(acc, g, C, row_base, col_base); break;
        case CLAVA_ACT_LEAKY: epilogue<CLAVA_ACT_LEAKY, TM, TN>(acc, g, C, row_base, col_base); break;
        case CLAVA_ACT_SILU: epilogue<CLAVA_ACT_SILU, TM, TN>(acc, g, C, row_base, col_base); break;
        default: epilogue<CLAVA_ACT_NONE, TM, TN>(acc, g, C, row_base, col_base); break;
    }
}

template <bool AK, bool BKM, int BM, int BN>
int launch_tiles(const GemmArgs& g, bool vec, int splitk, hipStream_t stream) {
    dim3 grid(g.tiles_m * g.tiles_n, splitk, 1);
    const int fam = (AK && BKM ? (g.e.a_scale ? PROF_GEMM_NT_128_BN : PROF_GEMM_NT_128) : (AK ? PROF_GEMM_NN_128 : PROF_GEMM_TN_128)) +
                    (BM == 64 ? 1 : 0);
    ProfScope prof(fam, 2.0 * g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * splitk), stream);
    const bool aff = g.e.a_scale != nullptr;
    if (AK && BKM && aff) {
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, BM, BN, true, AK && BKM>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, BM, BN, false, AK && BKM>), grid, dim3(256), 0, stream, g);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, BM, BN, true, false>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, BM, BN, false, false>), grid, dim3(256), 0, stream, g);
    }
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
    // vector (16-B) staging loads need aligned bases and leading dimensions, and extents that are multiples of 4
    // along the contiguous axis of each operand (so a float4 is entirely inside or outside the matrix)
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                     ((a_kmajor ? K : M) % 4 == 0) && ((b_kmajor ? K : N) % 4 == 0);
    if (a_kmajor && b_kmajor) return launch_layout<true, true>(g, vec, splitk, stream);
    if (a_kmajor && !b_kmajor) return launch_layout<true, false>(g, vec, splitk, stream);
    if (!a_kmajor && !b_kmajor) return launch_layout<false, false>(g, vec, splitk, stream);
    clava_set_error("gemm: layout (A MN-major, B K-major) is not used by any caller");
    return CLAVA_ERR_ARG;
}
