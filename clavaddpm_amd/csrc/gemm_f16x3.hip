// Split-f16 GEMM on the gfx950 f16 matrix cores: float32-accurate nn.Linear products at (f16 MFMA rate) / 3.
//
// Replaces the nn.Linear products of the sampling loop - MLPDiffusion.forward (tab_ddpm/modules.py:451-460),
// Classifier.forward (pipeline_utils.py:625-630) and the input-gradient chain behind cond_fn
// (pipeline_utils.py:853-864) - where weights are constant for a whole chain and can be packed once.
// Number format and accuracy argument: split.h.  The float32-input MFMA kernel (gemm_f32.hip) runs at 1/16 of the
// f16 MFMA rate; three f16 MFMAs per product leave a 16/3 headroom at the same (f32-level) accuracy.
//
// Kernel: out[m][n] = epi( sum_k A[m][k] * W[n][k] ), computed as D = Wtile (rows n) x Atile^T (columns m) so that a
// lane's four accumulator registers are four consecutive n of one output row m; W rows are staged permuted so that an
// even/odd tile pair gives a lane eight consecutive n: packed 16-byte stores of each plane, full 128-byte lines per row.
//   workgroup 512 threads = 8 waves (WN along n x 8/WN along m), tile BN x BM, one 32-deep k block per stage
//   (128 B per row: hi plane | lo plane), two LDS stages of (BN + BM) * 128 B, filled by global_load_lds_dwordx4
//   (16 B per lane, one wave-instruction = 8 rows x 128 B); the next stage's loads are issued before the current
//   stage's MFMAs and are awaited only at the stage's closing barrier.
//   LDS image: row-major [row][128 B], 16-B chunk index XOR ((row >> 1) & 7): every ds_read_b128 lane group of a
//   fragment read (16 rows x one chunk) lands on 16 distinct 16-B bank slots.  The permutation is applied to the
//   SOURCE address of the LDS-DMA (its LDS side is lane-linear) and to the fragment reads.
//   v_mfma_f32_16x16x32_f16: lane l supplies row (l & 15), k = 8 * (l >> 4) .. +7 of both operands.
//   BM is chosen per shape so that the tile count fills the 256 CUs in whole rounds (M = 20000 sampling batches).
#include "split.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#ifdef CLAVA_GEMM_STAMPS
// Diagnostic build only (tools/build_variant.py stamps gemm_f16x3.hip -DCLAVA_GEMM_STAMPS; never in the shipped library):
// per workgroup x wave {loop cycles, cycles parked at the closing wait + barrier, loop time in 100 MHz ticks, k blocks}
__device__ unsigned long long g_gemm_stamps[4096 * 8 * 8];
extern "C" int clava_debug_gemm_stamps_(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * n);
}
#endif

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

template <int BN, int BM, int WN, int NS, int NW = 8, int OCC = 1>
__global__ void __launch_bounds__(64 * NW, OCC) gemm_f16x3_kernel(SplitGemmArgs g) {
    constexpr int WM = NW / WN;                               // NW waves: WN along n x WM along m (8 / NW workgroups share a CU)
    constexpr int TN = BN / (16 * WN), TM = BM / (16 * WM);   // 16x16 MFMA tiles per wave
    constexpr int G = (BN + BM) / 8;                          // 8-row staging groups per stage (one LDS-DMA each)
    constexpr int Q = (G + NW - 1) / NW;                      // groups per wave (the last one may be absent)
    constexpr int STAGE = (BN + BM) * 128;
    constexpr int SPREAD = TN >= 2 ? TN / 2 : 1;             // MFMA groups over which a stage's LDS-DMA issue is spread
    static_assert(NS >= 2 && NS * STAGE <= 160 * 1024, "LDS stages");
    static_assert(TN * 16 * WN == BN && TM * 16 * WM == BM && (BN % 16) == 0 && (BM % 16) == 0, "tile shape");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
#ifdef CLAVA_GEMM_STAMPS
    const unsigned long long st_k0 = __builtin_amdgcn_s_memrealtime();
#endif

    // XCD-aware tile order (blocks with equal blockIdx % 8 share an XCD and take a contiguous run of tiles: the
    // column tiles of one activation row panel); bijective for any tile count
    const int nt = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nt >> 3, rem = nt & 7;
    const int tile = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (bid >> 3);
    const int m0 = (tile / g.tiles_n) * BM;
    const int n0 = (tile % g.tiles_n) * BN;
    int nkb = g.Kp >> 5;
    long long kbyte0 = 0;                                     // byte offset of this slab's first k block inside a row
    if (g.slabs > 1) {
        const int kb0 = blockIdx.y * g.kb_per_slab;
        nkb = min(nkb - kb0, g.kb_per_slab);
        kbyte0 = (long long)kb0 * 128;
        g.out_f32 += (long long)blockIdx.y * g.slab_stride;
    }

    // ---- bias sums of this tile's columns (tile-local order, zero beyond N), read back by the epilogue
    __shared__ __attribute__((aligned(16))) float bias_s[BN];
    for (int c = tid; c < BN; c += 64 * NW) {
        const int n = n0 + c;
        float b = 0.f;
        if (n < g.N) {
            if (g.bias) b += g.bias[n];
            if (g.bias2) b += g.bias2[n];
        }
        bias_s[c] = b;
    }

    // ---- staging sources: lane i of a group fills LDS row 8*group + (i >> 3), chunk slot (i & 7).  Every wave issues
    // exactly Q DMAs per stage (a wave without a last group re-issues its previous one: same source, same destination)
    // so that the counted vmcnt waits below mean the same thing in every wave.
    const char* src[Q];
    int dst[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int qq = (wave + NW * q < G) ? q : q - 1;
        const int r = (wave + NW * qq) * 8 + (lane >> 3);
        const int sw = ((lane & 7) ^ ((r >> 1) & 7)) << 4;
        dst[q] = qq * NW * 1024;
        if (r < BN) {
            // LDS row 32*p + 16*b + 4*q + rr holds output column 32*p + 8*q + 4*b + rr (epilogue: 8 consecutive columns per lane)
            const int pr = (r & ~31) | (((r >> 2) & 3) << 3) | (((r >> 4) & 1) << 2) | (r & 3);
            const int gr = min(n0 + pr, g.N - 1);
            src[q] = g.W + (long long)gr * g.pitchW + sw + kbyte0;
        } else {
            const int gr = min(m0 + (r - BN), g.M - 1);
            src[q] = g.A + (long long)gr * g.pitchA + sw + kbyte0;
        }
    }
    char* const st = smem + wave * 1024;

    // ---- fragment read offsets (same for every 16-row tile: the swizzle only uses row bits 1..3)
    const int lrow = lane & 15, kq = lane >> 4, sw = (lrow >> 1) & 7;
    const int off_hi = lrow * 128 + ((kq ^ sw) << 4);
    const int off_lo = lrow * 128 + (((4 + kq) ^ sw) << 4);
    const char* const frW = smem + wn * (BN / WN) * 128;
    const char* const frA = smem + BN * 128 + wm * (BM / WM) * 128;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- NS-stage pipeline: stages kb+1 .. kb+NS-1 are in flight while stage kb is multiplied.  The waits are counted
    // (`vmcnt(Q*(NS-2))` = everything but the NS-2 youngest stages has landed) and the barriers raw: __syncthreads()
    // would drain the whole DMA queue (vmcnt(0)) at every k block.
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) {
        const int ko = min(s0, nkb - 1) * 128;
#pragma unroll
        for (int q = 0; q < Q; ++q) glds16(src[q] + ko, st + s0 * STAGE + dst[q]);
    }
    if constexpr (NS == 2) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(Q * (NS - 2)) : "memory");

    int cur = 0, nxt = (NS - 1) * STAGE;                     // byte offsets of the stage read now / filled now
#ifdef CLAVA_GEMM_STAMPS
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_bar = 0, st_a = 0, st_b = 0;
#endif
    for (int kb = 0; kb < nkb; ++kb) {
        // branch-free body: in the tail (no block NS-1 ahead) the last block is fetched again into the stage nobody reads any
        // more - one basic block lets the compiler place counted lgkmcnt waits instead of draining the LDS queue at a join
        const int ko = min(kb + NS - 1, nkb - 1) * 128;
        // fragment reads in the order the first MFMA group consumes them (wl x ah[j], then wh x al[j], wh x ah[j]): LDS
        // returns data in issue order, so the first MFMAs start after two reads instead of after all 2*TM + 2
        h8 ah[TM], al[TM];
        h8 wl = *reinterpret_cast<const h8*>(frW + cur + off_lo);
#pragma unroll
        for (int j = 0; j < TM; ++j) ah[j] = *reinterpret_cast<const h8*>(frA + cur + j * 2048 + off_hi);
        // W fragments are read one MFMA group ahead: the reads of group i+1 are in flight while group i multiplies
        h8 wh = *reinterpret_cast<const h8*>(frW + cur + off_hi);
#pragma unroll
        for (int j = 0; j < TM; ++j) al[j] = *reinterpret_cast<const h8*>(frA + cur + j * 2048 + off_lo);
#ifdef CLAVA_GEMM_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            h8 wh_n = wh, wl_n = wl;
            if (i + 1 < TN) {
                wh_n = *reinterpret_cast<const h8*>(frW + cur + (i + 1) * 2048 + off_hi);
                wl_n = *reinterpret_cast<const h8*>(frW + cur + (i + 1) * 2048 + off_lo);
            }
            // the DMA of the stage NS-1 blocks ahead is issued behind the fragment reads of the first half of the wave's
            // MFMA groups: its issue cost (~100 cycles per instruction) hides under MFMAs
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if ((q * SPREAD) / Q == i) glds16(src[q] + ko, st + nxt + dst[q]);
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[j], acc[i][j], 0, 0, 0);
            wh = wh_n; wl = wl_n;
        }
        // stage kb+1 must have landed (this wave's part) and every wave must be done reading stage kb before the next
        // iteration overwrites it; in the tail (nothing issued this iteration) the queue is simply drained
#ifdef CLAVA_GEMM_STAMPS
        st_bar += st_b - st_a;                                // previous block's wait (both values have long arrived)
        st_a = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (NS == 2) {
            __syncthreads();      // two stages: the whole DMA queue is this stage; the compiler keeps rotating MFMAs across it
#ifdef CLAVA_GEMM_STAMPS
            st_b = __builtin_amdgcn_s_memtime();
#endif
        } else {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(Q * (NS - 2)) : "memory");
        }
        nxt = cur;
        cur = (cur + STAGE == NS * STAGE) ? 0 : cur + STAGE;
    }

#ifdef CLAVA_GEMM_STAMPS
    if (lane == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
        unsigned long long* o = g_gemm_stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - st_c0;
        o[1] = st_bar + (st_b - st_a);
        o[2] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[3] = nkb;
        o[4] = st_k0;
        o[5] = st_r0;
        o[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the tail's refetches must have landed before the workgroup's LDS is released

    // ---- epilogue.  W rows are staged permuted (see the staging sources): within every 32-column block, MFMA tile
    // b (0/1) row 4*q + r holds column 8*q + 4*b + r, so a lane's accumulators of an even/odd tile pair are 8
    // CONSECUTIVE columns of one output row: one 16-byte store per plane, and the four lanes of a row fill the row's
    // whole 128-byte block (64 B hi | 64 B lo).  Configurations with one tile per wave store 4 columns (8 bytes).
    // The element arithmetic is branch-free (measured: with per-element tests of the argument flags and one bias load +
    // wait per column the epilogue of a 256 x 320 tile took 15 us of a 96 us kernel, instruction-bound): the bias sums
    // come from LDS (staged at kernel start), the activation is max(x, slope * x), the overflow test an integer maximum of
    // |value| bit patterns, and the uniform flags (mask / which outputs) select one of six straight-line variants.
    constexpr int EW = (TN % 2 == 0) ? 8 : 4;               // columns per lane per store
    constexpr int EP = (TN % 2 == 0) ? TN / 2 : TN;         // stores per output row per plane
    const int Np = (g.N + 31) & ~31;
    const bool ragged = (g.N & 31) != 0 && n0 + BN > g.N;     // (uniform) only the last column tile has pad columns to zero
    const int m_lane = m0 + wm * (BM / WM) + lrow;
    const float slope = g.act == CLAVA_ACT_RELU ? 0.f : (g.act == CLAVA_ACT_LEAKY ? 0.01f : 1.f);
    // float32 rows that start 16-byte aligned take vector stores (nq is a multiple of EW)
    const bool f32_vec = g.out_f32 && (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.out_f32) & 15) == 0;
    // Overflow test: the float16 hi plane of a value beyond the float16 range is +-inf (NaN for NaN), and inf * 0 = NaN:
    // ovf accumulates hi * 0 with a packed fma, half an instruction per element, and stays 0 unless a plane left the range.
    f32x2 ovf = {0.f, 0.f};

    // Per element the epilogue is VALU work (~13 instructions in the first straight-line version: 8.5 us of a 256 x 320
    // tile's 7.4 + 1.5 us of prologue / epilogue at 160 elements per lane): it now runs on pairs of columns with the packed
    // float32 instructions (v_pk_fma / v_pk_mul / v_pk_add) and v_cvt_pk_f16_f32, and, when only split rows go out and
    // alpha == 1 (FOLD), with the output's power-of-two scale folded into the accumulator scale and the bias - exact, since
    // the activation is positively homogeneous and power-of-two factors commute with rounding: ~5 instructions per element.
    auto epilogue = [&](auto has_mask, auto out_mode, auto fold_t) {
        constexpr int MASKM = decltype(has_mask)::value;       // 0: none, 1: the activation's planes, 2: derivative bits
        constexpr bool MASK = MASKM == 1;
        constexpr bool SPLIT = decltype(out_mode)::value != 1, F32 = decltype(out_mode)::value != 0;
        constexpr bool FOLD = decltype(fold_t)::value;
        const float sa = FOLD ? g.scale_acc * g.scale_out : g.scale_acc;
        const f32x2 sa2 = {sa, sa}, slope2 = {slope, slope}, zero2 = {0.f, 0.f};
#pragma unroll
        for (int ip = 0; ip < EP; ++ip) {
            const int it = wn * TN + (EW == 8 ? 2 * ip : ip);    // n-subtile index inside the workgroup tile
            const int nl = 32 * (it >> 1) + 8 * kq + 4 * (it & 1);
            const int nq = n0 + nl;
            if (nq >= Np) continue;
            f32x2 bs2[EW / 2];
#pragma unroll
            for (int r = 0; r < EW; r += 4) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + nl + r);
                bs2[r / 2] = f32x2{b4[0], b4[1]};
                bs2[r / 2 + 1] = f32x2{b4[2], b4[3]};
                if (FOLD) { bs2[r / 2] *= g.scale_out; bs2[r / 2 + 1] *= g.scale_out; }
            }
            const long long col_off = (long long)(nq >> 5) * 128 + ((nq & 31) << 1);
            const bool inside = nq + EW <= g.N;
            h8 mh8[TM];
            h4 mh4[TM];
            unsigned mb[TM];
            if constexpr (MASKM == 2) {
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int m = min(m_lane + j * 16, g.M - 1);
                    mb[j] = g.mask_bits[(long long)m * g.pitchBits + (nq >> 3)];
                }
            }
            if constexpr (MASK) {                            // all of this column group's mask loads before the first store
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int m = min(m_lane + j * 16, g.M - 1);
                    const char* mp = g.mask + (long long)m * g.pitchMask + col_off;
                    if constexpr (EW == 8) mh8[j] = *reinterpret_cast<const h8*>(mp);
                    else mh4[j] = *reinterpret_cast<const h4*>(mp);
                }
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m_lane + j * 16;
                if (m >= g.M) continue;
                f32x2 v[EW / 2];
#pragma unroll
                for (int p = 0; p < EW / 2; ++p) {
                    const int r = 2 * p;
                    const f32x4& a4 = (EW == 8) ? acc[2 * ip + (r >> 2)][j] : acc[ip][j];
                    const f32x2 a2 = {a4[r & 3], a4[(r & 3) + 1]};
                    const f32x2 x = __builtin_elementwise_fma(a2, sa2, bs2[p]);     // (the scale is a power of two: same bits as mul + add)
                    const f32x2 y = x * slope2;
                    v[p] = f32x2{fmaxf(x[0], y[0]), fmaxf(x[1], y[1])};         // relu (slope 0), leaky relu (0.01), identity (1)
                }
                if constexpr (MASK) {
#pragma unroll
                    for (int r = 0; r < EW; ++r) {
                        const float mv = (EW == 8) ? (float)mh8[j][r] : (float)mh4[j][r];
                        v[r >> 1][r & 1] *= (mv > 0.f) ? 1.f : g.mask_slope;
                    }
                }
                if constexpr (MASKM == 2) {
#pragma unroll
                    for (int r = 0; r < EW; ++r) v[r >> 1][r & 1] *= ((mb[j] >> r) & 1u) ? 1.f : g.mask_slope;
                }
                if constexpr (!FOLD) {
#pragma unroll
                    for (int p = 0; p < EW / 2; ++p) v[p] *= g.alpha;
                }
                if (ragged && !inside) {                     // (uniform `ragged`: elsewhere there are no pad columns to zero)
#pragma unroll
                    for (int r = 0; r < EW; ++r) v[r >> 1][r & 1] = (nq + r < g.N) ? v[r >> 1][r & 1] : 0.f;
                }
                if constexpr (SPLIT) {
                    h2 hi[EW / 2], lo[EW / 2];
#pragma unroll
                    for (int p = 0; p < EW / 2; ++p) {
                        const f32x2 sc = FOLD ? v[p] : v[p] * g.scale_out;
                        hi[p] = __builtin_convertvector(sc, h2);
                        const f32x2 back = __builtin_convertvector(hi[p], f32x2);
                        lo[p] = __builtin_convertvector(sc - back, h2);
                        ovf = __builtin_elementwise_fma(back, zero2, ovf);
                    }
                    char* op = g.out_split + (long long)m * g.pitchOut + col_off;
                    if constexpr (EW == 8) {
                        *reinterpret_cast<h8*>(op) = *reinterpret_cast<const h8*>(hi);
                        *reinterpret_cast<h8*>(op + 64) = *reinterpret_cast<const h8*>(lo);
                        if (g.bits_out) {                // derivative bits of this lane's 8 columns (uniform branch)
                            unsigned b = 0;
#pragma unroll
                            for (int r = 0; r < 8; ++r) b |= (hi[r >> 1][r & 1] > (_Float16)0.f ? 1u : 0u) << r;
                            g.bits_out[(long long)m * g.pitchBits + (nq >> 3)] = (unsigned char)b;
                        }
                    } else {
                        *reinterpret_cast<h4*>(op) = *reinterpret_cast<const h4*>(hi);
                        *reinterpret_cast<h4*>(op + 64) = *reinterpret_cast<const h4*>(lo);
                    }
                }
                if constexpr (F32) {
                    float* cp = g.out_f32 + (long long)m * g.ldc + nq;
                    if (f32_vec && inside) {                 // 16-byte stores: a lane's EW consecutive columns of one row
#pragma unroll
                        for (int r = 0; r < EW; r += 4) *reinterpret_cast<f32x4*>(cp + r) = f32x4{v[r / 2][0], v[r / 2][1], v[r / 2 + 1][0], v[r / 2 + 1][1]};
                    } else {
#pragma unroll
                        for (int r = 0; r < EW; ++r)
                            if (nq + r < g.N) cp[r] = v[r >> 1][r & 1];
                    }
                }
            }
        }
    };
    using std::integral_constant;
    const int mode = g.out_split ? (g.out_f32 ? 2 : 0) : 1;
    const bool fold = mode == 0 && g.alpha == 1.f;
    auto run = [&](auto mm) {
        if (fold) epilogue(mm, integral_constant<int, 0>{}, integral_constant<bool, true>{});
        else if (mode == 0) epilogue(mm, integral_constant<int, 0>{}, integral_constant<bool, false>{});
        else if (mode == 1) epilogue(mm, integral_constant<int, 1>{}, integral_constant<bool, false>{});
        else epilogue(mm, integral_constant<int, 2>{}, integral_constant<bool, false>{});
    };
    if constexpr (EW == 8) {
        if (g.mask_bits) run(integral_constant<int, 2>{});
        else if (g.mask) run(integral_constant<int, 1>{});
        else run(integral_constant<int, 0>{});
    } else {
        if (g.mask) run(integral_constant<int, 1>{});
        else run(integral_constant<int, 0>{});
    }
    const float ovs = ovf[0] + ovf[1];
    if (!(ovs == ovs) && g.overflow) *g.overflow = 1;
#ifdef CLAVA_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the stores have left the wave
    if (lane == 0 && blockIdx.x < 4096 && blockIdx.y == 0)
        g_gemm_stamps[((size_t)blockIdx.x * 8 + wave) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}

// float32 [rows][cols] (ld) -> split rows; one thread per 4 consecutive columns
// element (row, col) of the source is src[row * sr + col * sc]: (ld, 1) packs a matrix as it stands, (1, ld) its transpose
__global__ void __launch_bounds__(256) split_pack_kernel(const float* __restrict__ src, long long sr, long long sc, long long rows, int cols,
                                                         int groups, float scale, char* __restrict__ dst, long long pitch,
                                                         int* overflow) {
    const long long total = rows * groups;
    bool ovf = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / groups;
        const int c0 = (int)(i - row * groups) << 2;
        h4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = (c0 + r < cols) ? src[row * sr + (c0 + r) * sc] * scale : 0.f;
            ovf |= !(fabsf(s) <= 65504.f);
            hi[r] = (_Float16)s;
            lo[r] = (_Float16)(s - (float)hi[r]);
        }
        char* op = dst + row * pitch + (long long)(c0 >> 5) * 128 + ((c0 & 31) << 1);
        *reinterpret_cast<h4*>(op) = hi;
        *reinterpret_cast<h4*>(op + 64) = lo;
    }
    if (ovf && overflow) *overflow = 1;
}

// Split rows [rows][cols] -> split rows [cols][rows] (the batch-major operands of the weight-gradient products).
// One workgroup moves a 64 x 64 tile of both planes through LDS: 128-byte reads along the source rows, 128-byte writes
// along the destination rows; elements beyond the source's rows / columns are written as zeros (pad of the contraction).
__global__ void __launch_bounds__(256) split_transpose_kernel(const char* __restrict__ src, long long pitch_src, long long rows, int cols,
                                                              char* __restrict__ dst, long long pitch_dst) {
    __shared__ unsigned short t_hi[64][66], t_lo[64][66];
    const long long r0 = (long long)blockIdx.y * 64;
    const int c0 = blockIdx.x * 64;
    const int colsp = (cols + 31) & ~31;
    // read: thread -> (row r0 + tid / 4, 16-column group tid % 4): two 16-byte loads per plane
    {
        const int r = threadIdx.x >> 2, cg = (threadIdx.x & 3) * 16;
        const long long row = r0 + r;
        const int c = c0 + cg;
        unsigned short hi[16], lo[16];
        if (row < rows && c < colsp) {
            const char* p = src + row * pitch_src + (long long)(c >> 5) * 128 + ((c & 31) << 1);
            *reinterpret_cast<uint4*>(hi) = *reinterpret_cast<const uint4*>(p);
            *reinterpret_cast<uint4*>(hi + 8) = *reinterpret_cast<const uint4*>(p + 16);
            *reinterpret_cast<uint4*>(lo) = *reinterpret_cast<const uint4*>(p + 64);
            *reinterpret_cast<uint4*>(lo + 8) = *reinterpret_cast<const uint4*>(p + 80);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) hi[i] = lo[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { t_hi[cg + i][r] = hi[i]; t_lo[cg + i][r] = lo[i]; }
    }
    __syncthreads();
    // write: thread -> (destination row c0 + tid / 4, 16 source rows (tid % 4) * 16 ..)
    {
        const int cr = threadIdx.x >> 2, rg = (threadIdx.x & 3) * 16;
        const int c = c0 + cr;
        const long long rr = r0 + rg;
        const long long rowsp = (rows + 31) & ~31LL;
        if (c < cols && rr < rowsp) {
            unsigned short hi[16], lo[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { hi[i] = t_hi[cr][rg + i]; lo[i] = t_lo[cr][rg + i]; }
            char* p = dst + (long long)c * pitch_dst + (rr >> 5) * 128 + ((rr & 31) << 1);
            *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(hi);
            *reinterpret_cast<uint4*>(p + 16) = *reinterpret_cast<const uint4*>(hi + 8);
            *reinterpret_cast<uint4*>(p + 64) = *reinterpret_cast<const uint4*>(lo);
            *reinterpret_cast<uint4*>(p + 80) = *reinterpret_cast<const uint4*>(lo + 8);
        }
    }
}

__global__ void __launch_bounds__(256) split_unpack_kernel(const char* __restrict__ src, long long pitch, long long rows, int cols,
                                                           float inv_scale, float* __restrict__ dst, long long ld) {
    const long long total = rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / cols;
        const int c = (int)(i - row * cols);
        const _Float16* p = reinterpret_cast<const _Float16*>(src + row * pitch + (long long)(c >> 5) * 128) + (c & 31);
        dst[row * ld + c] = ((float)p[0] + (float)p[32]) * inv_scale;
    }
}

inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// ---- tile configurations ---------------------------------------------------------------------------------------------
struct TileCfg {
    int bn, bm, ns, nw;
    void (*kernel)(SplitGemmArgs);
};
#define CLAVA_CFG(BN, BM, WN, NS) {BN, BM, NS, 8, gemm_f16x3_kernel<BN, BM, WN, NS>}
#define CLAVA_CFG4(BN, BM, WN, NS, OCC) {BN, BM, NS, 4, gemm_f16x3_kernel<BN, BM, WN, NS, 4, OCC>}
const TileCfg kCfgs[] = {
    CLAVA_CFG(256, 320, 2, 2),   // 0: 144 KB LDS, 40 MFMA tiles per wave
    CLAVA_CFG(128, 96, 4, 2),    // 1: 128-wide layers at M = 20000 (209 tiles)
    CLAVA_CFG(256, 160, 4, 2),   // 2 (three stages fit in LDS but measured 15 % slower)
    CLAVA_CFG(256, 80, 8, 2),    // 3
    CLAVA_CFG(128, 320, 2, 2),   // 4
    CLAVA_CFG(128, 160, 4, 2),   // 5
    CLAVA_CFG(128, 80, 8, 2),    // 6
    CLAVA_CFG(64, 128, 2, 2),    // 7: narrow outputs (heads, logits, the input gradient)
    CLAVA_CFG(128, 80, 8, 4),    // 8, 9: four-stage twins of 6 and 7 for short contractions in a single round of tiles - the
    CLAVA_CFG(64, 128, 2, 4),    //       whole k range is in flight at once (one exposed load latency instead of one per k block)
    CLAVA_CFG4(128, 160, 2, 2, 2),  // 10: four waves of 64 x 80, two workgroups per CU (their barriers are not in step)
    CLAVA_CFG4(128, 320, 2, 2, 1),  // 11: four waves of 64 x 160
    CLAVA_CFG4(256, 160, 2, 2, 1),  // 12: four waves of 128 x 80
    CLAVA_CFG(128, 128, 2, 2),   // 13, 14: square tiles for M = 4096 (training): 1024-wide layers in one round of 256 tiles
    CLAVA_CFG(128, 128, 2, 4),
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
static_assert(kNumCfgs == PROF_GEMM_F16X3_CFGS, "one profiling family per tile configuration (common.h, capi_core.hip)");
bool g_attr_set[64][kNumCfgs] = {};   // per device: the LDS limit is a per-device function attribute

// Cost model, calibrated on MI355X with tools/bench_f16x3.py: a configuration's sustained rate on a full chip (TFLOP/s
// algorithmic, K = 1024, M = 20000), the tiles of the most loaded CU run back to back, plus a fixed launch / prologue /
// epilogue cost.  It reproduces the measured times of the sampling (M = 20000) and training (M = 4096) shapes to ~15 %.
const double kRateTF[] = {384.0, 300.0, 370.0, 270.0, 362.0, 372.0, 285.0, 320.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // 0: never picked by cost
static_assert(sizeof(kRateTF) / sizeof(kRateTF[0]) == sizeof(kCfgs) / sizeof(kCfgs[0]), "one rate per configuration");

// CLAVA_F16X3_DISABLE="0,8,9": tile configurations the cost model must not pick (experiments)
unsigned disabled_cfgs() {
    static const unsigned mask = [] {
        unsigned m = 0;
        const char* v = getenv("CLAVA_F16X3_DISABLE");
        while (v && *v) {
            char* e = nullptr;
            const long c = strtol(v, &e, 10);
            if (e == v) break;
            if (c >= 0 && c < 32) m |= 1u << c;
            v = (*e == ',') ? e + 1 : e;
        }
        return m;
    }();
    return mask;
}

// configurations whose lanes hold 4 (not 8) consecutive output columns: they cannot write / read derivative bits
constexpr unsigned kOddTnCfgs = (1u << 6) | (1u << 8);

int pick_cfg(int M, int N, int Kp, bool need_bits = false) {
    int best = kNumCfgs - 1;
    double best_cost = 1e300;
    const unsigned off = disabled_cfgs() | (need_bits ? kOddTnCfgs : 0u);
    for (int c = 0; c < kNumCfgs; ++c) {
        const int bn = kCfgs[c].bn, bm = kCfgs[c].bm;
        if (kRateTF[c] <= 0.0 || (off >> c & 1)) continue;
        if (bn > 64 && N <= bn / 2) continue;
        const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        const double per_cu = (double)((tiles + 255) / 256);
        const double tile_us = 2.0 * bn * bm * (double)Kp / (kRateTF[c] * 1e6 / 256.0);
        const double cost = per_cu * tile_us + 5.0;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    // one-round corner (training batches, narrow outputs): when a multi-stage configuration covers the whole product with
    // at most one workgroup per CU, all of it is resident at once and the deeper pipeline wins (measured at M = 4096,
    // tools/bench_f16x3.py: 1024 x 1024 in 26.9 us on 128 x 128 four-stage tiles against 30.3 us; N = 512 in 16.4 us on
    // 64 x 128 four-stage tiles against 23.2 us).  Largest tile first; a configuration must fill at least 5/8 of the chip,
    // except the smallest one, which is the fallback of this corner whenever it fits in one round.
    const int one_round[] = {14, 9};
    for (int c : one_round) {
        if (off >> c & 1) continue;
        const long long tiles = (long long)((M + kCfgs[c].bm - 1) / kCfgs[c].bm) * ((N + kCfgs[c].bn - 1) / kCfgs[c].bn);
        if (tiles <= 256 && (tiles >= 160 || c == 9) && !(kCfgs[c].bn > 64 && N <= kCfgs[c].bn / 2)) return c;
    }
    // 129-256 outputs over a long contraction at sampling batches: 128 x 96 tiles put two workgroups on every CU in ONE round
    // (57 KB of LDS each), which the serial-tiles model above prices as two rounds (measured at M = 20000, N = 256, K = 512,
    // tools/bench_f16x3.py: 21.2 us against 23.8 us on the 128 x 160 tile it picks)
    if (!(off >> 1 & 1) && N > 128 && N <= 256 && Kp >= 512) {
        const long long tiles = (long long)((M + kCfgs[1].bm - 1) / kCfgs[1].bm) * ((N + kCfgs[1].bn - 1) / kCfgs[1].bn);
        if (tiles > 256 && tiles <= 512) best = 1;
    }
    if (best == 6) {
        const long long tiles = (long long)((M + kCfgs[6].bm - 1) / kCfgs[6].bm) * ((N + kCfgs[6].bn - 1) / kCfgs[6].bn);
        if (tiles <= 256 && Kp <= 256 && !(off >> 8 & 1)) best = 8;
    }
    return best;
}

}  // namespace

int clava_gemm_f16x3_config_count() { return kNumCfgs; }

int clava_launch_gemm_f16x3(SplitGemmArgs g, int cfg, hipStream_t stream) {
    CLAVA_CHECK_ARG(g.A && g.W && (g.out_split || g.out_f32), "gemm_f16x3: null operand");
    CLAVA_CHECK_ARG(g.M > 0 && g.N > 0 && g.Kp > 0 && g.Kp % 32 == 0, "gemm_f16x3: bad shape M=%d N=%d Kp=%d (Kp must be a multiple of 32)",
                    g.M, g.N, g.Kp);
    CLAVA_CHECK_ARG(g.pitchA >= 4LL * g.Kp && g.pitchW >= 4LL * g.Kp && g.pitchA % 16 == 0 && g.pitchW % 16 == 0 &&
                        aligned(g.A, 16) && aligned(g.W, 16),
                    "gemm_f16x3: operand rows must be 16-byte aligned and at least 4*Kp bytes apart");
    const int Np = (g.N + 31) & ~31;
    CLAVA_CHECK_ARG(!g.out_split || (g.pitchOut >= 4LL * Np && g.pitchOut % 8 == 0 && aligned(g.out_split, 8)),
                    "gemm_f16x3: output rows must be 8-byte aligned and at least 4*round_up(N,32) bytes apart");
    CLAVA_CHECK_ARG(!g.mask || (g.pitchMask >= 4LL * Np && g.pitchMask % 8 == 0 && aligned(g.mask, 8)), "gemm_f16x3: bad mask pitch");
    CLAVA_CHECK_ARG(!g.out_f32 || g.ldc >= g.N, "gemm_f16x3: ldc too small");
    CLAVA_CHECK_ARG(g.act == CLAVA_ACT_NONE || g.act == CLAVA_ACT_RELU || g.act == CLAVA_ACT_LEAKY, "gemm_f16x3: activation %d", g.act);
    CLAVA_CHECK_ARG(cfg < kNumCfgs, "gemm_f16x3: tile configuration %d does not exist", cfg);
    const int slabs = g.slabs > 1 ? g.slabs : 1;
    if (slabs > 1) {
        CLAVA_CHECK_ARG(g.out_f32 && !g.out_split && !g.mask && !g.bias && !g.bias2 && g.act == CLAVA_ACT_NONE,
                        "gemm_f16x3: a slab product writes plain float32 partial sums only");
        CLAVA_CHECK_ARG(g.kb_per_slab > 0 && (long long)(slabs - 1) * g.kb_per_slab < (g.Kp >> 5) && (long long)slabs * g.kb_per_slab >= (g.Kp >> 5),
                        "gemm_f16x3: %d slabs of %d k blocks do not cover Kp=%d", slabs, g.kb_per_slab, g.Kp);
        CLAVA_CHECK_ARG(g.slab_stride >= (long long)(g.M - 1) * g.ldc + g.N, "gemm_f16x3: slab stride too small");
    }
    // at most 16 outputs, float32 result (a head, the last input-gradient product): memory bound, no tile to fill
    static const bool narrow_on = !(getenv("CLAVA_NARROW_SPLIT") && getenv("CLAVA_NARROW_SPLIT")[0] == '0');
    if (cfg < 0 && narrow_on && clava_narrow_split_ok(g)) return clava_launch_narrow_split(g, stream);
    const bool bits = g.bits_out || g.mask_bits;
    if (bits) {
        CLAVA_CHECK_ARG(g.pitchBits >= Np / 8 && !(g.mask_bits && g.mask) && (!g.bits_out || g.out_split),
                        "gemm_f16x3: derivative bits need rows of >= round_up(N,32)/8 bytes, a split output, and replace the plane mask");
        CLAVA_CHECK_ARG(cfg < 0 || !((kOddTnCfgs >> cfg) & 1), "gemm_f16x3: tile configuration %d cannot handle derivative bits", cfg);
    }
    if (cfg < 0) cfg = pick_cfg(g.M * slabs, g.N, slabs > 1 ? g.kb_per_slab * 32 : g.Kp, bits);
    const TileCfg& tc = kCfgs[cfg];
    const int lds = tc.ns * (tc.bn + tc.bm) * 128;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!g_attr_set[dev][cfg]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tc.kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            clava_set_error("gemm_f16x3: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
            return CLAVA_ERR_LAUNCH;
        }
        g_attr_set[dev][cfg] = true;
    }
    g.tiles_m = (g.M + tc.bm - 1) / tc.bm;
    g.tiles_n = (g.N + tc.bn - 1) / tc.bn;
    const double flops = 2.0 * g.M * g.N * (double)(g.K > 0 ? g.K : g.Kp);
    const double bytes = 4.0 * ((double)g.M * g.Kp + (double)g.N * g.Kp + (double)g.M * g.N);
    ProfScope prof(PROF_GEMM_F16X3 + cfg, flops, bytes, stream);
    hipLaunchKernelGGL(tc.kernel, dim3(g.tiles_m * g.tiles_n, slabs), dim3(64 * tc.nw), lds, stream, g);
    CLAVA_CHECK_LAUNCH("clava_gemm_f16x3");
    return CLAVA_OK;
}

int clava_launch_split_pack(const float* src, long long ld, long long rows, int cols, int shift, char* dst,
                            long long pitch, int* overflow, hipStream_t stream, bool transposed) {
    CLAVA_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld >= (transposed ? rows : (long long)cols), "split_pack: bad arguments");
    const int Kp = (cols + 31) & ~31;
    CLAVA_CHECK_ARG(pitch >= 4LL * Kp && pitch % 8 == 0 && aligned(dst, 8), "split_pack: pitch %lld too small or misaligned", pitch);
    CLAVA_CHECK_ARG(shift >= -24 && shift <= 24, "split_pack: shift %d out of range", shift);
    const int groups = Kp / 4;
    const long long total = rows * groups;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    ProfScope prof(PROF_MISC, 0.0, 8.0 * rows * Kp, stream);
    hipLaunchKernelGGL(split_pack_kernel, dim3(blocks), dim3(256), 0, stream, src, transposed ? 1LL : ld, transposed ? ld : 1LL, rows, cols,
                       groups, ldexpf(1.f, shift), dst, pitch, overflow);
    CLAVA_CHECK_LAUNCH("clava_split_pack");
    return CLAVA_OK;
}

int clava_launch_split_transpose(const char* src, long long pitch_src, long long rows, int cols, char* dst, long long pitch_dst,
                                 hipStream_t stream) {
    CLAVA_CHECK_ARG(src && dst && rows > 0 && cols > 0, "split_transpose: bad arguments");
    const int colsp = (cols + 31) & ~31;
    const long long rowsp = (rows + 31) & ~31LL;
    CLAVA_CHECK_ARG(pitch_src >= 4LL * colsp && pitch_src % 16 == 0 && aligned(src, 16), "split_transpose: source rows must be 16-byte aligned, >= 4*round_up(cols,32) bytes apart");
    CLAVA_CHECK_ARG(pitch_dst >= 4LL * rowsp && pitch_dst % 16 == 0 && aligned(dst, 16), "split_transpose: destination rows must be 16-byte aligned, >= 4*round_up(rows,32) bytes apart");
    CLAVA_CHECK_ARG((rowsp + 63) / 64 <= 65535, "split_transpose: too many rows");
    ProfScope prof(PROF_MISC, 0.0, 8.0 * rows * colsp, stream);
    hipLaunchKernelGGL(split_transpose_kernel, dim3((colsp + 63) / 64, (unsigned)((rowsp + 63) / 64)), dim3(256), 0, stream, src, pitch_src,
                       rows, cols, dst, pitch_dst);
    CLAVA_CHECK_LAUNCH("clava_split_transpose");
    return CLAVA_OK;
}

int clava_launch_split_unpack(const char* src, long long pitch, long long rows, int cols, int shift, float* dst,
                              long long ld, hipStream_t stream) {
    CLAVA_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld >= cols, "split_unpack: bad arguments");
    const int Kp = (cols + 31) & ~31;
    CLAVA_CHECK_ARG(pitch >= 4LL * Kp, "split_unpack: pitch too small");
    const long long total = rows * cols;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(split_unpack_kernel, dim3(blocks), dim3(256), 0, stream, src, pitch, rows, cols, ldexpf(1.f, -shift), dst, ld);
    CLAVA_CHECK_LAUNCH("clava_split_unpack");
    return CLAVA_OK;
}
