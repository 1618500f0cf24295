// Weight-gradient product of the split-f16 training path, straight from batch-major operands (no transposed copies).
//
//   dW[n_out][k_in] = sum_b dy[b][n_out] * x[b][k_in]          (scripts/train.py:37, the weight half of loss.backward())
//
// dy and x are the split rows the forward / input-gradient kernels already wrote: row = batch element, the contraction
// index.  v_mfma_f32_16x16x32_f16 wants, per lane, 8 consecutive contraction elements of one operand row (here: one
// COLUMN of dy or x).  gfx950's ds_read_b64_tr_b16 delivers exactly that from a row-major LDS tile (a 4-row x 16-column
// block per 16 lanes, column-major into the lanes), so the kernel stages [32 batch rows] x [128 columns] tiles of both
// planes of both operands with LDS-DMA as they lie in memory and reads them transposed.  It replaces two
// split_transpose launches per layer (15 % of the training step's GPU time).
//
//   workgroup 512 threads = 8 waves (2 along k_in x 4 along n_out), tile BM (n_out) x BN (k_in) = 256 x 128 (wave tile
//   64 x 64) or 128 x 128; per stage one 8 KB plane image per 128 columns, plane and operand (dy hi, dy lo, x hi, x lo),
//   each [32 rows][256 B] with the 16-byte chunk index XOR
//   ((row & 3) << 2 | (row >> 2) & 3): conflict-free for the transposed reads (a 32-lane half reads two 4-row blocks 8
//   rows apart in the same 16 columns; cdna_hip_programming.md T10 image (b)).  The XOR is applied to the SOURCE
//   address of the LDS-DMA (its LDS side is lane-linear).  Three stages, counted vmcnt waits, raw barriers.
//   blockIdx.y = slab of the batch axis (split-K): partial sums go to out_f32 + slab * slab_stride, summed by the
//   optimizer kernel like the float32 weight-gradient kernels' slabs.
#include "split.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

// 8 consecutive contraction elements (batch rows 8 * (lane >> 4) ..) of image column (16 * tile + (lane & 15)): two
// transposed reads of 4 rows each
__device__ __forceinline__ h8 tr_frag(const char* img, int a0, int a1) {
    union { s4v s[2]; h8 h; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(img + a0));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(img + a1));
    return u.h;
}

constexpr int IMG = 32 * 256;        // one plane image: 32 batch rows x 128 columns x 2 bytes
constexpr int NS = 3;

template <int BM, int BN>
__global__ void __launch_bounds__(512) gemm_f16x3_tn_kernel(SplitGemmArgs g) {
    constexpr int IA = BM / 128, IW = BN / 128;               // 128-column images per plane of each operand
    constexpr int NI = 2 * (IA + IW);                         // images per stage = LDS-DMA instructions per wave per stage
    constexpr int STAGE = NI * IMG;
    constexpr int TM = BM / 64, TN = BN / 32;                 // 16 x 16 MFMA tiles per wave (4 waves along n_out, 2 along k_in)
    static_assert(BM % 128 == 0 && BN % 128 == 0 && NS * STAGE <= 160 * 1024, "tile shape");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 2, wm = wave & 3;
    // XCD-aware order: workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with its own L2.  The
    // (slab, tile) pairs in slab-major order are cut into 8 contiguous runs, one per XCD: with 8 slabs an XCD owns exactly
    // one batch range, whose rows of dy and x its 32 workgroups read in step (each byte enters that L2 once; with the
    // plain order every XCD read every slab and the kernel ran at the Infinity Cache's 4.5 TB/s).  Bijective for any count.
    const int nt = g.tiles_m * g.tiles_n;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int total = nt * (g.slabs > 1 ? g.slabs : 1);
    const int xcd = lin & 7, q8 = total >> 3, rem = total & 7;
    const int virt = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (lin >> 3);
    const int slab = virt / nt, tile = virt - slab * nt;
    const int m0 = (tile / g.tiles_n) * BM;                   // first n_out of the tile (rows of dW)
    const int n0 = (tile % g.tiles_n) * BN;                   // first k_in (columns of dW)
    int nkb = g.Kp >> 5;
    long long row0 = 0;                                       // first batch row of this slab
    if (g.slabs > 1) {
        const int kb0 = slab * g.kb_per_slab;
        nkb = min(nkb - kb0, g.kb_per_slab);
        row0 = (long long)kb0 * 32;
        g.out_f32 += (long long)slab * g.slab_stride;
        if (g.out_colsum) g.out_colsum += (long long)slab * g.slab_stride;
    }
    // the bias gradient (column sums of dy over the batch) rides along in the workgroups of the first k_in tile: one more
    // MFMA per dy fragment and plane against a fragment of ones
    const bool colsum = g.out_colsum != nullptr && n0 == 0 && wn == 0;

    // ---- staging: per stage a wave issues one DMA per plane image (order: dy hi images, dy lo, x hi, x lo), each filling
    // image rows 4*wave .. 4*wave+3 (lane i -> row 4*wave + (i >> 4), chunk slot i & 15); the slot holds source chunk
    // (slot ^ swizzle(row))
    const char* src[NI];
    {
        const int row = 4 * wave + (lane >> 4);
        const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        const int Mp = (g.M + 31) & ~31, Np = (g.N + 31) & ~31;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const bool isA = q < 2 * IA;
            const int qq = isA ? q : q - 2 * IA;
            const int plane = qq / (isA ? IA : IW), im = qq % (isA ? IA : IW);
            // columns beyond the operand's padded width re-read its last chunk: those outputs are never stored
            const int col = min((isA ? m0 : n0) + 128 * im + 8 * ch, (isA ? Mp : Np) - 8);
            src[q] = (isA ? g.A : g.W) + (row0 + row) * (isA ? g.pitchA : g.pitchW) + (long long)(col >> 5) * 128 + ((col & 31) << 1) + plane * 64;
        }
    }
    const long long stepA = 32 * g.pitchA, stepW = 32 * g.pitchW;
    char* const st = smem + wave * 1024;

    // ---- transposed fragment reads: lane (group gq = lane >> 4, i = lane & 15 = 4*q + p) supplies the address of block row q,
    // columns 4p .. 4p+3, and receives column i of the block's four rows
    int ad[TM][2], aw[TN][2];
    {
        const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 8 * gq + 4 * h + q;
            const int sw = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int t = wm * TM + j;                             // 16-column tile of the dy operand
                const int cb = 2 * (t & 7) + (p >> 1);                 // 16-byte chunk of columns 16 * t + 4p inside its image
                ad[j][h] = (t >> 3) * IMG + 256 * r + 16 * (cb ^ sw) + 8 * (p & 1);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int t = wn * TN + i;
                const int cb = 2 * (t & 7) + (p >> 1);
                aw[i][h] = (2 * IA + (t >> 3)) * IMG + 256 * r + 16 * (cb ^ sw) + 8 * (p & 1);
            }
        }
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 accb[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    h8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;

    auto issue = [&](int kb, int stage) {
        const int k = min(kb, nkb - 1);                      // the tail refetches the last block into a stage nobody reads
#pragma unroll
        for (int q = 0; q < NI; ++q) glds16(src[q] + k * (q < 2 * IA ? stepA : stepW), st + stage * STAGE + q * IMG);
    };
    // the same DMAs one group at a time (images q = part, part + TN, ...), placed between the MFMA groups of the main loop:
    // an LDS-DMA holds the wave's issue port for ~100 cycles, and all NI of them in a row at the top of an iteration -- in
    // both waves of a SIMD at once, right after the barrier -- left the matrix pipe idle for as long as a k block's MFMAs
    // take (measured: 1.84 us per k block with every CU busy against 0.9 us of MFMA work)
    auto issue_part = [&](int kb, int stage, int part) {
        const int k = min(kb, nkb - 1);
#pragma unroll
        for (int q = part; q < NI; q += TN) glds16(src[q] + k * (q < 2 * IA ? stepA : stepW), st + stage * STAGE + q * IMG);
    };
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0, s0);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NI * (NS - 2)) : "memory");

    // Main loop, software pipelined over the fragment reads: the operand fragments of MFMA group i+1 are read from LDS
    // while group i multiplies, and the dy fragments + first x fragment of the NEXT k block while the last group does.
    // (With every group's reads in front of its own MFMAs, and one barrier per k block after the last group, a k block
    // took ~2550 cycles for 1536 cycles of MFMA work per SIMD: each group exposed an LDS round trip.)  The barrier
    // therefore sits before the last group: by then this wave has read everything it needs from stage kb (lgkmcnt(0)),
    // so after the barrier stage kb's buffer may be refilled (the DMAs of the next iteration) and stage kb+1, whose DMAs
    // every wave has waited for, may be read.
    constexpr int LAST_PART = (NI > TN - 1) ? (NI - (TN - 1) + TN - 1) / TN : 0;      // DMAs issued with the last group
    auto d_frags = [&](const char* img, h8 (&dh)[TM], h8 (&dl)[TM]) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            dh[j] = tr_frag(img, ad[j][0], ad[j][1]);
            dl[j] = tr_frag(img + IA * IMG, ad[j][0], ad[j][1]);
        }
    };
    int cur = 0, nxt = NS - 1;
    h8 dh[TM], dl[TM];
    d_frags(smem, dh, dl);
    h8 wh = tr_frag(smem, aw[0][0], aw[0][1]), wl = tr_frag(smem + IW * IMG, aw[0][0], aw[0][1]);
    for (int kb = 0; kb < nkb; ++kb) {
        const char* img = smem + cur * STAGE;
        const char* imgn = smem + ((cur + 1 == NS) ? 0 : cur + 1) * STAGE;
        h8 dhn[TM], dln[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            h8 whn, wln;
            if (i + 1 < TN) {
                whn = tr_frag(img, aw[i + 1][0], aw[i + 1][1]);
                wln = tr_frag(img + IW * IMG, aw[i + 1][0], aw[i + 1][1]);
            } else {                                          // past the barrier: the next k block's first fragments
                d_frags(imgn, dhn, dln);
                whn = tr_frag(imgn, aw[0][0], aw[0][1]);
                wln = tr_frag(imgn + IW * IMG, aw[0][0], aw[0][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0 && colsum) {                           // wave-uniform
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, dl[j], accb[j], 0, 0, 0);
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, dh[j], accb[j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, dh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, dl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, dh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue_part(kb + NS - 1, nxt, i);
            if (i == TN - 2)        // stage kb+1 has landed: only younger stages and this iteration's DMAs may be in flight
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NI * (NS - 3) + NI - LAST_PART) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            wh = whn; wl = wln;
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) { dh[j] = dhn[j]; dl[j] = dln[j]; }
        nxt = cur;
        cur = (cur + 1 == NS) ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: D = (x columns as MFMA rows) x (dy columns as MFMA columns): a lane's four accumulator registers are
    // four consecutive k_in of one n_out row: 16-byte float32 stores, four lanes fill 64 contiguous bytes
    const int lrow = lane & 15, kq = lane >> 4;
    const float sc = g.scale_acc * g.alpha;
    const bool vec = (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.out_f32) & 15) == 0;
    if (colsum && kq == 0) {                                 // every row of the ones product holds the sums: take row 0
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + 16 * (wm * TM + j) + lrow;
            if (m < g.M) g.out_colsum[m] = accb[j][0] * g.scale_colsum;
        }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + 16 * (wm * TM + j) + lrow;
        if (m >= g.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + 16 * (wn * TN + i) + 4 * kq;
            if (n >= g.N) continue;
            float* cp = g.out_f32 + (long long)m * g.ldc + n;
            const f32x4 v = acc[i][j] * sc;
            if (vec && n + 4 <= g.N) *reinterpret_cast<f32x4*>(cp) = v;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < g.N) cp[r] = v[r];
            }
        }
    }
}

bool g_tn_attr_set[64][2] = {};

}  // namespace

int clava_launch_gemm_f16x3_tn(SplitGemmArgs g, hipStream_t stream) {
    CLAVA_CHECK_ARG(g.A && g.W && g.out_f32 && !g.out_split && !g.mask && !g.bias && !g.bias2 && g.act == CLAVA_ACT_NONE,
                    "gemm_f16x3_tn: the batch-major product writes plain float32 sums only");
    CLAVA_CHECK_ARG(g.M > 0 && g.N > 0 && g.Kp > 0 && g.Kp % 32 == 0 && g.K == g.Kp,
                    "gemm_f16x3_tn: bad shape n_out=%d k_in=%d batch=%d (the batch must be a multiple of 32: pad rows would enter the sums)",
                    g.M, g.N, g.K);
    const long long Mp = (g.M + 31) & ~31, Np = (g.N + 31) & ~31;
    CLAVA_CHECK_ARG(g.pitchA >= 4 * Mp && g.pitchW >= 4 * Np && g.pitchA % 16 == 0 && g.pitchW % 16 == 0 &&
                        (reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.W) & 15) == 0,
                    "gemm_f16x3_tn: operand rows must be 16-byte aligned and at least 4*round_up(cols,32) bytes apart");
    CLAVA_CHECK_ARG(g.ldc >= g.N, "gemm_f16x3_tn: ldc too small");
    const int slabs = g.slabs > 1 ? g.slabs : 1;
    if (slabs > 1) {
        CLAVA_CHECK_ARG(g.kb_per_slab > 0 && (long long)(slabs - 1) * g.kb_per_slab < (g.Kp >> 5) && (long long)slabs * g.kb_per_slab >= (g.Kp >> 5),
                        "gemm_f16x3_tn: %d slabs of %d k blocks do not cover batch %d", slabs, g.kb_per_slab, g.Kp);
        CLAVA_CHECK_ARG(g.slab_stride >= (long long)(g.M - 1) * g.ldc + g.N, "gemm_f16x3_tn: slab stride too small");
    }
    // 256 x 128 tiles (wave tile 64 x 64: a third fewer LDS bytes per MFMA) when they still give every CU a workgroup
    const long long big_tiles = (long long)((g.M + 255) / 256) * ((g.N + 127) / 128) * slabs;
    const int which = (g.M > 128 && big_tiles >= 200) ? 1 : 0;
    const int bm = which ? 256 : 128, bn = 128;
    void (*kernel)(SplitGemmArgs) = which ? gemm_f16x3_tn_kernel<256, 128> : gemm_f16x3_tn_kernel<128, 128>;
    const int lds = NS * 2 * ((bm + bn) / 128) * IMG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!g_tn_attr_set[dev][which]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            clava_set_error("gemm_f16x3_tn: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
            return CLAVA_ERR_LAUNCH;
        }
        g_tn_attr_set[dev][which] = true;
    }
    g.tiles_m = (g.M + bm - 1) / bm;
    g.tiles_n = (g.N + bn - 1) / bn;
    const double flops = 2.0 * g.M * g.N * (double)g.K;
    const double bytes = 4.0 * ((double)g.M * g.Kp + (double)g.N * g.Kp + (double)g.M * g.N * slabs);
    ProfScope prof(PROF_GEMM_F16X3_TN, flops, bytes, stream);
    hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n, slabs), dim3(512), lds, stream, g);
    CLAVA_CHECK_LAUNCH("clava_gemm_f16x3_tn");
    return CLAVA_OK;
}
