// Fused "narrow" stacks of the guided reverse step: the skinny ends of the denoiser and of the guidance classifier,
// each as ONE kernel in which a wavefront carries 32 rows through several Linear layers entirely in registers.
//
// Replaces, per reverse step of `conditional_sample` (tab_ddpm/gaussian_multinomial_diffsuion.py:995-1014) at the
// configs' shapes (configs/*.json: classifier d_layers [128,256,512,1024,512,256,128], dim_t 128, <= 64 clusters,
// <= 32 numeric columns; denoiser [512,1024x4,512]):
//   den_front   proj(x) + emb -> Linear(128,512)+ReLU                          (tab_ddpm/modules.py:451-460; 2 launches)
//   cls_front   proj(x) + emb -> Linear(128,128)+ReLU+BN -> Linear(128,256)+LeakyReLU      (pipeline_utils.py:625-630; 3)
//   cls_tail    Linear(256,128)+LeakyReLU+BN -> Linear(128,C) -> onehot(y) - softmax -> the two input-gradient products back
//               to 256 columns                                                 (pipeline_utils.py:861-864; 5 launches)
//   cls_back    the last three input-gradient products 256 -> 128 -> 128 -> n_num, times classifier_scale   (:864; 3)
//   den_head    Linear(512, n_num) + the whole fused Gaussian step             (modules.py:460 + :200-237, 315-328, 354-383; 2)
// 15 launches with an HBM round trip each become 5, and the [B,128] / [B,C] / [B,n_num] intermediates never leave
// registers.
//
// Mechanism (split-f16 arithmetic of split.h, float32-level accuracy): v_mfma_f32_32x32x16_f16 with the WEIGHTS as the
// first operand (rows = output features) and the activations as the second (columns = batch rows), so a result tile has
// its batch row on the lane and its 32 output features in the 16 accumulator registers of the two lane halves: exactly
// the operand layout of the next layer's contraction, which therefore takes it with no lane movement and no LDS (the
// k order inside a 16-deep step is permuted - element j of lane half h is feature 16 s + 8 (j >> 2) + 4 h + (j & 3) - and
// the host packs every weight matrix in that order: clavaddpm_amd/split_nets.py::pack_fused).
// Weights: a kernel consumes its layers' fragments in one fixed order, so the host lays them out as ONE linear stream per
// kernel; a workgroup of four wavefronts (128 rows) pulls that stream through a three-buffer LDS ring of 32 KB chunks
// with global_load_lds (no registers, two chunks in flight while one is multiplied; one barrier per chunk, counted vmcnt),
// and every wavefront reads its fragments from the ring with conflict-free ds_read_b128.  (A first version loaded the
// fragments straight from L2 into registers: at one wavefront per SIMD and ~250 live registers the compiler issued each
// load right before its use, and the kernels ran 8x slower than their MFMA time.)
#include "clava_hip.h"
#include "fused_narrow.h"
#include "rng.h"
#include "step.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Developer aid (tools/dbg_fused_stamps.py): cycle-counter stamps of workgroup 0, wavefront 0 of the last cls_tail launch
static __device__ long long g_stamps[16];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

// -DCLAVA_BACK_STAMPS (tools/build_variant.py): the stamps come from cls_back instead
#ifdef CLAVA_BACK_STAMPS
#define BSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#undef STAMP
#define STAMP(i)
#else
#define BSTAMP(i)
#endif
namespace {

constexpr int WAVES_PER_WG = 4;

struct Lane {
    int m, h;            // batch row inside the wavefront's 32-row tile, lane half
    long long row;       // global row (may be >= B in the last tile)
    long long row_ld;    // row clamped for loads
    bool valid;
};

__device__ __forceinline__ Lane lane_of(int B) {
    Lane L;
    const int lane = threadIdx.x & 63;
    L.m = lane & 31; L.h = lane >> 5;
    L.row = ((long long)blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6)) * 32 + L.m;
    L.valid = L.row < B;
    L.row_ld = L.valid ? L.row : (long long)B - 1;
    return L;
}

// Activation fragments of KS 16-deep steps from split rows in the permuted k order: per step and plane two 8-byte pieces
// (features 16 s + 4 h .. + 3 and 16 s + 8 + 4 h .. + 3 of the 32-column block s >> 1).
template <int KS>
__device__ __forceinline__ void load_act(const char* __restrict__ rowp, int h, h8 (&ah)[KS], h8 (&al)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const char* p = rowp + (s >> 1) * 128 + ((s & 1) * 16 + 4 * h) * 2;
        const h4 a0 = *reinterpret_cast<const h4*>(p), a1 = *reinterpret_cast<const h4*>(p + 16);
        const h4 b0 = *reinterpret_cast<const h4*>(p + 64), b1 = *reinterpret_cast<const h4*>(p + 80);
        ah[s] = h8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        al[s] = h8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }
}

// The same fragments for a WIDE activation (NB 128-byte blocks = 2 NB steps), fetched row-contiguously: one load
// instruction covers 8 rows x 128 bytes (8 cache lines; the per-lane fetch above touches 32 lines per instruction for 512
// bytes, and with four wavefronts on one texture unit that was ~9 us of a 40 us kernel), then the wavefront transposes block
// by block through a private 32 x 136-byte LDS tile (LDS executes a wavefront's accesses in order: no waits between a
// block's reads and the next block's writes; the 136-byte row stride keeps the 8-byte reads conflict-free).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int STAGE_ROW = 136, STAGE_WAVE = 32 * STAGE_ROW;

template <int NB>
struct StagedAct {
    u32x4 raw[NB][4];
    // issue the loads (rows of the wavefront's tile beyond B - 1 read row B - 1)
    __device__ __forceinline__ void issue(const char* __restrict__ base, long long pitch, long long row0, int B) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long row = row0 + 8 * i + (lane >> 3);
            row = row < B ? row : (long long)B - 1;
            const char* p = base + row * pitch + (lane & 7) * 16;
#pragma unroll
            for (int b = 0; b < NB; ++b) raw[b][i] = *reinterpret_cast<const u32x4*>(p + b * 128);
        }
    }
    // transpose into the fragment registers of lane (m, h); `tile` = this wavefront's STAGE_WAVE bytes of LDS
    __device__ __forceinline__ void finish(char* tile, h8 (&ah)[2 * NB], h8 (&al)[2 * NB]) {
        const int lane = threadIdx.x & 63;
        char* wr = tile + (lane >> 3) * STAGE_ROW + (lane & 7) * 16;
        const char* rd = tile + (lane & 31) * STAGE_ROW + (lane >> 5) * 8;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32x2*>(wr + 8 * i * STAGE_ROW) = u32x2{raw[b][i][0], raw[b][i][1]};
                *reinterpret_cast<u32x2*>(wr + 8 * i * STAGE_ROW + 8) = u32x2{raw[b][i][2], raw[b][i][3]};
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h4 a0 = *reinterpret_cast<const h4*>(rd + 32 * s2), a1 = *reinterpret_cast<const h4*>(rd + 32 * s2 + 16);
                const h4 b0 = *reinterpret_cast<const h4*>(rd + 64 + 32 * s2), b1 = *reinterpret_cast<const h4*>(rd + 64 + 32 * s2 + 16);
                ah[2 * b + s2] = h8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                al[2 * b + s2] = h8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            }
        }
    }
};

// ---- the weight stream: three 32 KB LDS buffers filled by LDS-DMA, 16 fragment pairs (hi 1 KB | lo 1 KB) per chunk
constexpr int RING_CHUNK = 32768, RING_BUFS = 3, PAIRS_PER_CHUNK = 16, WAVES = WAVES_PER_WG, DMA_PER_WAVE = RING_CHUNK / 1024 / WAVES;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct WRing {
    const char* g;     // the kernel's weight stream (whole chunks, zero padded)
    char* lds;         // ring base
    int n_chunks;
    int wave, lane;
    // one of the DMA_PER_WAVE LDS-DMA instructions with which this wavefront fills its share of a chunk
    __device__ __forceinline__ void issue_part(int chunk, int q) const {
        if (chunk >= n_chunks) return;
        const char* src = g + (long long)chunk * RING_CHUNK + (wave * DMA_PER_WAVE + q) * 1024 + lane * 16;
        char* dst = lds + (chunk % RING_BUFS) * RING_CHUNK + (wave * DMA_PER_WAVE + q) * 1024;      // wave-uniform: the DMA adds lane * 16
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    }
    __device__ __forceinline__ void prologue() const {
#pragma unroll
        for (int q = 0; q < 2 * DMA_PER_WAVE; ++q) issue_part(q / DMA_PER_WAVE, q % DMA_PER_WAVE);
    }
    // before the first fragment of chunk `chunk` is read: this wavefront's part of it has landed (everything but the newest
    // chunk's DMA_PER_WAVE loads - other memory operations issued since only make the wait stricter), every wavefront's has
    // (barrier), and every wavefront is done with chunk - 1, whose buffer the DMAs of chunk + 2 overwrite; those are issued
    // one at a time behind the MFMAs of this chunk (an LDS-DMA costs ~100 issue cycles: eight in a row in front of the MFMAs
    // stalled the matrix pipe for half a chunk).
    // (wait and barrier in one asm statement with a memory clobber: the compiler moves no LDS read or DMA across it)
    __device__ __forceinline__ void enter(int chunk) const {
        if (chunk + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
};

// acc[nt] (nt < NT) += the next KS * NT fragment pairs of the stream (order: k-step outside, n-tile inside, so that
// consecutive MFMAs write different accumulators) x activation fragments.  P0 = index of the first pair in the stream.
template <int KS, int NT, int P0>
__device__ __forceinline__ void wave_linear(const WRing& r, const h8 (&ah)[KS], const h8 (&al)[KS], f32x16 (&acc)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nt][v] = 0.f;
    auto frag = [&](int p) -> const char* {
        return r.lds + ((p / PAIRS_PER_CHUNK) % RING_BUFS) * RING_CHUNK + (p % PAIRS_PER_CHUNK) * 2048 + r.lane * 16;
    };
    // The fragment pair of step i + 1 is read from the ring while the three MFMAs of step i run (a ds_read_b128 takes longer
    // than they do when it is issued right in front of them); only the first pair of a chunk waits for its data, behind the
    // chunk's barrier.  All indices fold to constants after unrolling.
    if ((P0 % PAIRS_PER_CHUNK) == 0) r.enter(P0 / PAIRS_PER_CHUNK);
    h8 wh = *reinterpret_cast<const h8*>(frag(P0)), wl = *reinterpret_cast<const h8*>(frag(P0) + 1024);
#pragma unroll
    for (int i = 0; i < KS * NT; ++i) {
        const int s = i / NT, nt = i % NT, pn = P0 + i + 1;
        const bool more = i + 1 < KS * NT, boundary = (pn % PAIRS_PER_CHUNK) == 0;
        h8 wh_n = wh, wl_n = wl;
        if (more && !boundary) {
            wh_n = *reinterpret_cast<const h8*>(frag(pn));
            wl_n = *reinterpret_cast<const h8*>(frag(pn) + 1024);
        }
        // pin the order: left alone, the scheduler sinks the two reads behind the MFMAs (one register set, every read
        // followed by lgkmcnt(0)) because the kernel sits at the 256-VGPR limit
        __builtin_amdgcn_sched_barrier(0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah[s], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al[s], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah[s], acc[nt], 0, 0, 0);
        // chunk + 2 is fetched during this chunk: one DMA behind every second pair's MFMAs
        if (((P0 + i) % PAIRS_PER_CHUNK) % (PAIRS_PER_CHUNK / DMA_PER_WAVE) == 0)
            r.issue_part((P0 + i) / PAIRS_PER_CHUNK + 2, ((P0 + i) % PAIRS_PER_CHUNK) / (PAIRS_PER_CHUNK / DMA_PER_WAVE));
        __builtin_amdgcn_sched_barrier(0);
        if (more && boundary) {
            r.enter(pn / PAIRS_PER_CHUNK);
            wh_n = *reinterpret_cast<const h8*>(frag(pn));
            wl_n = *reinterpret_cast<const h8*>(frag(pn) + 1024);
        }
        wh = wh_n; wl = wl_n;
    }
}

// feature index of accumulator register v of n-tile nt in lane half h
__device__ __forceinline__ int feat(int nt, int v, int h) { return 32 * nt + 8 * (v >> 2) + 4 * h + (v & 3); }

// Epilogue of a forward layer: y = act(acc * scale_acc + bias) in place (true values).  `bias` points into the LDS copy
// made by stage_bias (32 * n-tiles floats, padded; the broadcast time-embedding row already added for layer 0): a global
// load here would expose an L2 / HBM round trip once per layer.
// The activation is max(x, slope * x) (slope 0 / 0.01 / 1 for ReLU / LeakyReLU / none): straight-line code - with the
// activation tested per element the compiler emitted a scalar branch for each of a lane's 64 values.
template <int NT>
__device__ __forceinline__ void epi_forward(f32x16 (&acc)[NT], int nt0, int h, const float* bias, float scale_acc, int act) {
    const float slope = act == CLAVA_ACT_RELU ? 0.f : (act == CLAVA_ACT_LEAKY ? 0.01f : 1.f);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 32 * (nt0 + nt) + 8 * g + 4 * h;
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = acc[nt][4 * g + r] * scale_acc + b[r];
                acc[nt][4 * g + r] = fmaxf(x, slope * x);
            }
        }
}

// dst[0 .. n) = bias[i] (+ bias2[i]): cooperative copy into LDS; visible after the ring's first barrier (enter(0) waits for
// the wavefront's own LDS writes before it)
__device__ __forceinline__ void stage_bias(float* dst, const float* __restrict__ bias, const float* __restrict__ bias2, int n) {
    for (int i = threadIdx.x; i < n; i += 64 * WAVES_PER_WG) dst[i] = bias[i] + (bias2 ? bias2[i] : 0.f);
}

// true values in acc -> operand fragments of the next contraction (2 * NT steps), planes scaled by scale_out
template <int NT>
__device__ __forceinline__ void to_frags(const f32x16 (&acc)[NT], float scale_out, h8 (&ah)[2 * NT], h8 (&al)[2 * NT], float& ovf) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sc = acc[nt][8 * s2 + j] * scale_out;
                ovf = fmaxf(ovf, fabsf(sc));
                const _Float16 hi = (_Float16)sc;
                ah[2 * nt + s2][j] = hi;
                al[2 * nt + s2][j] = (_Float16)(sc - (float)hi);
            }
}

// true values in acc (n-tiles nt0 ..) -> split rows in global memory (columns >= n_valid written as zeros)
template <int NT>
__device__ __forceinline__ void store_split_rows(const f32x16 (&acc)[NT], int nt0, const Lane& L, char* __restrict__ base,
                                                 long long pitch, float scale_out, int n_valid, float& ovf) {
    if (!L.valid) return;
    char* rowp = base + L.row * pitch;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            h4 hi, lo;
            const int f0 = 32 * (nt0 + nt) + 8 * g + 4 * L.h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = (f0 + r < n_valid) ? acc[nt][4 * g + r] * scale_out : 0.f;
                ovf = fmaxf(ovf, fabsf(sc));
                hi[r] = (_Float16)sc;
                lo[r] = (_Float16)(sc - (float)hi[r]);
            }
            char* p = rowp + (long long)(nt0 + nt) * 128 + (8 * g + 4 * L.h) * 2;
            *reinterpret_cast<h4*>(p) = hi;
            *reinterpret_cast<h4*>(p + 64) = lo;
        }
}

// multiply acc by act'(a) read off the operand-layout hi plane of the activation (steps M0 .. of `mask`: 1 where > 0 else
// slope), then by scale.  The mask array is taken by reference with a compile-time offset: a pointer into a register array
// would send it to scratch memory.
template <int NT, int M0, int MS>
__device__ __forceinline__ void epi_backward(f32x16 (&acc)[NT], const h8 (&mask)[MS], float slope, float scale) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float x = acc[nt][v] * scale;
            acc[nt][v] = x * (((float)mask[M0 + 2 * nt + (v >> 3)][v & 7] > 0.f) ? 1.f : slope);
        }
}
// the same with the derivative as bits: bit 16 nt + v of the lane's words (mask_bits_of below)
template <int NT>
__device__ __forceinline__ void epi_backward_bits(f32x16 (&acc)[NT], const unsigned (&bits)[(NT + 1) / 2], float slope, float scale) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v)
            acc[nt][v] = acc[nt][v] * scale * (((bits[nt >> 1] >> (16 * (nt & 1) + v)) & 1u) ? 1.f : slope);
}
// bit 8 s + j of the result = (fragment element j of step s > 0): the lane's 16 NT features in accumulator order
template <int NT>
__device__ __forceinline__ void mask_bits_of(const h8 (&ah)[2 * NT], unsigned (&bits)[(NT + 1) / 2]) {
#pragma unroll
    for (int w = 0; w < (NT + 1) / 2; ++w) bits[w] = 0u;
#pragma unroll
    for (int s = 0; s < 2 * NT; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) bits[s >> 2] |= (ah[s][j] > (_Float16)0 ? 1u : 0u) << (8 * (s & 3) + j);
}
template <int NT>
__device__ __forceinline__ void epi_scale(f32x16 (&acc)[NT], float scale) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nt][v] *= scale;
}

// ---------------------------------------------------------------------------------------------------- forward stacks
constexpr int RING_LDS = RING_CHUNK * RING_BUFS, BIAS_LDS_FLOATS = 2048;
// the ring, then BIAS_LDS_FLOATS floats of staged biases, then one transposition tile per wavefront
extern __shared__ __attribute__((aligned(1024))) char ring_mem[];
__device__ __forceinline__ float* bias_lds() { return reinterpret_cast<float*>(ring_mem + RING_LDS); }
__device__ __forceinline__ char* stage_tile() { return ring_mem + RING_LDS + BIAS_LDS_FLOATS * 4 + (threadIdx.x >> 6) * STAGE_WAVE; }
__device__ __forceinline__ long long wave_row0() { return ((long long)blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6)) * 32; }

__device__ __forceinline__ WRing make_ring(const char* stream, int n_chunks) {
    WRing r;
    r.g = stream; r.lds = ring_mem; r.n_chunks = n_chunks;
    r.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // provably wave-uniform: no waterfall loop around the DMA's LDS base
    r.lane = threadIdx.x & 63;
    r.prologue();
    return r;
}

// x (split rows, <= 16 * KS0 columns) -> L0 (+ broadcast time-embedding row): the shared front of cls_front, cls_back (which
// recomputes the mask of L1) and den_front.  Fills L1's input fragments (8 steps).  Consumes pairs 0 .. 4 * KS0 - 1.
template <int KS0>
__device__ __forceinline__ void front_l0(const WRing& r, const FusedLayer& l0, const float* b0, const char* __restrict__ x,
                                         long long pitch_x, const Lane& L, h8 (&ah)[8], h8 (&al)[8], float& ovf) {
    h8 xh[KS0], xl[KS0];
    load_act<KS0>(x + L.row_ld * pitch_x, L.h, xh, xl);
    f32x16 acc[4];
    wave_linear<KS0, 4, 0>(r, xh, xl, acc);
    epi_forward<4>(acc, 0, L.h, b0, l0.scale_acc, l0.act);
    to_frags<4>(acc, l0.scale_out, ah, al, ovf);
}

template <int KS0>
__global__ void __launch_bounds__(256) cls_front_kernel(FusedFrontArgs a) {
    const Lane L = lane_of(a.B);
    const WRing r = make_ring(a.stream, a.n_chunks);
    float ovf = 0.f;      // largest plane magnitude written: beyond the float16 range raises the caller's overflow flag
    constexpr int P1 = 4 * KS0, P2 = P1 + 32;
    float* const b0 = bias_lds(); float* const b1 = b0 + 128; float* const b2 = b1 + 128;
    stage_bias(b0, a.l[0].bias, a.l[0].bias2, 128); stage_bias(b1, a.l[1].bias, nullptr, 128); stage_bias(b2, a.l[2].bias, nullptr, 256);
    h8 a0h[8], a0l[8], a1h[8], a1l[8];
    front_l0<KS0>(r, a.l[0], b0, a.x, a.pitch_x, L, a0h, a0l, ovf);
    {
        f32x16 acc[4];
        wave_linear<8, 4, P1>(r, a0h, a0l, acc);
        epi_forward<4>(acc, 0, L.h, b1, a.l[1].scale_acc, a.l[1].act);
        to_frags<4>(acc, a.l[1].scale_out, a1h, a1l, ovf);
        if (a.mask1 && L.valid) {            // what cls_back multiplies its second product by: no recomputation there
            unsigned bits[2];
            mask_bits_of<4>(a1h, bits);
            *reinterpret_cast<u32x2*>(a.mask1 + (L.row * 2 + L.h) * 2) = u32x2{bits[0], bits[1]};
        }
    }
    {                                        // 256 output columns in two halves: 64 accumulator registers at a time
        f32x16 acc[4];
        wave_linear<8, 4, P2>(r, a1h, a1l, acc);
        epi_forward<4>(acc, 0, L.h, b2, a.l[2].scale_acc, a.l[2].act);
        store_split_rows<4>(acc, 0, L, a.out, a.pitch_out, a.l[2].scale_out, 1 << 30, ovf);
    }
    {
        f32x16 acc[4];
        wave_linear<8, 4, P2 + 32>(r, a1h, a1l, acc);
        epi_forward<4>(acc, 4, L.h, b2, a.l[2].scale_acc, a.l[2].act);
        store_split_rows<4>(acc, 4, L, a.out, a.pitch_out, a.l[2].scale_out, 1 << 30, ovf);
    }
    if (!(ovf <= 65504.f) && a.overflow) *a.overflow = 1;
}

// den_front: x -> proj + emb -> Linear(128, 128 * NG) + ReLU, written as split rows
template <int KS0, int NG, int G0 = 0>
struct DenFrontGroups {
    static __device__ __forceinline__ void run(const WRing& r, const FusedFrontArgs& a, const Lane& L, const h8 (&a0h)[8],
                                               const h8 (&a0l)[8], float& ovf) {
        f32x16 acc[4];
        wave_linear<8, 4, 4 * KS0 + 32 * G0>(r, a0h, a0l, acc);
        epi_forward<4>(acc, 4 * G0, L.h, bias_lds() + 128, a.l[1].scale_acc, a.l[1].act);
        store_split_rows<4>(acc, 4 * G0, L, a.out, a.pitch_out, a.l[1].scale_out, 1 << 30, ovf);
        if constexpr (G0 + 1 < NG) DenFrontGroups<KS0, NG, G0 + 1>::run(r, a, L, a0h, a0l, ovf);
    }
};

template <int KS0, int NG>
__global__ void __launch_bounds__(256) den_front_kernel(FusedFrontArgs a) {
    const Lane L = lane_of(a.B);
    const WRing r = make_ring(a.stream, a.n_chunks);
    float ovf = 0.f;      // largest plane magnitude written: beyond the float16 range raises the caller's overflow flag
    stage_bias(bias_lds(), a.l[0].bias, a.l[0].bias2, 128); stage_bias(bias_lds() + 128, a.l[1].bias, nullptr, 128 * NG);
    h8 a0h[8], a0l[8];
    front_l0<KS0>(r, a.l[0], bias_lds(), a.x, a.pitch_x, L, a0h, a0l, ovf);
    DenFrontGroups<KS0, NG>::run(r, a, L, a0h, a0l, ovf);
    if (!(ovf <= 65504.f) && a.overflow) *a.overflow = 1;
}

// ---------------------------------------------------------------------------------------------------- classifier tail
// a6 [B,256] -> L7 (256->128, leaky) -> L8 (128->C) -> dlogits = onehot(y) - softmax(logits) -> d/d(pre-activation of L7)
// (128) -> d/d(pre-activation of L6) (256), written as split rows for the wide input-gradient products that follow.
// Stream: L7 (64 pairs), L8 (16), L8^T (16), L7^T (2 x 32) = 160 pairs = 10 chunks.
__global__ void __launch_bounds__(256) cls_tail_kernel(FusedTailArgs a) {
    STAMP(0);
    const Lane L = lane_of(a.B);
    h8 a6h[16], a7h[8], a7l[8];
    StagedAct<8> in;
    in.issue(a.a6, a.pitch_a6, wave_row0(), a.B);            // the oldest loads in flight: what the first product waits for
    const WRing r = make_ring(a.stream, a.n_chunks);
    float ovf = 0.f;      // largest plane magnitude written: beyond the float16 range raises the caller's overflow flag
    float* const b7 = bias_lds(); float* const b8 = b7 + 128;
    stage_bias(b7, a.l7.bias, nullptr, 128); stage_bias(b8, a.l8.bias, nullptr, 64);
    {
        h8 a6l[16];
        in.finish(stage_tile(), a6h, a6l);
        f32x16 acc[4];
        STAMP(1);
        wave_linear<16, 4, 0>(r, a6h, a6l, acc);
        STAMP(2);
        epi_forward<4>(acc, 0, L.h, b7, a.l7.scale_acc, a.l7.act);
        to_frags<4>(acc, a.l7.scale_out, a7h, a7l, ovf);
        STAMP(3);
    }
    h8 dlh[4], dll[4];
    {
        f32x16 lg[2];
        wave_linear<8, 2, 64>(r, a7h, a7l, lg);
        epi_forward<2>(lg, 0, L.h, b8, a.l8.scale_acc, CLAVA_ACT_NONE);
        // log-softmax over the C classes of the row: this lane holds 32 of the 64 padded columns, lane ^ 32 the others
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                if (feat(nt, v, L.h) >= a.C) lg[nt][v] = -INFINITY;
                mx = fmaxf(mx, lg[nt][v]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // softmax = e / sum(e), e = 2^((logit - max) log2 e) on the hardware exp2 (one pass; the padded classes' -inf gives 0):
        // the rounding of the exponent moves a probability by < 4e-8 absolute
        float sm = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                lg[nt][v] = __builtin_amdgcn_exp2f((lg[nt][v] - mx) * 1.44269504088896340736f);
                sm += lg[nt][v];
            }
        sm += __shfl_xor(sm, 32, 64);
        const float inv = 1.f / sm;
        const int yy = a.y[L.row_ld];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) lg[nt][v] = (feat(nt, v, L.h) == yy ? 1.f : 0.f) - lg[nt][v] * inv;
        to_frags<2>(lg, a.scale_dlast, dlh, dll, ovf);
    }
    STAMP(4);
    h8 g7h[8], g7l[8];
    {
        f32x16 acc[4];
        wave_linear<4, 4, 80>(r, dlh, dll, acc);
        STAMP(5);
        epi_backward<4, 0>(acc, a7h, a.slope7, a.scale_acc_g8);
        to_frags<4>(acc, a.scale_out_g8, g7h, g7l, ovf);
    }
    STAMP(6);
    {
        f32x16 acc[4];
        wave_linear<8, 4, 96>(r, g7h, g7l, acc);
        STAMP(7);
        epi_backward<4, 0>(acc, a6h, a.slope6, a.scale_acc_g7);
        store_split_rows<4>(acc, 0, L, a.out, a.pitch_out, a.scale_out_g7, 1 << 30, ovf);
    }
    STAMP(8);
    {
        f32x16 acc[4];
        wave_linear<8, 4, 128>(r, g7h, g7l, acc);
        STAMP(9);
        epi_backward<4, 8>(acc, a6h, a.slope6, a.scale_acc_g7);
        store_split_rows<4>(acc, 4, L, a.out, a.pitch_out, a.scale_out_g7, 1 << 30, ovf);
    }
    STAMP(10);
    if (!(ovf <= 65504.f) && a.overflow) *a.overflow = 1;
}

// ---------------------------------------------------------------------------------------------------- classifier back
// g3 [B,256] (d/d pre-activation of L2) -> x W2 -> times act'(L1) (the bits cls_front stored) -> x W1 -> x W0 ->
// gx [B, n_num] = classifier_scale * d/dx, float32.  Stream: L2^T (64 pairs), L1^T (32), L0^T (8).
// GAUSS: gx is the guidance gradient of a reverse step whose Gaussian posterior step (step.h: gauss_step_element) follows in
// the same thread, as in narrow_split_kernel<.., true>: one launch and one cross-stream hand-over less per step.
template <bool GAUSS>
__global__ void __launch_bounds__(256) cls_back_kernel(FusedBackArgs a) {
    BSTAMP(0);
    const Lane L = lane_of(a.B);
    // the Gaussian step's inputs that do not depend on the gradient - table entries, state, eps, the noise - for the lane's
    // first four columns (all of them up to 8 numeric columns): their loads are the kernel's FIRST (the ring's counted waits
    // retire memory operations in order: issued later they would hold up a chunk barrier), the Philox draw follows once the
    // gradient rows and the first weight chunks are on their way
    GaussRow gk = {};
    GaussIn gin0[4] = {};
    float ge0[4] = {0.f, 0.f, 0.f, 0.f};
    const bool g_lane = GAUSS && L.valid && 4 * L.h < a.n_num;
    if constexpr (GAUSS) {
        if (g_lane) {
            gk = gauss_row(a.g, (int)L.row);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (4 * L.h + u < a.n_num) {
                    const long long i = (long long)L.row * a.n_num + 4 * L.h + u;
                    ge0[u] = a.g.eps[(size_t)L.row * a.g.ld_eps + 4 * L.h + u];
                    gin0[u].xd = a.g.x[i];
                    if (a.g.noise) gin0[u].z = a.g.noise[i];
                }
        }
    }
    StagedAct<8> in;
    in.issue(a.g3, a.pitch_g3, wave_row0(), a.B);
    const u32x2 mb = *reinterpret_cast<const u32x2*>(a.mask1 + (L.row_ld * 2 + L.h) * 2);
    const WRing r = make_ring(a.stream, a.n_chunks);
    if constexpr (GAUSS) {
        if (g_lane && !a.g.noise) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (4 * L.h + u < a.n_num) gin0[u].z = gauss_draw(a.g, (int)L.row, 4 * L.h + u);
        }
    }
    float ovf = 0.f;      // largest plane magnitude written: beyond the float16 range raises the caller's overflow flag
    constexpr int P3 = 64, P4 = P3 + 32;
    h8 g2h[8], g2l[8];
    {
        h8 g3h[16], g3l[16];
        in.finish(stage_tile(), g3h, g3l);
        f32x16 acc[4];
        BSTAMP(1);
        wave_linear<16, 4, 0>(r, g3h, g3l, acc);
        BSTAMP(2);
        const unsigned bits[2] = {mb[0], mb[1]};
        epi_backward_bits<4>(acc, bits, a.slope1, a.scale_acc_g2);
        to_frags<4>(acc, a.scale_out_g2, g2h, g2l, ovf);
        BSTAMP(3);
    }
    h8 g1h[8], g1l[8];
    {
        f32x16 acc[4];
        wave_linear<8, 4, P3>(r, g2h, g2l, acc);
        BSTAMP(4);
        epi_scale<4>(acc, a.scale_acc_g1);                     // proj has no activation
        to_frags<4>(acc, a.scale_out_g1, g1h, g1l, ovf);
        BSTAMP(5);
    }
    {
        f32x16 acc[1];
        wave_linear<8, 1, P4>(r, g1h, g1l, acc);
        BSTAMP(6);
        if (L.valid) {
            if constexpr (GAUSS) {
                // the step of this lane's elements, four at a time (accumulator registers 4 q .. 4 q + 3 are 4 consecutive
                // columns): every fetch of a group is in flight before its arithmetic starts
                const GaussRow k = (a.n_num > 8) ? gauss_row(a.g, (int)L.row) : gk;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = feat(0, 4 * q, L.h);
                    if (c0 < a.n_num) {
                        GaussIn in4[4];
                        float e4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (c0 + u < a.n_num) {
                                if (q == 0) { e4[u] = ge0[u]; in4[u] = gin0[u]; continue; }     // fetched at the start
                                e4[u] = a.g.eps[(size_t)L.row * a.g.ld_eps + c0 + u];
                                in4[u] = gauss_in(a.g, (int)L.row, c0 + u);
                            }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (c0 + u < a.n_num) {
                                const float g = acc[0][4 * q + u] * a.scale_acc_g0;           // alpha folded into the scale
                                a.gx[L.row * a.ld_gx + c0 + u] = g;
                                gauss_apply(a.g, k, in4[u], (int)L.row, c0 + u, e4[u], g);
                            }
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int c = feat(0, v, L.h);
                    if (c < a.n_num) a.gx[L.row * a.ld_gx + c] = acc[0][v] * a.scale_acc_g0;  // alpha folded into the scale
                }
            }
        }
    }
    BSTAMP(7);
    if (!(ovf <= 65504.f) && a.overflow) *a.overflow = 1;
}

// ---------------------------------------------------------------------------------------------------- denoiser head + step
// h [B, 16 * KSH] (split rows) -> eps = head(h) (n_num <= 32 columns) -> the fused Gaussian reverse step of
// diffusion_step.hip on the wavefront's 32 rows: eps never goes to memory.  Stream: the head, KSH pairs.
template <int KSH, int S0 = 0>
struct HeadChunks {
    static __device__ __forceinline__ void run(const WRing& r, const char* rowp, int h, f32x16 (&acc)[1]) {
        h8 ah[8], al[8];
        load_act<8>(rowp + (S0 >> 1) * 128, h, ah, al);
        f32x16 part[1];
        wave_linear<8, 1, S0>(r, ah, al, part);
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[0][v] += part[0][v];
        if constexpr (S0 + 8 < KSH) HeadChunks<KSH, S0 + 8>::run(r, rowp, h, acc);
    }
};

template <int KSH>
__global__ void __launch_bounds__(256) den_head_gauss_kernel(FusedHeadArgs a) {
    const Lane L = lane_of(a.g.B);
    const WRing r = make_ring(a.stream, a.n_chunks);
    stage_bias(bias_lds(), a.l.bias, nullptr, 32);
    f32x16 acc[1];
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[0][v] = 0.f;
    // the contraction is long and there is one accumulator tile: the activations stream through in chunks of 8 steps
    HeadChunks<KSH>::run(r, a.h + L.row_ld * a.pitch_h, L.h, acc);
    epi_forward<1>(acc, 0, L.h, bias_lds(), a.l.scale_acc, CLAVA_ACT_NONE);
    if (!L.valid) return;
    const GaussStepArgs& g = a.g;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int col = feat(0, v, L.h);
        if (col < g.n_num) gauss_step_element(g, (int)L.row, col, acc[0][v]);
    }
}

constexpr int FUSED_LDS = RING_LDS + BIAS_LDS_FLOATS * 4 + WAVES_PER_WG * STAGE_WAVE;
bool g_ring_attr[64][16] = {};

template <typename K, typename A>
int launch_rows(K kernel, int slot, const A& args, int B, hipStream_t st, const char* name, double flops, double bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    if (!g_ring_attr[dev][slot]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS);
        if (e != hipSuccess) {
            clava_set_error("%s: cannot reserve %d bytes of LDS: %s", name, FUSED_LDS, hipGetErrorString(e));
            return CLAVA_ERR_LAUNCH;
        }
        g_ring_attr[dev][slot] = true;
    }
    const int wgs = (B + 32 * WAVES_PER_WG - 1) / (32 * WAVES_PER_WG);
    ProfScope prof(PROF_FUSED_NARROW, flops, bytes, st);
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(64 * WAVES_PER_WG), FUSED_LDS, st, args);
    CLAVA_CHECK_LAUNCH(name);
    return CLAVA_OK;
}

int chunks_of(int pairs) { return (pairs + PAIRS_PER_CHUNK - 1) / PAIRS_PER_CHUNK; }

}  // namespace

// number of bytes of the weight stream of a fused kernel with `pairs` fragment pairs (whole ring chunks)
long long clava_fused_stream_bytes(int pairs) { return (long long)chunks_of(pairs) * RING_CHUNK; }

int clava_launch_cls_front(FusedFrontArgs a, int n_in, hipStream_t st) {
    const int ks0 = n_in <= 16 ? 1 : 2;
    a.n_chunks = chunks_of(4 * ks0 + 96);
    const double fl = 2.0 * a.B * (128.0 * n_in + 128.0 * 128 + 128.0 * 256), by = (double)a.B * (128 + 1024);
    return ks0 == 1 ? launch_rows(cls_front_kernel<1>, 0, a, a.B, st, "cls_front", fl, by)
                    : launch_rows(cls_front_kernel<2>, 1, a, a.B, st, "cls_front", fl, by);
}

bool clava_den_front_supports(int hidden) { return hidden == 256 || hidden == 512 || hidden == 1024; }

int clava_launch_den_front(FusedFrontArgs a, int n_in, hipStream_t st) {
    const int ks0 = n_in <= 16 ? 1 : 2, ng = a.nt_out / 4;
    a.n_chunks = chunks_of(4 * ks0 + 32 * ng);
    const double fl = 2.0 * a.B * (128.0 * n_in + 128.0 * 32 * a.nt_out), by = (double)a.B * (128 + 128.0 * a.nt_out);
    switch (ks0 * 16 + ng) {
        case 16 + 2: return launch_rows(den_front_kernel<1, 2>, 2, a, a.B, st, "den_front", fl, by);
        case 16 + 4: return launch_rows(den_front_kernel<1, 4>, 3, a, a.B, st, "den_front", fl, by);
        case 16 + 8: return launch_rows(den_front_kernel<1, 8>, 4, a, a.B, st, "den_front", fl, by);
        case 32 + 2: return launch_rows(den_front_kernel<2, 2>, 5, a, a.B, st, "den_front", fl, by);
        case 32 + 4: return launch_rows(den_front_kernel<2, 4>, 6, a, a.B, st, "den_front", fl, by);
        case 32 + 8: return launch_rows(den_front_kernel<2, 8>, 7, a, a.B, st, "den_front", fl, by);
    }
    clava_set_error("den_front: first hidden width %d is not built (256, 512, 1024)", 32 * a.nt_out);
    return CLAVA_ERR_ARG;
}

int clava_launch_cls_tail(FusedTailArgs a, hipStream_t st) {
    a.n_chunks = chunks_of(160);
    const double fl = 4.0 * a.B * (256.0 * 128 + 128.0 * a.C), by = (double)a.B * 2048;
    return launch_rows(cls_tail_kernel, 8, a, a.B, st, "cls_tail", fl, by);
}

int clava_launch_cls_back(FusedBackArgs a, hipStream_t st) {
    a.n_chunks = chunks_of(104);
    if (!a.mask1 || !a.g3 || a.n_num > 32) {
        clava_set_error("cls_back: needs the derivative bits cls_front stored, the gradient rows and <= 32 numeric columns");
        return CLAVA_ERR_ARG;
    }
    const double fl = 2.0 * a.B * (256.0 * 128 + 128.0 * 128 + 128.0 * a.n_num), by = (double)a.B * (1024 + 16 + 4.0 * a.n_num);
    if (a.gauss) {
        if (!a.g.eps || a.g.grad != a.gx || a.g.ld_grad != a.ld_gx || a.g.B != a.B || a.g.n_num != a.n_num) {
            clava_set_error("cls_back: the fused Gaussian step must take this launch's gradient and the denoiser's eps");
            return CLAVA_ERR_ARG;
        }
        return launch_rows(cls_back_kernel<true>, 14, a, a.B, st, "cls_back_gauss", fl, by);
    }
    return launch_rows(cls_back_kernel<false>, 9, a, a.B, st, "cls_back", fl, by);
}

bool clava_den_head_supports(int hidden) { return hidden == 256 || hidden == 512 || hidden == 1024; }

int clava_launch_den_head_gauss(FusedHeadArgs a, hipStream_t st) {
    a.n_chunks = chunks_of(a.ks);
    const double fl = 2.0 * a.g.B * 16.0 * a.ks * a.g.n_num, by = (double)a.g.B * (64.0 * a.ks + 24.0 * a.g.n_num);
    switch (a.ks) {
        case 16: return launch_rows(den_head_gauss_kernel<16>, 11, a, a.g.B, st, "den_head_gauss", fl, by);
        case 32: return launch_rows(den_head_gauss_kernel<32>, 12, a, a.g.B, st, "den_head_gauss", fl, by);
        case 64: return launch_rows(den_head_gauss_kernel<64>, 13, a, a.g.B, st, "den_head_gauss", fl, by);
    }
    clava_set_error("den_head_gauss: last hidden width %d is not built (256, 512, 1024)", 16 * a.ks);
    return CLAVA_ERR_ARG;
}

// developer aid: the stamps of the last cls_tail launch (host array of 16); not part of include/clava_hip.h
extern "C" int clava_debug_fused_stamps_(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 16) == hipSuccess ? 0 : -2;
}
