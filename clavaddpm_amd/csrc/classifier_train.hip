// Kernels of the guidance-classifier training step (reference pipeline_utils.py:655-700 + Classifier :584-630 in
// train mode): BatchNorm1d with batch statistics, Dropout, cross-entropy with top-k accuracy, and their backward.
// The Linear layers themselves run on the GEMM kernels (gemm_f32.hip).
//
// One hidden block of the classifier is  a = act(Linear(h_prev));  y = BN_train(a);  h = dropout(y).
//   forward :  bn_stats (column mean / M2 per batch slab)  ->  bn_apply (combine slabs, normalise, scale/shift, dropout,
//              running-statistics update)
//   backward:  bn_bwd_reduce (column sums of dy and dy*xhat per batch slab)  ->  bn_bwd_apply (combine slabs, publish
//              dgamma / dbeta, dz = act'(a) * gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)))
// Only `a` is kept from the forward: xhat is recomputed from (a, mean, invstd) and the dropout mask from the counter-based
// generator (or read from an injected byte mask in parity tests).  Every reduction has a fixed order (bitwise
// reproducible, no atomics).
//
// All four kernels are HBM/L2-bound streams over [B, C] float32.  A workgroup owns 128 columns (512 contiguous bytes per
// row: 32 lanes x float4, two rows per wavefront) and a batch slab; slabs are sized so that even the 1024-wide layer at
// batch 4096 launches 512 workgroups.
#include "common.h"
#include "rng.h"
#include "clava_hip.h"

namespace {

constexpr int BN_MAX_SLABS = 64;
constexpr int BN_COLS = 128;       // columns per workgroup
constexpr int BN_ROWS = 64;        // rows per workgroup of the elementwise kernels

struct BnFwdArgs {
    const float* a; int lda;          // [B, C] post-activation input of the BatchNorm
    int B, C, S;                      // S batch slabs
    float* part;                      // [2][S][C]: slab mean, slab M2
    const float* gamma; const float* beta;
    float* run_mean; float* run_var;  // updated in place (momentum), or null
    float momentum, eps;
    float* save_mean; float* save_invstd;   // [C]
    float* h; int ldh;                // [B, C] output
    const unsigned char* mask; int ldmask;  // injected keep mask (1 = keep) or null
    float p;                          // dropout probability (0: no dropout)
    unsigned long long seed; unsigned int step, layer;
    const unsigned int* step_dev;     // when not null the step index is read from device memory (captured graphs)
    // optional second copy of h as split rows (csrc/split.h): the operand of the next block's split-f16 product
    char* xs = nullptr; long long xs_pitch = 0; float xs_scale = 1.f; int* xs_overflow = nullptr;
};

__device__ __forceinline__ int slab_rows(int B, int S) { return (B + S - 1) / S; }

// 4 consecutive columns [col, col+4) of row r; VEC: one 16-byte access (C % 4 == 0, 16-byte aligned rows)
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int ld, int r, int col, int C) {
    if (VEC) return *reinterpret_cast<const float4*>(p + (size_t)r * ld + col);
    float4 v = {0.f, 0.f, 0.f, 0.f};
    const float* q = p + (size_t)r * ld + col;
    if (col + 0 < C) v.x = q[0];
    if (col + 1 < C) v.y = q[1];
    if (col + 2 < C) v.z = q[2];
    if (col + 3 < C) v.w = q[3];
    return v;
}
template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ p, int ld, int r, int col, int C, float4 v) {
    if (VEC) { *reinterpret_cast<float4*>(p + (size_t)r * ld + col) = v; return; }
    float* q = p + (size_t)r * ld + col;
    if (col + 0 < C) q[0] = v.x;
    if (col + 1 < C) q[1] = v.y;
    if (col + 2 < C) q[2] = v.z;
    if (col + 3 < C) q[3] = v.w;
}

// the same 4 columns as split rows: hi = f16(v * scale), lo = f16(v * scale - hi); 32-column blocks of 64 B hi | 64 B lo
typedef _Float16 bn_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_split4(char* base, long long pitch, float scale, int* overflow, int r, int col, float4 v) {
    const float s[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    bn_h4 hi, lo;
    bool ovf = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ovf |= !(fabsf(s[i]) <= 65504.f);
        hi[i] = (_Float16)s[i];
        lo[i] = (_Float16)(s[i] - (float)hi[i]);
    }
    char* p = base + (long long)r * pitch + (long long)(col >> 5) * 128 + ((col & 31) << 1);
    *reinterpret_cast<bn_h4*>(p) = hi;
    *reinterpret_cast<bn_h4*>(p + 64) = lo;
    if (ovf && overflow) *overflow = 1;
}

// keep flags of the 4 consecutive columns [c4*4, c4*4+4) of row r
__device__ __forceinline__ void keep4(const unsigned char* mask, int ldmask, float p, unsigned long long seed, unsigned int step,
                                      unsigned int layer, int r, int c4, int C, bool k[4]) {
    if (mask) {
#pragma unroll
        for (int i = 0; i < 4; ++i) k[i] = (c4 * 4 + i < C) ? mask[(size_t)r * ldmask + c4 * 4 + i] != 0 : false;
    } else if (p > 0.f) {
        const unsigned long long elem = (unsigned long long)r * (unsigned long long)((C + 3) / 4) + c4;
        const clava_rng::u4 w = clava_rng::philox4x32_10(seed, elem, step * 32u + layer, clava_rng::STREAM_TRAIN);
        k[0] = clava_rng::to_uniform(w.x) >= p; k[1] = clava_rng::to_uniform(w.y) >= p;
        k[2] = clava_rng::to_uniform(w.z) >= p; k[3] = clava_rng::to_uniform(w.w) >= p;
    } else {
        k[0] = k[1] = k[2] = k[3] = true;
    }
}

// LDS reduction of per-thread float4 partials over the 8 row lanes of a workgroup; result valid in threads < 32
__device__ __forceinline__ float4 reduce_rows(float4 v, float (*sh)[BN_COLS + 4]) {
    const int cg = threadIdx.x & 31, q = threadIdx.x >> 5;
    *reinterpret_cast<float4*>(&sh[q][cg * 4]) = v;
    __syncthreads();
    float4 t = {0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 u = *reinterpret_cast<const float4*>(&sh[i][cg * 4]);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
    }
    __syncthreads();
    return t;
}

// grid (ceil(C/128), S), 256 threads = 32 column groups (4 columns each) x 8 row lanes
template <bool VEC>
__global__ void __launch_bounds__(256) bn_stats_kernel(BnFwdArgs g) {
    __shared__ __align__(16) float sh[8][BN_COLS + 4];
    __shared__ __align__(16) float mean_s[BN_COLS];
    const int cg = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int col = blockIdx.x * BN_COLS + cg * 4;
    const bool live = col < g.C;
    const int rp = slab_rows(g.B, g.S);
    const int r0 = blockIdx.y * rp, r1 = min(g.B, r0 + rp);
    const int n = r1 - r0;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live)
        for (int r = r0 + q; r < r1; r += 8) {
            const float4 v = load4<VEC>(g.a, g.lda, r, col, g.C);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    float4 t = reduce_rows(acc, sh);
    if (threadIdx.x < 32) {
        const float inv_n = n > 0 ? 1.0f / (float)n : 0.f;
        t.x *= inv_n; t.y *= inv_n; t.z *= inv_n; t.w *= inv_n;
        *reinterpret_cast<float4*>(&mean_s[cg * 4]) = t;
    }
    __syncthreads();
    const float4 mu = *reinterpret_cast<const float4*>(&mean_s[cg * 4]);
    acc = {0.f, 0.f, 0.f, 0.f};
    if (live)
        for (int r = r0 + q; r < r1; r += 8) {         // second pass: the slab is L2-resident
            const float4 v = load4<VEC>(g.a, g.lda, r, col, g.C);
            const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
            acc.x += dx * dx; acc.y += dy * dy; acc.z += dz * dz; acc.w += dw * dw;
        }
    t = reduce_rows(acc, sh);
    if (threadIdx.x < 32 && live) {
        store4<VEC>(g.part, g.C, blockIdx.y, col, g.C, mu);
        store4<VEC>(g.part, g.C, g.S + blockIdx.y, col, g.C, t);
    }
}

// grid (ceil(C/128), ceil(B/64)): combine the slab statistics of this workgroup's columns (Chan's update, fixed order),
// then normalise 64 rows.  Row block 0 also publishes mean / invstd and updates the running statistics
// (torch: running = (1-m)*running + m*batch, with the unbiased batch variance).
template <bool VEC>
__global__ void __launch_bounds__(256) bn_apply_kernel(BnFwdArgs g) {
    __shared__ __align__(16) float s_mean[BN_COLS], s_k[BN_COLS], s_beta[BN_COLS];
    __shared__ float h_cnt[BN_COLS], h_mean[BN_COLS], h_m2[BN_COLS];
    {
        // two threads per column, each folds half of the slabs (loads issued eight at a time so that their latencies
        // overlap), then thread c merges the halves: a fixed order, hence the same bits in every workgroup
        const int cl = threadIdx.x & (BN_COLS - 1), half = threadIdx.x >> 7;
        const int c = blockIdx.x * BN_COLS + cl;
        const int rp = slab_rows(g.B, g.S);
        const int sh = (g.S + 1) / 2, s0 = half * sh, s1 = min(g.S, s0 + sh);
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
        if (c < g.C)
            for (int sb = s0; sb < s1; sb += 8) {
                float ms[8], qs[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int s = min(sb + i, s1 - 1);
                    ms[i] = g.part[(size_t)s * g.C + c];
                    qs[i] = g.part[(size_t)(g.S + s) * g.C + c];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int s = sb + i;
                    const float nb = s < s1 ? (float)max(0, min(g.B, (s + 1) * rp) - s * rp) : 0.f;
                    if (nb > 0.f) {
                        const float tot = cnt + nb, delta = ms[i] - mean;
                        mean += delta * (nb / tot);
                        m2 += qs[i] + delta * delta * (cnt * nb / tot);
                        cnt = tot;
                    }
                }
            }
        if (half == 1) { h_cnt[cl] = cnt; h_mean[cl] = mean; h_m2[cl] = m2; }
        __syncthreads();
        if (half == 0) {
            float inv = 0.f, ga = 0.f, be = 0.f;
            if (c < g.C) {
                const float nb = h_cnt[cl];
                if (nb > 0.f) {
                    const float tot = cnt + nb, delta = h_mean[cl] - mean;
                    mean += delta * (nb / tot);
                    m2 += h_m2[cl] + delta * delta * (cnt * nb / tot);
                }
                const float var = m2 / (float)g.B;
                inv = 1.0f / sqrtf(var + g.eps);
                ga = g.gamma[c]; be = g.beta[c];
                if (blockIdx.y == 0) {
                    g.save_mean[c] = mean;
                    g.save_invstd[c] = inv;
                    if (g.run_mean) {
                        const float unb = g.B > 1 ? m2 / (float)(g.B - 1) : var;
                        g.run_mean[c] = (1.0f - g.momentum) * g.run_mean[c] + g.momentum * mean;
                        g.run_var[c] = (1.0f - g.momentum) * g.run_var[c] + g.momentum * unb;
                    }
                }
            }
            s_mean[cl] = mean; s_k[cl] = inv * ga; s_beta[cl] = be;
        }
    }
    __syncthreads();
    const int cg = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int col = blockIdx.x * BN_COLS + cg * 4;
    if (col >= g.C) return;
    const float scale = g.p > 0.f ? 1.0f / (1.0f - g.p) : 1.0f;
    const unsigned int step = g.step_dev ? g.step_dev[0] : g.step;
    const float4 mu = *reinterpret_cast<const float4*>(&s_mean[cg * 4]);
    const float4 kk = *reinterpret_cast<const float4*>(&s_k[cg * 4]);
    const float4 be = *reinterpret_cast<const float4*>(&s_beta[cg * 4]);
    const int r1 = min(g.B, (int)(blockIdx.y + 1) * BN_ROWS);
    for (int r = blockIdx.y * BN_ROWS + q; r < r1; r += 8) {
        bool k[4];
        keep4(g.mask, g.ldmask, g.p, g.seed, step, g.layer, r, col >> 2, g.C, k);
        const float4 v = load4<VEC>(g.a, g.lda, r, col, g.C);
        float4 o;
        o.x = k[0] ? ((v.x - mu.x) * kk.x + be.x) * scale : 0.f;
        o.y = k[1] ? ((v.y - mu.y) * kk.y + be.y) * scale : 0.f;
        o.z = k[2] ? ((v.z - mu.z) * kk.z + be.z) * scale : 0.f;
        o.w = k[3] ? ((v.w - mu.w) * kk.w + be.w) * scale : 0.f;
        store4<VEC>(g.h, g.ldh, r, col, g.C, o);
        if (g.xs) store_split4(g.xs, g.xs_pitch, g.xs_scale, g.xs_overflow, r, col, o);
    }
}

struct BnBwdArgs {
    const float* dh; int lddh;        // [B, C] gradient w.r.t. the dropout output
    const float* a; int lda;          // [B, C] saved post-activation
    int B, C, S;                      // S batch slabs of the reduction
    float* part;                      // [2][S][C]: slab sums of dy, of dy*xhat
    const float* mean; const float* invstd; const float* gamma;
    const unsigned char* mask; int ldmask; float p;
    unsigned long long seed; unsigned int step, layer;
    const unsigned int* step_dev;
    float* dgamma; float* dbeta; long long slab_stride; int splits;   // gradient slabs: slab 0 gets the sum, the others 0
    int act; float slope;             // activation in front of the BatchNorm: act'(a) = a > 0 ? 1 : slope
    float* dz; int lddz;              // [B, C] gradient w.r.t. the Linear output
    // optional second copy of dz as split rows: the operand of this block's split-f16 input- and weight-gradient products
    char* xs = nullptr; long long xs_pitch = 0; float xs_scale = 1.f; int* xs_overflow = nullptr;
};

// grid (ceil(C/128), S): slab sums of dy = dh*keep/(1-p) and of dy*xhat
template <bool VEC>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(BnBwdArgs g) {
    __shared__ __align__(16) float sh[8][BN_COLS + 4];
    const int cg = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int col = blockIdx.x * BN_COLS + cg * 4;
    const bool live = col < g.C;
    const int rp = slab_rows(g.B, g.S);
    const int r0 = blockIdx.y * rp, r1 = min(g.B, r0 + rp);
    const float scale = g.p > 0.f ? 1.0f / (1.0f - g.p) : 1.0f;
    const unsigned int step = g.step_dev ? g.step_dev[0] : g.step;
    float4 sb = {0.f, 0.f, 0.f, 0.f}, sg = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float4 mu = load4<VEC>(g.mean, 0, 0, col, g.C), inv = load4<VEC>(g.invstd, 0, 0, col, g.C);
        for (int r = r0 + q; r < r1; r += 8) {
            bool k[4];
            keep4(g.mask, g.ldmask, g.p, g.seed, step, g.layer, r, col >> 2, g.C, k);
            const float4 d = load4<VEC>(g.dh, g.lddh, r, col, g.C);
            const float4 v = load4<VEC>(g.a, g.lda, r, col, g.C);
            const float d0 = k[0] ? d.x * scale : 0.f, d1 = k[1] ? d.y * scale : 0.f;
            const float d2 = k[2] ? d.z * scale : 0.f, d3 = k[3] ? d.w * scale : 0.f;
            sb.x += d0; sb.y += d1; sb.z += d2; sb.w += d3;
            sg.x += d0 * ((v.x - mu.x) * inv.x); sg.y += d1 * ((v.y - mu.y) * inv.y);
            sg.z += d2 * ((v.z - mu.z) * inv.z); sg.w += d3 * ((v.w - mu.w) * inv.w);
        }
    }
    const float4 tb = reduce_rows(sb, sh);
    const float4 tg = reduce_rows(sg, sh);
    if (threadIdx.x < 32 && live) {
        store4<VEC>(g.part, g.C, blockIdx.y, col, g.C, tb);
        store4<VEC>(g.part, g.C, g.S + blockIdx.y, col, g.C, tg);
    }
}

// grid (ceil(C/128), ceil(B/64))
template <bool VEC>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(BnBwdArgs g) {
    __shared__ __align__(16) float s_mean[BN_COLS], s_inv[BN_COLS], s_k[BN_COLS], s_db[BN_COLS], s_dg[BN_COLS];
    __shared__ float h_db[BN_COLS], h_dg[BN_COLS];
    {
        const int cl = threadIdx.x & (BN_COLS - 1), half = threadIdx.x >> 7;
        const int c = blockIdx.x * BN_COLS + cl;
        const int sh = (g.S + 1) / 2, s0 = half * sh, s1 = min(g.S, s0 + sh);
        float db = 0.f, dg = 0.f;
        if (c < g.C)
            for (int sb = s0; sb < s1; sb += 8) {
                float b8[8], g8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int s = min(sb + i, s1 - 1);
                    b8[i] = g.part[(size_t)s * g.C + c];
                    g8[i] = g.part[(size_t)(g.S + s) * g.C + c];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (sb + i < s1) { db += b8[i]; dg += g8[i]; }
            }
        if (half == 1) { h_db[cl] = db; h_dg[cl] = dg; }
        __syncthreads();
        if (half == 0) {
            float mean = 0.f, inv = 0.f, ga = 0.f;
            if (c < g.C) {
                db += h_db[cl]; dg += h_dg[cl];
                mean = g.mean[c]; inv = g.invstd[c]; ga = g.gamma[c];
                if (blockIdx.y == 0) {
                    g.dbeta[c] = db;
                    g.dgamma[c] = dg;
                    for (int s = 1; s < g.splits; ++s) {
                        g.dbeta[(size_t)s * g.slab_stride + c] = 0.f;
                        g.dgamma[(size_t)s * g.slab_stride + c] = 0.f;
                    }
                }
            }
            s_mean[cl] = mean; s_inv[cl] = inv; s_k[cl] = ga * inv;
            s_db[cl] = db / (float)g.B; s_dg[cl] = dg / (float)g.B;
        }
    }
    __syncthreads();
    const int cg = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int col = blockIdx.x * BN_COLS + cg * 4;
    if (col >= g.C) return;
    const float scale = g.p > 0.f ? 1.0f / (1.0f - g.p) : 1.0f;
    const unsigned int step = g.step_dev ? g.step_dev[0] : g.step;
    const float4 mu = *reinterpret_cast<const float4*>(&s_mean[cg * 4]);
    const float4 inv = *reinterpret_cast<const float4*>(&s_inv[cg * 4]);
    const float4 kk = *reinterpret_cast<const float4*>(&s_k[cg * 4]);
    const float4 mb = *reinterpret_cast<const float4*>(&s_db[cg * 4]);
    const float4 mg = *reinterpret_cast<const float4*>(&s_dg[cg * 4]);
    const bool has_act = g.act != CLAVA_ACT_NONE;
    const float slope = g.slope;
    const int r1 = min(g.B, (int)(blockIdx.y + 1) * BN_ROWS);
    for (int r = blockIdx.y * BN_ROWS + q; r < r1; r += 8) {
        bool k[4];
        keep4(g.mask, g.ldmask, g.p, g.seed, step, g.layer, r, col >> 2, g.C, k);
        const float4 d = load4<VEC>(g.dh, g.lddh, r, col, g.C);
        const float4 v = load4<VEC>(g.a, g.lda, r, col, g.C);
        float4 o;
        o.x = kk.x * ((k[0] ? d.x * scale : 0.f) - mb.x - ((v.x - mu.x) * inv.x) * mg.x);
        o.y = kk.y * ((k[1] ? d.y * scale : 0.f) - mb.y - ((v.y - mu.y) * inv.y) * mg.y);
        o.z = kk.z * ((k[2] ? d.z * scale : 0.f) - mb.z - ((v.z - mu.z) * inv.z) * mg.z);
        o.w = kk.w * ((k[3] ? d.w * scale : 0.f) - mb.w - ((v.w - mu.w) * inv.w) * mg.w);
        if (has_act) {
            o.x *= v.x > 0.f ? 1.0f : slope; o.y *= v.y > 0.f ? 1.0f : slope;
            o.z *= v.z > 0.f ? 1.0f : slope; o.w *= v.w > 0.f ? 1.0f : slope;
        }
        store4<VEC>(g.dz, g.lddz, r, col, g.C, o);
        if (g.xs) store_split4(g.xs, g.xs_pitch, g.xs_scale, g.xs_overflow, r, col, o);
    }
}

// ---------------------------------------------------------------------------------------------- cross-entropy
// One wavefront per row: loss = logsumexp(z) - z[y] (F.cross_entropy, reduction='none'), acc@1 / acc@5 as
// compute_top_k does (pipeline_utils.py:640-645; rank = number of classes ahead of the label, ties to the lower index
// like torch.topk), and the row's logsumexp for the gradient kernel.
__global__ void __launch_bounds__(256) ce_rows_kernel(const float* __restrict__ z, int ldz, const long long* __restrict__ y,
                                                      int B, int C, float* __restrict__ loss, float* __restrict__ acc1,
                                                      float* __restrict__ acc5, float* __restrict__ lse_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* zr = z + (size_t)row * ldz;
    const int yy = (int)y[row];
    if (yy < 0 || yy >= C) {                      // invalid label: poison the row instead of reading out of bounds
        if (lane == 0) {
            loss[row] = NAN; lse_out[row] = NAN;
            if (acc1) acc1[row] = 0.f;
            if (acc5) acc5[row] = 0.f;
        }
        return;
    }
    const float zy = zr[yy];
    float mx = -INFINITY;
    int ahead = 0;
    for (int c = lane; c < C; c += 64) {
        const float v = zr[c];
        mx = fmaxf(mx, v);
        ahead += (v > zy || (v == zy && c < yy)) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        ahead += __shfl_xor(ahead, o);
    }
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(zr[c] - mx);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float lse = mx + logf(s);
        loss[row] = lse - zy;
        lse_out[row] = lse;
        if (acc1) acc1[row] = ahead < 1 ? 1.f : 0.f;
        if (acc5) acc5[row] = ahead < 5 ? 1.f : 0.f;
    }
}

// dlogits = factor * scale[0] * (softmax(z) - onehot(y)); `scale` is a device scalar (the batch-mean loss: the
// reference calls loss.backward(loss), pipeline_utils.py:700) or null
__global__ void __launch_bounds__(256) ce_grad_kernel(const float* __restrict__ z, int ldz, const long long* __restrict__ y,
                                                      const float* __restrict__ lse, int B, int C, const float* __restrict__ scale,
                                                      float factor, float* __restrict__ dz, int lddz) {
    const float f = factor * (scale ? scale[0] : 1.0f);
    const long long n = (long long)B * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i - (long long)r * C);
        const float p = expf(z[(size_t)r * ldz + c] - lse[r]);
        dz[(size_t)r * lddz + c] = f * (p - ((int)y[r] == c ? 1.0f : 0.0f));
    }
}

// every workgroup of the elementwise kernels folds all slab statistics of its columns again (a chain of Chan updates): the
// slab count trades parallelism of the reductions against that prologue (CLAVA_BN_SLABS, at most BN_MAX_SLABS)
int bn_slab_cap() {
    static const int cap = [] {
        const char* v = getenv("CLAVA_BN_SLABS");
        const int c = v ? atoi(v) : BN_MAX_SLABS;
        return c < 1 ? 1 : (c > BN_MAX_SLABS ? BN_MAX_SLABS : c);
    }();
    return cap;
}
int pick_slabs(int B, int C) {
    const int colblocks = (C + BN_COLS - 1) / BN_COLS;
    int s = (512 + colblocks - 1) / colblocks;
    s = min(s, bn_slab_cap());
    s = min(s, max(1, B / 32));
    return max(1, s);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

long long clava_bn_workspace_floats(int C) { return 2LL * BN_MAX_SLABS * (long long)C; }

int clava_bn_dropout_train_split_f32(const float* a, int lda, int B, int C, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float momentum, float eps, float p, const unsigned char* keep_mask,
                                     int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev,
                                     unsigned int layer, float* workspace, float* save_mean, float* save_invstd, float* h, int ldh,
                                     void* h_split, long long split_pitch, int split_shift, int* overflow, void* stream);

int clava_bn_dropout_train_f32(const float* a, int lda, int B, int C, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, float p, const unsigned char* keep_mask,
                               int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev, unsigned int layer,
                               float* workspace, float* save_mean, float* save_invstd, float* h, int ldh, void* stream) {
    return clava_bn_dropout_train_split_f32(a, lda, B, C, gamma, beta, running_mean, running_var, momentum, eps, p, keep_mask, ldmask, seed,
                                            step, step_dev, layer, workspace, save_mean, save_invstd, h, ldh, nullptr, 0, 0, nullptr, stream);
}

int clava_bn_dropout_train_split_f32(const float* a, int lda, int B, int C, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float momentum, float eps, float p, const unsigned char* keep_mask,
                                     int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev,
                                     unsigned int layer, float* workspace, float* save_mean, float* save_invstd, float* h, int ldh,
                                     void* h_split, long long split_pitch, int split_shift, int* overflow, void* stream) {
    CLAVA_CHECK_ARG(!h_split || (C % 4 == 0 && split_pitch >= 4LL * ((C + 31) & ~31) && split_pitch % 8 == 0 && aligned16(h_split) &&
                                 split_shift >= -24 && split_shift <= 24),
                    "bn_dropout_train: the split copy needs C %% 4 == 0 and rows of >= 4*round_up(C,32) bytes");
    CLAVA_CHECK_ARG(a && gamma && beta && workspace && save_mean && save_invstd && h && B > 0 && C > 0 && lda >= C && ldh >= C,
                    "bn_dropout_train: bad arguments");
    CLAVA_CHECK_ARG(p >= 0.f && p < 1.f && (!keep_mask || ldmask >= C) && (!running_mean == !running_var) && layer < 32,
                    "bn_dropout_train: bad dropout / running-statistics arguments");
    BnFwdArgs g;
    g.a = a; g.lda = lda; g.B = B; g.C = C; g.S = pick_slabs(B, C); g.part = workspace; g.gamma = gamma; g.beta = beta;
    g.run_mean = running_mean; g.run_var = running_var; g.momentum = momentum; g.eps = eps; g.save_mean = save_mean;
    g.save_invstd = save_invstd; g.h = h; g.ldh = ldh; g.mask = keep_mask; g.ldmask = ldmask; g.p = p; g.seed = seed; g.step = step;
    g.step_dev = step_dev; g.layer = layer;
    CLAVA_CHECK_ARG(!h_split || (C % 4 == 0 && split_pitch >= 4LL * ((C + 31) / 32 * 32) && split_pitch % 8 == 0 && aligned16(h_split)),
                    "bn_dropout_train_split: split rows need C % 4 == 0 and a pitch of at least 4 * roundup32(C) bytes");
    g.xs = (char*)h_split; g.xs_pitch = split_pitch; g.xs_scale = ldexpf(1.f, split_shift); g.xs_overflow = overflow;
    const bool vec = C % 4 == 0 && lda % 4 == 0 && ldh % 4 == 0 && aligned16(a) && aligned16(h) && aligned16(workspace);
    const dim3 gs((C + BN_COLS - 1) / BN_COLS, g.S), ga((C + BN_COLS - 1) / BN_COLS, (B + BN_ROWS - 1) / BN_ROWS);
    {
        ProfScope prof(PROF_BN, 0.0, 8.0 * B * C, (hipStream_t)stream);
        if (vec) hipLaunchKernelGGL(bn_stats_kernel<true>, gs, dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(bn_stats_kernel<false>, gs, dim3(256), 0, (hipStream_t)stream, g);
        CLAVA_CHECK_LAUNCH("clava_bn_dropout_train_f32(stats)");
    }
    {
        ProfScope prof(PROF_BN, 0.0, 8.0 * B * C, (hipStream_t)stream);
        if (vec) hipLaunchKernelGGL(bn_apply_kernel<true>, ga, dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(bn_apply_kernel<false>, ga, dim3(256), 0, (hipStream_t)stream, g);
        CLAVA_CHECK_LAUNCH("clava_bn_dropout_train_f32(apply)");
    }
    return CLAVA_OK;
}

int clava_bn_dropout_backward_split_f32(const float* dh, int lddh, const float* a, int lda, int B, int C, const float* save_mean,
                                        const float* save_invstd, const float* gamma, float p, const unsigned char* keep_mask,
                                        int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev,
                                        unsigned int layer, int act, float slope, float* workspace, int splits, float* dgamma_slabs,
                                        float* dbeta_slabs, long long slab_stride, float* dz, int lddz, void* dz_split,
                                        long long split_pitch, int split_shift, int* overflow, void* stream);

int clava_bn_dropout_backward_f32(const float* dh, int lddh, const float* a, int lda, int B, int C, const float* save_mean,
                                  const float* save_invstd, const float* gamma, float p, const unsigned char* keep_mask,
                                  int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev, unsigned int layer,
                                  int act, float slope, float* workspace, int splits, float* dgamma_slabs, float* dbeta_slabs,
                                  long long slab_stride, float* dz, int lddz, void* stream) {
    return clava_bn_dropout_backward_split_f32(dh, lddh, a, lda, B, C, save_mean, save_invstd, gamma, p, keep_mask, ldmask, seed, step,
                                               step_dev, layer, act, slope, workspace, splits, dgamma_slabs, dbeta_slabs, slab_stride, dz,
                                               lddz, nullptr, 0, 0, nullptr, stream);
}

int clava_bn_dropout_backward_split_f32(const float* dh, int lddh, const float* a, int lda, int B, int C, const float* save_mean,
                                        const float* save_invstd, const float* gamma, float p, const unsigned char* keep_mask,
                                        int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev,
                                        unsigned int layer, int act, float slope, float* workspace, int splits, float* dgamma_slabs,
                                        float* dbeta_slabs, long long slab_stride, float* dz, int lddz, void* dz_split,
                                        long long split_pitch, int split_shift, int* overflow, void* stream) {
    CLAVA_CHECK_ARG(!dz_split || (C % 4 == 0 && split_pitch >= 4LL * ((C + 31) & ~31) && split_pitch % 8 == 0 && aligned16(dz_split) &&
                                  split_shift >= -24 && split_shift <= 24),
                    "bn_dropout_backward: the split copy needs C %% 4 == 0 and rows of >= 4*round_up(C,32) bytes");
    CLAVA_CHECK_ARG(dh && a && save_mean && save_invstd && gamma && workspace && dgamma_slabs && dbeta_slabs && dz && B > 0 && C > 0 &&
                        lddh >= C && lda >= C && lddz >= C && splits >= 1 && splits <= 64,
                    "bn_dropout_backward: bad arguments");
    CLAVA_CHECK_ARG(p >= 0.f && p < 1.f && (!keep_mask || ldmask >= C) && layer < 32, "bn_dropout_backward: bad dropout arguments");
    BnBwdArgs g;
    g.dh = dh; g.lddh = lddh; g.a = a; g.lda = lda; g.B = B; g.C = C; g.S = pick_slabs(B, C); g.part = workspace;
    g.mean = save_mean; g.invstd = save_invstd; g.gamma = gamma; g.mask = keep_mask; g.ldmask = ldmask; g.p = p; g.seed = seed;
    g.step = step; g.step_dev = step_dev; g.layer = layer; g.dgamma = dgamma_slabs; g.dbeta = dbeta_slabs;
    g.slab_stride = slab_stride; g.splits = splits; g.act = act; g.slope = slope; g.dz = dz; g.lddz = lddz;
    CLAVA_CHECK_ARG(!dz_split || (C % 4 == 0 && split_pitch >= 4LL * ((C + 31) / 32 * 32) && split_pitch % 8 == 0 && aligned16(dz_split)),
                    "bn_dropout_backward_split: split rows need C % 4 == 0 and a pitch of at least 4 * roundup32(C) bytes");
    g.xs = (char*)dz_split; g.xs_pitch = split_pitch; g.xs_scale = ldexpf(1.f, split_shift); g.xs_overflow = overflow;
    const bool vec = C % 4 == 0 && lda % 4 == 0 && lddh % 4 == 0 && lddz % 4 == 0 && aligned16(a) && aligned16(dh) && aligned16(dz) &&
                     aligned16(workspace) && aligned16(save_mean) && aligned16(save_invstd);
    const dim3 gs((C + BN_COLS - 1) / BN_COLS, g.S), ga((C + BN_COLS - 1) / BN_COLS, (B + BN_ROWS - 1) / BN_ROWS);
    {
        ProfScope prof(PROF_BN, 0.0, 8.0 * B * C, (hipStream_t)stream);
        if (vec) hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, gs, dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, gs, dim3(256), 0, (hipStream_t)stream, g);
        CLAVA_CHECK_LAUNCH("clava_bn_dropout_backward_f32(reduce)");
    }
    {
        ProfScope prof(PROF_BN, 0.0, 12.0 * B * C, (hipStream_t)stream);
        if (vec) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, ga, dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, ga, dim3(256), 0, (hipStream_t)stream, g);
        CLAVA_CHECK_LAUNCH("clava_bn_dropout_backward_f32(apply)");
    }
    return CLAVA_OK;
}

int clava_cross_entropy_rows_f32(const float* logits, int ld, const long long* labels, int B, int C, float* loss_rows, float* acc1,
                                 float* acc5, float* lse, void* stream) {
    CLAVA_CHECK_ARG(logits && labels && loss_rows && lse && B > 0 && C > 0 && ld >= C, "cross_entropy_rows: bad arguments");
    ProfScope prof(PROF_LOSS, 0.0, 4.0 * B * C, (hipStream_t)stream);
    hipLaunchKernelGGL(ce_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, B, C, loss_rows, acc1,
                       acc5, lse);
    CLAVA_CHECK_LAUNCH("clava_cross_entropy_rows_f32");
    return CLAVA_OK;
}

int clava_cross_entropy_grad_f32(const float* logits, int ld, const long long* labels, const float* lse, int B, int C,
                                 const float* scale, float factor, float* dlogits, int ldd, void* stream) {
    CLAVA_CHECK_ARG(logits && labels && lse && dlogits && B > 0 && C > 0 && ld >= C && ldd >= C, "cross_entropy_grad: bad arguments");
    ProfScope prof(PROF_LOSS, 0.0, 8.0 * B * C, (hipStream_t)stream);
    const long long blocks = ((long long)B * C + 255) / 256;
    hipLaunchKernelGGL(ce_grad_kernel, dim3((int)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, logits, ld,
                       labels, lse, B, C, scale, factor, dlogits, ldd);
    CLAVA_CHECK_LAUNCH("clava_cross_entropy_grad_f32");
    return CLAVA_OK;
}

}  // extern "C"
