// Kernels of the guidance-classifier training step (reference pipeline_utils.py:655-700 + Classifier :584-630 in
// train mode): BatchNorm1d with batch statistics, Dropout, cross-entropy with top-k accuracy, and their backward.
// The Linear layers themselves run on the GEMM kernels (gemm_f32.hip).
//
// One hidden block of the classifier is  a = act(Linear(h_prev));  y = BN_train(a);  h = dropout(y).
//   forward :  bn_stats (column mean / M2 per batch slab)  ->  bn_apply (combine slabs, normalise, scale/shift, dropout,
//              running-statistics update)
//   backward:  bn_bwd_reduce (column sums of dy and dy*xhat per batch slab, written straight into the gradient slabs of
//              beta / gamma)  ->  bn_bwd_apply (dz = act'(a) * gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)))
// Only `a` is kept from the forward: xhat is recomputed from (a, mean, invstd) and the dropout mask from the counter-based
// generator (or read from an injected byte mask in parity tests).  Every reduction has a fixed order (bitwise
// reproducible, no atomics).  All kernels are HBM/L2-bound streams over [B, C] float32.
#include "common.h"
#include "rng.h"
#include "clava_hip.h"

namespace {

constexpr int BN_MAX_SLABS = 32;

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
};

__device__ __forceinline__ int slab_rows(int B, int S) { return (B + S - 1) / S; }

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

// grid (ceil(C/32), S), 256 threads = 8 column groups (4 columns each) x 32 row lanes
__global__ void __launch_bounds__(256) bn_stats_kernel(BnFwdArgs g) {
    __shared__ float part[32][33];
    __shared__ float mean_s[32];
    const int cg = threadIdx.x & 7, q = threadIdx.x >> 3;
    const int col = blockIdx.x * 32 + cg * 4;
    const int rp = slab_rows(g.B, g.S);
    const int r0 = blockIdx.y * rp, r1 = min(g.B, r0 + rp);
    const int n = r1 - r0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + q; r < r1; r += 32)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col + c < g.C) acc[c] += g.a[(size_t)r * g.lda + col + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) part[q][cg * 4 + c] = acc[c];
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += part[i][threadIdx.x];
        mean_s[threadIdx.x] = n > 0 ? t / (float)n : 0.f;
    }
    __syncthreads();
    float mu[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { mu[c] = mean_s[cg * 4 + c]; acc[c] = 0.f; }
    for (int r = r0 + q; r < r1; r += 32)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col + c < g.C) {
                const float d = g.a[(size_t)r * g.lda + col + c] - mu[c];
                acc[c] += d * d;
            }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) part[q][cg * 4 + c] = acc[c];
    __syncthreads();
    if (threadIdx.x < 32 && blockIdx.x * 32 + threadIdx.x < g.C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += part[i][threadIdx.x];
        const int c = blockIdx.x * 32 + threadIdx.x;
        g.part[(size_t)blockIdx.y * g.C + c] = mean_s[threadIdx.x];
        g.part[(size_t)(g.S + blockIdx.y) * g.C + c] = t;
    }
}

// grid (ceil(C/32), ceil(B/128)): combine the slab statistics of this block's 32 columns (Chan's update, fixed order),
// then normalise 128 rows.  Row block 0 also publishes mean / invstd and updates the running statistics
// (torch: running = (1-m)*running + m*batch, with the unbiased batch variance).
__global__ void __launch_bounds__(256) bn_apply_kernel(BnFwdArgs g) {
    __shared__ float s_mean[32], s_inv[32], s_gamma[32], s_beta[32];
    if (threadIdx.x < 32) {
        const int c = blockIdx.x * 32 + threadIdx.x;
        float mean = 0.f, m2 = 0.f, inv = 0.f, ga = 0.f, be = 0.f;
        if (c < g.C) {
            const int rp = slab_rows(g.B, g.S);
            float cnt = 0.f;
            for (int s = 0; s < g.S; ++s) {
                const float nb = (float)max(0, min(g.B, (s + 1) * rp) - s * rp);
                if (nb <= 0.f) continue;
                const float ms = g.part[(size_t)s * g.C + c], qs = g.part[(size_t)(g.S + s) * g.C + c];
                const float tot = cnt + nb, delta = ms - mean;
                mean += delta * (nb / tot);
                m2 += qs + delta * delta * (cnt * nb / tot);
                cnt = tot;
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
        s_mean[threadIdx.x] = mean; s_inv[threadIdx.x] = inv; s_gamma[threadIdx.x] = ga; s_beta[threadIdx.x] = be;
    }
    __syncthreads();
    const int cg = threadIdx.x & 7, q = threadIdx.x >> 3;
    const int col = blockIdx.x * 32 + cg * 4;
    if (col >= g.C) return;
    const float scale = g.p > 0.f ? 1.0f / (1.0f - g.p) : 1.0f;
    const unsigned int step = g.step_dev ? g.step_dev[0] : g.step;
    const int r1 = min(g.B, (int)(blockIdx.y + 1) * 128);
    for (int r = blockIdx.y * 128 + q; r < r1; r += 32) {
        bool k[4];
        keep4(g.mask, g.ldmask, g.p, g.seed, step, g.layer, r, col >> 2, g.C, k);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col + c < g.C) {
                const int j = cg * 4 + c;
                const float y = (g.a[(size_t)r * g.lda + col + c] - s_mean[j]) * s_inv[j] * s_gamma[j] + s_beta[j];
                g.h[(size_t)r * g.ldh + col + c] = k[c] ? y * scale : 0.f;
            }
    }
}

struct BnBwdArgs {
    const float* dh; int lddh;        // [B, C] gradient w.r.t. the dropout output
    const float* a; int lda;          // [B, C] saved post-activation
    int B, C, S;
    const float* mean; const float* invstd; const float* gamma;
    const unsigned char* mask; int ldmask; float p;
    unsigned long long seed; unsigned int step, layer;
    const unsigned int* step_dev;
    float* dgamma; float* dbeta; long long slab_stride;   // slab s at + s*slab_stride: partial column sums
    int act; float slope;             // activation in front of the BatchNorm: act'(a) = a > 0 ? 1 : slope
    float* dz; int lddz;              // [B, C] gradient w.r.t. the Linear output
};

// grid (ceil(C/32), S): dbeta_s[c] = sum_rows dy, dgamma_s[c] = sum_rows dy*xhat over slab s
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(BnBwdArgs g) {
    __shared__ float p0[32][33], p1[32][33];
    const int cg = threadIdx.x & 7, q = threadIdx.x >> 3;
    const int col = blockIdx.x * 32 + cg * 4;
    const int rp = slab_rows(g.B, g.S);
    const int r0 = blockIdx.y * rp, r1 = min(g.B, r0 + rp);
    const float scale = g.p > 0.f ? 1.0f / (1.0f - g.p) : 1.0f;
    const unsigned int step = g.step_dev ? g.step_dev[0] : g.step;
    float mu[4], inv[4], sb[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        mu[c] = col + c < g.C ? g.mean[col + c] : 0.f;
        inv[c] = col + c < g.C ? g.invstd[col + c] : 0.f;
    }
    if (col < g.C)
        for (int r = r0 + q; r < r1; r += 32) {
            bool k[4];
            keep4(g.mask, g.ldmask, g.p, g.seed, step, g.layer, r, col >> 2, g.C, k);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (col + c < g.C) {
                    const float dy = k[c] ? g.dh[(size_t)r * g.lddh + col + c] * scale : 0.f;
                    const float xh = (g.a[(size_t)r * g.lda + col + c] - mu[c]) * inv[c];
                    sb[c] += dy;
                    sg[c] += dy * xh;
                }
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) { p0[q][cg * 4 + c] = sb[c]; p1[q][cg * 4 + c] = sg[c]; }
    __syncthreads();
    if (threadIdx.x < 32 && blockIdx.x * 32 + threadIdx.x < g.C) {
        float tb = 0.f, tg = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) { tb += p0[i][threadIdx.x]; tg += p1[i][threadIdx.x]; }
        const int c = blockIdx.x * 32 + threadIdx.x;
        g.dbeta[(size_t)blockIdx.y * g.slab_stride + c] = tb;
        g.dgamma[(size_t)blockIdx.y * g.slab_stride + c] = tg;
    }
}

// grid (ceil(C/32), ceil(B/128))
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(BnBwdArgs g) {
    __shared__ float s_mean[32], s_inv[32], s_k[32], s_db[32], s_dg[32];
    if (threadIdx.x < 32) {
        const int c = blockIdx.x * 32 + threadIdx.x;
        float db = 0.f, dg = 0.f, mean = 0.f, inv = 0.f, ga = 0.f;
        if (c < g.C) {
            for (int s = 0; s < g.S; ++s) {
                db += g.dbeta[(size_t)s * g.slab_stride + c];
                dg += g.dgamma[(size_t)s * g.slab_stride + c];
            }
            mean = g.mean[c]; inv = g.invstd[c]; ga = g.gamma[c];
        }
        s_mean[threadIdx.x] = mean; s_inv[threadIdx.x] = inv; s_k[threadIdx.x] = ga * inv;
        s_db[threadIdx.x] = db / (float)g.B; s_dg[threadIdx.x] = dg / (float)g.B;
    }
    __syncthreads();
    const int cg = threadIdx.x & 7, q = threadIdx.x >> 3;
    const int col = blockIdx.x * 32 + cg * 4;
    if (col >= g.C) return;
    const float scale = g.p > 0.f ? 1.0f / (1.0f - g.p) : 1.0f;
    const unsigned int step = g.step_dev ? g.step_dev[0] : g.step;
    const int r1 = min(g.B, (int)(blockIdx.y + 1) * 128);
    for (int r = blockIdx.y * 128 + q; r < r1; r += 32) {
        bool k[4];
        keep4(g.mask, g.ldmask, g.p, g.seed, step, g.layer, r, col >> 2, g.C, k);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col + c < g.C) {
                const int j = cg * 4 + c;
                const float av = g.a[(size_t)r * g.lda + col + c];
                const float dy = k[c] ? g.dh[(size_t)r * g.lddh + col + c] * scale : 0.f;
                const float xh = (av - s_mean[j]) * s_inv[j];
                float d = s_k[j] * (dy - s_db[j] - xh * s_dg[j]);
                if (g.act != CLAVA_ACT_NONE) d *= av > 0.f ? 1.0f : g.slope;
                g.dz[(size_t)r * g.lddz + col + c] = d;
            }
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

int pick_slabs(int B, int C) {
    const int colblocks = (C + 31) / 32;
    int s = (512 + colblocks - 1) / colblocks;
    s = min(s, BN_MAX_SLABS);
    s = min(s, max(1, B / 64));
    return max(1, s);
}

}  // namespace

extern "C" {

long long clava_bn_workspace_floats(int C) { return 2LL * BN_MAX_SLABS * (long long)C; }

int clava_bn_dropout_train_f32(const float* a, int lda, int B, int C, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, float p, const unsigned char* keep_mask,
                               int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev, unsigned int layer,
                               float* workspace,
                               float* save_mean, float* save_invstd, float* h, int ldh, void* stream) {
    CLAVA_CHECK_ARG(a && gamma && beta && workspace && save_mean && save_invstd && h && B > 0 && C > 0 && lda >= C && ldh >= C,
                    "bn_dropout_train: bad arguments");
    CLAVA_CHECK_ARG(p >= 0.f && p < 1.f && (!keep_mask || ldmask >= C) && (!running_mean == !running_var) && layer < 32,
                    "bn_dropout_train: bad dropout / running-statistics arguments");
    BnFwdArgs g;
    g.a = a; g.lda = lda; g.B = B; g.C = C; g.S = pick_slabs(B, C); g.part = workspace; g.gamma = gamma; g.beta = beta;
    g.run_mean = running_mean; g.run_var = running_var; g.momentum = momentum; g.eps = eps; g.save_mean = save_mean;
    g.save_invstd = save_invstd; g.h = h; g.ldh = ldh; g.mask = keep_mask; g.ldmask = ldmask; g.p = p; g.seed = seed; g.step = step;
    g.step_dev = step_dev; g.layer = layer;
    {
        ProfScope prof(PROF_MISC, 0.0, 8.0 * B * C, (hipStream_t)stream);
        hipLaunchKernelGGL(bn_stats_kernel, dim3((C + 31) / 32, g.S), dim3(256), 0, (hipStream_t)stream, g);
        CLAVA_CHECK_LAUNCH("clava_bn_dropout_train_f32(stats)");
    }
    {
        ProfScope prof(PROF_MISC, 0.0, 8.0 * B * C, (hipStream_t)stream);
        hipLaunchKernelGGL(bn_apply_kernel, dim3((C + 31) / 32, (B + 127) / 128), dim3(256), 0, (hipStream_t)stream, g);
        CLAVA_CHECK_LAUNCH("clava_bn_dropout_train_f32(apply)");
    }
    return CLAVA_OK;
}

int clava_bn_dropout_backward_f32(const float* dh, int lddh, const float* a, int lda, int B, int C, const float* save_mean,
                                  const float* save_invstd, const float* gamma, float p, const unsigned char* keep_mask,
                                  int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev, unsigned int layer,
                                  int act, float slope,
                                  int splits, float* dgamma_slabs, float* dbeta_slabs, long long slab_stride, float* dz,
                                  int lddz, void* stream) {
    CLAVA_CHECK_ARG(dh && a && save_mean && save_invstd && gamma && dgamma_slabs && dbeta_slabs && dz && B > 0 && C > 0 &&
                        lddh >= C && lda >= C && lddz >= C && splits >= 1 && splits <= 64,
                    "bn_dropout_backward: bad arguments");
    CLAVA_CHECK_ARG(p >= 0.f && p < 1.f && (!keep_mask || ldmask >= C) && layer < 32, "bn_dropout_backward: bad dropout arguments");
    BnBwdArgs g;
    g.dh = dh; g.lddh = lddh; g.a = a; g.lda = lda; g.B = B; g.C = C; g.S = splits; g.mean = save_mean; g.invstd = save_invstd;
    g.gamma = gamma; g.mask = keep_mask; g.ldmask = ldmask; g.p = p; g.seed = seed; g.step = step; g.step_dev = step_dev; g.layer = layer;
    g.dgamma = dgamma_slabs; g.dbeta = dbeta_slabs; g.slab_stride = slab_stride; g.act = act; g.slope = slope; g.dz = dz; g.lddz = lddz;
    {
        ProfScope prof(PROF_MISC, 0.0, 8.0 * B * C, (hipStream_t)stream);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((C + 31) / 32, splits), dim3(256), 0, (hipStream_t)stream, g);
        CLAVA_CHECK_LAUNCH("clava_bn_dropout_backward_f32(reduce)");
    }
    {
        ProfScope prof(PROF_MISC, 0.0, 12.0 * B * C, (hipStream_t)stream);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((C + 31) / 32, (B + 127) / 128), dim3(256), 0, (hipStream_t)stream, g);
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
