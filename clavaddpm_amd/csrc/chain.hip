// Layer-stack launchers (denoiser / classifier forward, classifier input gradient), the small guidance and
// time-embedding kernels, and the host-side driver of a whole reverse-diffusion chain.  Everything is enqueued
// on one stream; nothing synchronises with or copies to the host.
#include <stdlib.h>

#include <mutex>

#include "clava_hip.h"
#include "fused_narrow.h"
#include "split.h"
#include "step.h"

namespace {

// dlogits = onehot(y) - softmax(logits): one wavefront per row (C <= ~51 cluster classes, configs/*.json:11)
__global__ void __launch_bounds__(256) logsoftmax_select_grad_kernel(const float* __restrict__ logits, int ld,
                                                                     const int* __restrict__ y, float* __restrict__ d,
                                                                     int B, int C) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= B) return;
    const float* l = logits + (size_t)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, l[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sm = 0.f;
    for (int c = lane; c < C; c += 64) sm += expf(l[c] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    const float lse = mx + logf(sm);
    const int yy = y[row];
    for (int c = lane; c < C; c += 64) d[(size_t)row * C + c] = (c == yy ? 1.f : 0.f) - expf(l[c] - lse);
}

__global__ void __launch_bounds__(256) timestep_embedding_kernel(const long long* __restrict__ t, int n,
                                                                 const float* __restrict__ freqs, int half,
                                                                 float* __restrict__ emb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const int r = i / half, k = i - r * half;
    const float arg = (float)t[r] * freqs[k];
    emb[(size_t)r * 2 * half + k] = cosf(arg);
    emb[(size_t)r * 2 * half + half + k] = sinf(arg);
}

float act_slope(int act) { return act == CLAVA_ACT_LEAKY ? 0.01f : 0.f; }

int max_width_dims(const int* dims, int from, int to) {
    int w = 0;
    for (int l = from; l <= to; ++l) w = dims[l] > w ? dims[l] : w;
    return w;
}

int check_net(const clava_net* net, const char* who) {
    CLAVA_CHECK_ARG(net && net->n_layers > 0 && net->n_layers <= 64 && net->dims && net->w && net->b && net->act,
                    "%s: incomplete clava_net", who);
    for (int l = 0; l < net->n_layers; ++l) {
        CLAVA_CHECK_ARG(net->dims[l] > 0 && net->dims[l + 1] > 0 && net->w[l], "%s: layer %d is empty", who, l);
        CLAVA_CHECK_ARG(net->act[l] == CLAVA_ACT_NONE || net->act[l] == CLAVA_ACT_RELU || net->act[l] == CLAVA_ACT_LEAKY,
                        "%s: layer %d: activation %d not supported in a stack", who, l, net->act[l]);
    }
    return CLAVA_OK;
}

const float* bn_of(const float* const* arr, int l) { return arr ? arr[l] : nullptr; }

int net_forward(const clava_net* net, const float* x, int ldx, int M, const float* rowadd0, int ld_rowadd0,
                float* const* outs, hipStream_t st) {
    const float* in = x;
    int ld = ldx;
    for (int l = 0; l < net->n_layers; ++l) {
        GemmEpi e = gemm_epi_plain();
        e.bias = net->b[l];
        e.act = net->act[l];
        if (l == 0) { e.rowadd = rowadd0; e.ldr = ld_rowadd0; }
        if (l > 0) { e.a_scale = bn_of(net->bn_scale, l - 1); e.a_shift = bn_of(net->bn_shift, l - 1); }
        CLAVA_CHECK_ARG((e.a_scale == nullptr) == (e.a_shift == nullptr), "net_forward: bn_scale/bn_shift mismatch at layer %d", l);
        int rc = clava_launch_gemm_f32(in, ld, true, net->w[l], net->dims[l], true, outs[l], net->dims[l + 1], M,
                                       net->dims[l + 1], net->dims[l], e, 1, 0, st);
        if (rc) return rc;
        in = outs[l];
        ld = net->dims[l + 1];
    }
    return CLAVA_OK;
}

int net_input_grad(const clava_net* net, int M, float* const* outs, const float* dlast, float* const* grads,
                   float alpha, hipStream_t st) {
    const float* g = dlast;   // d/d(output of layer l), layer n-1 has no activation above it
    for (int l = net->n_layers - 1; l >= 0; --l) {
        GemmEpi e = gemm_epi_plain();
        if (l > 0) {          // fold the derivative of layer l-1's activation and BatchNorm into this product
            if (net->act[l - 1] != CLAVA_ACT_NONE) {
                e.mask_src = outs[l - 1];
                e.ldm = net->dims[l];
                e.mask_slope = act_slope(net->act[l - 1]);
            }
            e.col_scale = bn_of(net->bn_scale, l - 1);
        } else {
            e.alpha = alpha;
        }
        int rc = clava_launch_gemm_f32(g, net->dims[l + 1], true, net->w[l], net->dims[l], false, grads[l], net->dims[l],
                                       M, net->dims[l], net->dims[l + 1], e, 1, 0, st);
        if (rc) return rc;
        g = grads[l];
    }
    return CLAVA_OK;
}


// same gradient, written as split rows (split.h) for the f16 matrix-core input-gradient chain; pad columns get zeros
__global__ void __launch_bounds__(256) logsoftmax_select_grad_split_kernel(const float* __restrict__ logits, int ld,
                                                                           const int* __restrict__ y, char* __restrict__ d,
                                                                           long long pitch, float scale, int B, int C) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= B) return;
    const float* l = logits + (size_t)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, l[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sm = 0.f;
    for (int c = lane; c < C; c += 64) sm += expf(l[c] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    const float lse = mx + logf(sm);
    const int yy = y[row];
    const int Cp = (C + 31) & ~31;
    _Float16* out = reinterpret_cast<_Float16*>(d + (size_t)row * pitch);
    for (int c = lane; c < Cp; c += 64) {
        const float v = (c < C) ? ((c == yy ? 1.f : 0.f) - expf(l[c] - lse)) * scale : 0.f;
        const _Float16 hi = (_Float16)v;
        _Float16* p = out + (c >> 5) * 64 + (c & 31);
        p[0] = hi;
        p[32] = (_Float16)(v - (float)hi);
    }
}

inline long long split_pitch(int cols) { return 4LL * ((cols + 31) & ~31); }

int check_split_net(const clava_split_net* n, bool need_grad, const char* who) {
    CLAVA_CHECK_ARG(n && n->n_layers > 0 && n->n_layers <= 64 && n->dims && n->w && n->b && n->act && n->shift_w && n->shift_out,
                    "%s: incomplete clava_split_net", who);
    CLAVA_CHECK_ARG(!need_grad || (n->wt && n->shift_grad), "%s: the input gradient needs wt and shift_grad", who);
    for (int l = 0; l < n->n_layers; ++l) {
        CLAVA_CHECK_ARG(n->dims[l] > 0 && n->dims[l + 1] > 0 && n->w[l] && (!need_grad || n->wt[l]), "%s: layer %d is empty", who, l);
        CLAVA_CHECK_ARG(n->act[l] == CLAVA_ACT_NONE || n->act[l] == CLAVA_ACT_RELU || n->act[l] == CLAVA_ACT_LEAKY,
                        "%s: layer %d: activation %d not supported in a stack", who, l, n->act[l]);
    }
    return CLAVA_OK;
}

// forward of a split stack: outs[l] (split rows, shift_out[l]) for l < n-1, last layer to float32 `last` (ld_last)
#define CLAVA_CHECK_HIP(call, what)                                                        \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            clava_set_error("%s: %s", what, hipGetErrorString(e_));                        \
            return CLAVA_ERR_LAUNCH;                                                       \
        }                                                                                  \
    } while (0)

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

inline long long bits_pitch(int cols) { return ((cols + 31) & ~31) / 8; }
// layers whose activation derivative travels as bits (split.h): wide enough that its planes would come from memory
inline bool bits_layer(const clava_split_net* net, int l) {
    return l + 1 < net->n_layers && net->act[l] != CLAVA_ACT_NONE && net->dims[l + 1] >= 256;
}

int split_net_forward(const clava_split_net* net, const char* x, int M, const float* rowadd0, char* const* outs, float* last,
                      int ld_last, int* overflow, hipStream_t st, int l0 = 0, unsigned char* const* bits = nullptr, int l_end = -1) {
    const char* in = x;                           // l0 > 0: x = activation of layer l0 - 1
    int shift_in = l0 == 0 ? net->shift_in : net->shift_out[l0 - 1];
    if (l_end < 0) l_end = net->n_layers;         // layers [l0, l_end)
    for (int l = l0; l < l_end; ++l) {
        const bool is_last = l == net->n_layers - 1;
        SplitGemmArgs g;
        g.A = in; g.W = (const char*)net->w[l];
        g.M = M; g.N = net->dims[l + 1]; g.Kp = (net->dims[l] + 31) & ~31; g.K = net->dims[l];
        g.pitchA = split_pitch(net->dims[l]); g.pitchW = g.pitchA;
        g.bias = net->b[l]; g.bias2 = (l == 0) ? rowadd0 : nullptr; g.act = net->act[l];
        g.mask = nullptr; g.pitchMask = 0; g.mask_slope = 0.f;
        g.scale_acc = ldexpf(1.f, -(shift_in + net->shift_w[l]));
        g.alpha = 1.f;
        g.scale_out = is_last ? 1.f : ldexpf(1.f, net->shift_out[l]);
        g.out_split = is_last ? nullptr : outs[l]; g.pitchOut = split_pitch(net->dims[l + 1]);
        g.out_f32 = is_last ? last : nullptr; g.ldc = ld_last;
        g.overflow = overflow; g.tiles_m = g.tiles_n = 0;
        if (bits && bits[l]) { g.bits_out = bits[l]; g.pitchBits = bits_pitch(net->dims[l + 1]); }
        int rc = clava_launch_gemm_f16x3(g, -1, st);
        if (rc) return rc;
        in = outs[l];
        shift_in = net->shift_out[l];
    }
    return CLAVA_OK;
}

// input gradient of a split stack: dlast = split rows (shift_dlast) of d/d(last output); grads[l] (split rows,
// shift_grad[l]) = d/d(pre-activation of layer l-1) for l >= 1; gx = alpha * d/d(input), float32 [M, dims[0]]
// `fuse` (the Gaussian reverse step that consumes gx as its guidance gradient): when the last product runs on the
// memory-bound narrow kernel, the step is done by that launch (after `wait_before_last`, the event behind which the
// denoiser's eps is complete) and *fused is set
int split_net_input_grad(const clava_split_net* net, int M, char* const* outs, const char* dlast, char* const* grads, float* gx,
                         int ld_gx, float alpha, int* overflow, hipStream_t st, unsigned char* const* bits = nullptr,
                         const GaussStepArgs* fuse = nullptr, hipEvent_t wait_before_last = nullptr, bool* fused = nullptr,
                         int l_from = -1) {
    const char* gin = dlast;
    if (l_from < 0) l_from = net->n_layers - 1;   // l_from < n_layers - 1: dlast = grads[l_from + 1] of a fused tail
    int shift_g = (l_from == net->n_layers - 1) ? net->shift_dlast : net->shift_grad[l_from + 1];
    for (int l = l_from; l >= 0; --l) {
        SplitGemmArgs g;
        g.A = gin; g.W = (const char*)net->wt[l];
        g.M = M; g.N = net->dims[l]; g.Kp = (net->dims[l + 1] + 31) & ~31; g.K = net->dims[l + 1];
        g.pitchA = split_pitch(net->dims[l + 1]); g.pitchW = g.pitchA;
        g.bias = nullptr; g.bias2 = nullptr; g.act = CLAVA_ACT_NONE;
        g.mask = nullptr; g.pitchMask = 0; g.mask_slope = 0.f;
        if (l > 0 && net->act[l - 1] != CLAVA_ACT_NONE) {      // derivative of layer l-1's activation, read off its output
            g.mask_slope = act_slope(net->act[l - 1]);
            if (bits && bits[l - 1]) { g.mask_bits = bits[l - 1]; g.pitchBits = bits_pitch(net->dims[l]); }    // ... or off its bits
            else { g.mask = outs[l - 1]; g.pitchMask = split_pitch(net->dims[l]); }
        }
        g.scale_acc = ldexpf(1.f, -(shift_g + net->shift_w[l]));
        g.alpha = (l == 0) ? alpha : 1.f;
        g.scale_out = (l == 0) ? 1.f : ldexpf(1.f, net->shift_grad[l]);
        g.out_split = (l == 0) ? nullptr : grads[l]; g.pitchOut = split_pitch(net->dims[l]);
        g.out_f32 = (l == 0) ? gx : nullptr; g.ldc = ld_gx;
        g.overflow = overflow; g.tiles_m = g.tiles_n = 0;
        int rc;
        if (l == 0 && fuse && env_int("CLAVA_NARROW_SPLIT", 1) != 0 && clava_narrow_split_ok(g)) {
            if (wait_before_last) CLAVA_CHECK_HIP(hipStreamWaitEvent(st, wait_before_last, 0), "guidance: wait for the denoiser");
            rc = clava_launch_narrow_split_gauss(g, *fuse, st);
            if (fused) *fused = true;
        } else {
            rc = clava_launch_gemm_f16x3(g, -1, st);
        }
        if (rc) return rc;
        gin = grads[l];
        shift_g = (l == 0) ? 0 : net->shift_grad[l];
    }
    return CLAVA_OK;
}

// ---- pieces of a stack for the fused narrow-stack path (fused_narrow.hip): layers [l0, l1] of the forward as separate
// products (in = activation of layer l0-1, outs[l] = activation of layer l), and the input-gradient products l_hi .. l_lo
// (gin = d/d(pre-activation of layer l_hi), i.e. grads[l_hi + 1]; writes grads[l] for l = l_hi .. l_lo, all split rows)
int split_forward_range(const clava_split_net* net, int l0, int l1, const char* in, int M, char* const* outs, int* overflow,
                        hipStream_t st) {
    for (int l = l0; l <= l1; ++l) {
        SplitGemmArgs g;
        g.A = in; g.W = (const char*)net->w[l];
        g.M = M; g.N = net->dims[l + 1]; g.Kp = (net->dims[l] + 31) & ~31; g.K = net->dims[l];
        g.pitchA = split_pitch(net->dims[l]); g.pitchW = g.pitchA;
        g.bias = net->b[l]; g.bias2 = nullptr; g.act = net->act[l];
        g.mask = nullptr; g.pitchMask = 0; g.mask_slope = 0.f;
        g.scale_acc = ldexpf(1.f, -((l == 0 ? net->shift_in : net->shift_out[l - 1]) + net->shift_w[l]));
        g.alpha = 1.f;
        g.scale_out = ldexpf(1.f, net->shift_out[l]);
        g.out_split = outs[l]; g.pitchOut = split_pitch(net->dims[l + 1]);
        g.out_f32 = nullptr; g.ldc = 0;
        g.overflow = overflow; g.tiles_m = g.tiles_n = 0;
        int rc = clava_launch_gemm_f16x3(g, -1, st);
        if (rc) return rc;
        in = outs[l];
    }
    return CLAVA_OK;
}

int split_grad_range(const clava_split_net* net, int l_hi, int l_lo, int M, char* const* outs, const char* gin, char* const* grads,
                     int* overflow, hipStream_t st, unsigned char* const* bits = nullptr) {
    for (int l = l_hi; l >= l_lo; --l) {
        SplitGemmArgs g;
        g.A = gin; g.W = (const char*)net->wt[l];
        g.M = M; g.N = net->dims[l]; g.Kp = (net->dims[l + 1] + 31) & ~31; g.K = net->dims[l + 1];
        g.pitchA = split_pitch(net->dims[l + 1]); g.pitchW = g.pitchA;
        g.bias = nullptr; g.bias2 = nullptr; g.act = CLAVA_ACT_NONE;
        g.mask = nullptr; g.pitchMask = 0; g.mask_slope = 0.f;
        if (net->act[l - 1] != CLAVA_ACT_NONE) {
            g.mask_slope = act_slope(net->act[l - 1]);
            if (bits && bits[l - 1]) { g.mask_bits = bits[l - 1]; g.pitchBits = bits_pitch(net->dims[l]); }
            else { g.mask = outs[l - 1]; g.pitchMask = split_pitch(net->dims[l]); }
        }
        g.scale_acc = ldexpf(1.f, -((l == net->n_layers - 1 ? net->shift_dlast : net->shift_grad[l + 1]) + net->shift_w[l]));
        g.alpha = 1.f;
        g.scale_out = ldexpf(1.f, net->shift_grad[l]);
        g.out_split = grads[l]; g.pitchOut = split_pitch(net->dims[l]);
        g.out_f32 = nullptr; g.ldc = 0;
        g.overflow = overflow; g.tiles_m = g.tiles_n = 0;
        int rc = clava_launch_gemm_f16x3(g, -1, st);
        if (rc) return rc;
        gin = grads[l];
    }
    return CLAVA_OK;
}

FusedLayer fused_layer(const clava_split_net* net, int l, const float* bias2) {
    FusedLayer f;
    f.bias = net->bf[l];
    f.bias2 = bias2;
    f.scale_acc = ldexpf(1.f, -((l == 0 ? net->shift_in : net->shift_out[l - 1]) + net->shift_w[l]));
    f.scale_out = ldexpf(1.f, net->shift_out[l]);
    f.act = net->act[l];
    return f;
}

// Which parts of a step the fused narrow-stack kernels cover for these stacks (all decided by shapes and by the presence of
// the fused operands; CLAVA_FUSED=0 in the environment of the Python side leaves the operands out).
bool denoiser_fusable(const clava_chain* c, const clava_split_net* d) {
    const int n = d->n_layers;
    if (c->m != 0 || !d->fs_front || !d->fs_head || !d->bf || n < 3) return false;
    if (!d->bf[0] || !d->bf[1] || !d->bf[n - 1]) return false;
    return d->dims[0] <= 32 && d->dims[1] == 128 && clava_den_front_supports(d->dims[2]) && clava_den_head_supports(d->dims[n - 1]) &&
           d->dims[n] <= 32 && d->act[0] == CLAVA_ACT_NONE;
}
bool classifier_fusable(const clava_chain* c, const clava_split_net* k) {
    if (!k || c->m != 0 || !k->fs_front || !k->fs_tail || !k->fs_back || !k->bf || !k->wt || k->n_layers != 9) return false;
    const int need[] = {0, 1, 2, 7, 8};
    for (int l : need)
        if (!k->bf[l]) return false;
    const int* d = k->dims;
    return d[0] <= 32 && d[1] == 128 && d[2] == 128 && d[3] == 256 && d[7] == 256 && d[8] == 128 && d[9] <= 64 &&
           k->act[0] == CLAVA_ACT_NONE && k->act[8] == CLAVA_ACT_NONE;
}

inline long long align256(long long v) { return (v + 255) / 256 * 256; }
// rows of the denoiser output padded to whole float4s: the head's epilogue then stores 16 bytes per lane (with an odd row
// stride -- 8397 columns at the scale-sweep shape -- it fell back to 4-byte stores: 183 us against 105 for that product)
inline int model_out_ld(int D) { return (D + 3) & ~3; }

struct SplitChainWs {
    float* x_in;        // [B, D] float32 denoiser input (written by the step kernels)
    char* x_split;      // the same as split rows
    char* den[2];       // ping-pong hidden activations
    float* model_out;   // [B, model_out_ld(D)]
    char* cls_out[64];  // classifier activations, kept for the backward pass
    unsigned char* cls_bits[64];   // their derivative bits (wide layers only, else null)
    float* logits;      // [B, C]
    char* dlogits;      // split rows [B, C]
    char* cls_grad[2];
    float* gx;          // [B, n_num]
    long long total;    // bytes
};

void carve_split(const clava_chain* c, const clava_split_net* d, const clava_split_net* k, int B, char* base, SplitChainWs& w) {
    long long off = 0;
    auto take = [&](long long bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    const int D = d->dims[0];
    w.x_in = (float*)take(4LL * B * D);
    w.x_split = take(B * split_pitch(D));
    const int hw = max_width_dims(d->dims, 1, d->n_layers - 1);
    w.den[0] = take(B * split_pitch(hw));
    w.den[1] = take(B * split_pitch(hw));
    w.model_out = (float*)take(4LL * B * model_out_ld(D));
    w.logits = nullptr; w.dlogits = nullptr; w.gx = nullptr; w.cls_grad[0] = w.cls_grad[1] = nullptr;
    if (k) {
        for (int l = 0; l + 1 < k->n_layers; ++l) w.cls_out[l] = take(B * split_pitch(k->dims[l + 1]));
        const bool use_bits = env_int("CLAVA_MASK_BITS", 1) != 0;
        for (int l = 0; l < 64; ++l) w.cls_bits[l] = nullptr;
        for (int l = 0; l + 1 < k->n_layers; ++l)
            if (use_bits && bits_layer(k, l)) w.cls_bits[l] = (unsigned char*)take(B * bits_pitch(k->dims[l + 1]));
        const int C = k->dims[k->n_layers];
        w.logits = (float*)take(4LL * B * C);
        w.dlogits = take(B * split_pitch(C));
        const int gw = max_width_dims(k->dims, 1, k->n_layers - 1);
        w.cls_grad[0] = take(B * split_pitch(gw));
        w.cls_grad[1] = take(B * split_pitch(gw));
        w.gx = (float*)take(4LL * B * c->n_num);
    }
    w.total = off;
}

// Concurrency inside one chain call.  (1) The classifier-guidance half of a step (forward, log-softmax gradient,
// input gradient) does not depend on the denoiser forward: both only read the step's input, so it runs on its own
// stream.  (2) Rows are independent for the whole chain (noise is keyed by the global row), so a large batch is cut
// into row groups that run their T steps on separate streams with no synchronisation between groups (opt-in through
// CLAVA_CHAIN_GROUPS: on MI355X at 20000 rows (1) alone is +12 %, adding (2) loses 5 % to smaller tiles).  The kernel
// sequences drift out of phase and one sequence's MFMA work fills the other's prologue / epilogue-store / last-round
// bubbles.  Streams and events are created once per device and reused; the caller's stream is forked at the start
// of the call and joined at its end, so stream-ordered callers see an ordinary asynchronous call.
enum { MAX_GROUPS = 4, MAX_PRIOS = 4 };
struct StreamPool {
    hipStream_t den[MAX_GROUPS] = {}, cls[MAX_GROUPS] = {};     // den[0] is never created: group 0 runs on the caller's stream
    hipEvent_t fork[MAX_GROUPS] = {}, join[MAX_GROUPS] = {};    // per-step fork/join between a group's two streams
    hipEvent_t begin = nullptr, done[MAX_GROUPS] = {}, done_cls[MAX_GROUPS] = {};   // whole-call fork / join with the caller's stream
    int priority = 0;
    bool ok = false, tried = false;
};
// One pool per (device, stream priority of the caller).  The side streams are created with the CALLER'S priority: every
// step makes the two streams of a group wait for each other (fork: cls waits for den, join: den waits for cls), and with
// unequal priorities the waiting side can keep the hardware queue slot from the side it waits for.  That is the most
// probable cause of the stall recorded in round 1 (guidance stream at the highest priority, GPU_MAX_HW_QUEUES = 4, ~50 000
// launches and ~4000 cross-stream waits enqueued without a host synchronisation: the high-priority queue's barrier
// packets were always runnable-to-poll while the normal-priority queue that had to signal them was time-sliced out -
// priority inversion, progress collapses but the GPU stays responsive).  Equal priorities cannot invert.
// `g_pool_mutex[dev]` serialises whole chain calls of one device: the pool's streams and events are shared by every host
// thread, and two interleaved enqueues would wait on each other's re-recorded events.
StreamPool g_pool[64][MAX_PRIOS];
std::mutex g_pool_mutex[64];

StreamPool* stream_pool(int dev, hipStream_t caller) {
    int prio = 0;
    if (hipStreamGetPriority(caller, &prio) != hipSuccess) { (void)hipGetLastError(); prio = 0; }
    StreamPool* p = nullptr;
    for (int i = 0; i < MAX_PRIOS && !p; ++i)
        if (g_pool[dev][i].tried ? g_pool[dev][i].priority == prio : true) p = &g_pool[dev][i];
    if (!p) return nullptr;                   // more distinct caller priorities than slots: run on one stream
    if (!p->tried) {
        p->tried = true;
        p->priority = prio;
        bool ok = hipEventCreateWithFlags(&p->begin, hipEventDisableTiming) == hipSuccess;
        for (int g = 0; g < MAX_GROUPS && ok; ++g) {
            ok = (g == 0 || hipStreamCreateWithPriority(&p->den[g], hipStreamNonBlocking, prio) == hipSuccess) &&
                 hipStreamCreateWithPriority(&p->cls[g], hipStreamNonBlocking, prio) == hipSuccess &&
                 hipEventCreateWithFlags(&p->fork[g], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&p->join[g], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&p->done[g], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&p->done_cls[g], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) (void)hipGetLastError();
        p->ok = ok;
    }
    return p->ok ? p : nullptr;
}



int check_split_chain(const clava_chain* c, const clava_split_net* d, const clava_split_net* k, int B) {
    CLAVA_CHECK_ARG(c && B > 0, "sample_chain_f16x3: null descriptor or empty batch");
    CLAVA_CHECK_ARG(c->sched.T > 0 && c->n_steps >= 0 && c->t_start < c->sched.T && c->t_start + 1 >= c->n_steps,
                    "sample_chain_f16x3: steps t_start=%d n_steps=%d do not fit T=%d", c->t_start, c->n_steps, c->sched.T);
    int rc = check_split_net(d, false, "sample_chain_f16x3(denoiser)");
    if (rc) return rc;
    CLAVA_CHECK_ARG(d->n_layers >= 2, "sample_chain_f16x3: denoiser needs proj + head at least");
    CLAVA_CHECK_ARG(d->dims[0] == c->n_num + c->sumK && d->dims[d->n_layers] == c->n_num + c->sumK,
                    "sample_chain_f16x3: denoiser width %d/%d != n_num + sumK = %d", d->dims[0], d->dims[d->n_layers], c->n_num + c->sumK);
    CLAVA_CHECK_ARG(c->denoiser_emb, "sample_chain_f16x3: denoiser_emb table missing");
    CLAVA_CHECK_ARG((c->m == 0) == (c->sumK == 0), "sample_chain_f16x3: m / sumK mismatch");
    CLAVA_CHECK_ARG(c->m == 0 || (c->offsets && c->log_k), "sample_chain_f16x3: offsets / log_k missing");
    if (k) {
        rc = check_split_net(k, true, "sample_chain_f16x3(classifier)");
        if (rc) return rc;
        CLAVA_CHECK_ARG(k->dims[0] == c->n_num && k->shift_in == d->shift_in,
                        "sample_chain_f16x3: classifier input (width %d, shift %d) must match the denoiser's numeric input", k->dims[0], k->shift_in);
        CLAVA_CHECK_ARG(c->m == 0, "sample_chain_f16x3: guidance is Gaussian-only (conditional_sample never carries categories)");
        CLAVA_CHECK_ARG(c->classifier_emb && c->y, "sample_chain_f16x3: classifier_emb / y missing");
    }
    return CLAVA_OK;
}

inline long long align64(long long v) { return (v + 63) / 64 * 64; }

struct ChainWs {
    float* x_in;        // [B, D_in] denoiser input
    float* den[2];      // ping-pong hidden activations [B, max hidden width]
    float* model_out;   // [B, D_in]
    float* cls_out[64]; // classifier activations, all kept for the backward pass
    float* cls_grad[2]; // ping-pong gradients [B, max width]
    float* dlogits;     // [B, C]
    float* gx;          // [B, n_num] guidance gradient
    long long total;
};

int max_width(const clava_net* n, int from, int to) {
    int w = 0;
    for (int l = from; l <= to; ++l) w = n->dims[l] > w ? n->dims[l] : w;
    return w;
}

void carve(const clava_chain* c, int B, float* base, ChainWs& w) {
    long long off = 0;
    auto take = [&](long long n) { float* p = base ? base + off : nullptr; off += align64(n); return p; };
    const clava_net* d = c->denoiser;
    const int D = d->dims[0];
    w.x_in = take((long long)B * D);
    const int hw = max_width(d, 1, d->n_layers - 1);
    w.den[0] = take((long long)B * hw);
    w.den[1] = take((long long)B * hw);
    w.model_out = take((long long)B * D);
    w.dlogits = nullptr; w.gx = nullptr; w.cls_grad[0] = w.cls_grad[1] = nullptr;
    if (c->classifier) {
        const clava_net* k = c->classifier;
        for (int l = 0; l < k->n_layers; ++l) w.cls_out[l] = take((long long)B * k->dims[l + 1]);
        const int gw = max_width(k, 1, k->n_layers - 1);
        w.cls_grad[0] = take((long long)B * gw);
        w.cls_grad[1] = take((long long)B * gw);
        w.dlogits = take((long long)B * k->dims[k->n_layers]);
        w.gx = take((long long)B * c->n_num);
    }
    w.total = off;
}

int check_chain(const clava_chain* c, int B) {
    CLAVA_CHECK_ARG(c && B > 0, "sample_chain: null descriptor or empty batch");
    CLAVA_CHECK_ARG(c->sched.T > 0 && c->n_steps >= 0 && c->t_start < c->sched.T && c->t_start + 1 >= c->n_steps,
                    "sample_chain: steps t_start=%d n_steps=%d do not fit T=%d", c->t_start, c->n_steps, c->sched.T);
    int rc = check_net(c->denoiser, "sample_chain(denoiser)");
    if (rc) return rc;
    const clava_net* d = c->denoiser;
    CLAVA_CHECK_ARG(d->n_layers >= 2, "sample_chain: denoiser needs proj + head at least");
    CLAVA_CHECK_ARG(d->dims[0] == c->n_num + c->sumK && d->dims[d->n_layers] == c->n_num + c->sumK,
                    "sample_chain: denoiser width %d/%d != n_num + sumK = %d", d->dims[0], d->dims[d->n_layers], c->n_num + c->sumK);
    CLAVA_CHECK_ARG(c->denoiser_emb, "sample_chain: denoiser_emb table missing");
    CLAVA_CHECK_ARG((c->m == 0) == (c->sumK == 0), "sample_chain: m / sumK mismatch");
    CLAVA_CHECK_ARG(c->m == 0 || (c->offsets && c->log_k), "sample_chain: offsets / log_k missing");
    if (c->classifier) {
        rc = check_net(c->classifier, "sample_chain(classifier)");
        if (rc) return rc;
        CLAVA_CHECK_ARG(c->classifier->dims[0] == c->n_num, "sample_chain: classifier input width %d != n_num %d",
                        c->classifier->dims[0], c->n_num);
        CLAVA_CHECK_ARG(c->m == 0, "sample_chain: guidance is Gaussian-only (conditional_sample never carries categories)");
        CLAVA_CHECK_ARG(c->classifier_emb && c->y, "sample_chain: classifier_emb / y missing");
    }
    return CLAVA_OK;
}

}  // namespace

extern "C" {

int clava_logsoftmax_select_grad_f32(const float* logits, int ld, const int* y, float* dlogits, int B, int C,
                                     void* stream) {
    CLAVA_CHECK_ARG(logits && y && dlogits && B > 0 && C > 0 && ld >= C, "logsoftmax_select_grad: bad arguments");
    hipLaunchKernelGGL(logsoftmax_select_grad_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, y,
                       dlogits, B, C);
    CLAVA_CHECK_LAUNCH("clava_logsoftmax_select_grad_f32");
    return CLAVA_OK;
}

int clava_timestep_embedding_f32(const long long* t, int n, const float* freqs, int dim, float* emb, void* stream) {
    CLAVA_CHECK_ARG(t && freqs && emb && n > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad arguments (dim must be even)");
    const int half = dim / 2;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n * half + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, n,
                       freqs, half, emb);
    CLAVA_CHECK_LAUNCH("clava_timestep_embedding_f32");
    return CLAVA_OK;
}

int clava_net_forward_f32(const clava_net* net, const float* x, int ldx, int M, const float* rowadd0, int ld_rowadd0,
                          float* const* outs, void* stream) {
    int rc = check_net(net, "net_forward");
    if (rc) return rc;
    CLAVA_CHECK_ARG(x && outs && M > 0 && ldx >= net->dims[0], "net_forward: bad arguments");
    for (int l = 0; l < net->n_layers; ++l) CLAVA_CHECK_ARG(outs[l], "net_forward: outs[%d] is null", l);
    return net_forward(net, x, ldx, M, rowadd0, ld_rowadd0, outs, (hipStream_t)stream);
}

int clava_net_input_grad_f32(const clava_net* net, int M, float* const* outs, const float* dlast, float* const* grads,
                             float alpha, void* stream) {
    int rc = check_net(net, "net_input_grad");
    if (rc) return rc;
    CLAVA_CHECK_ARG(outs && dlast && grads && M > 0, "net_input_grad: bad arguments");
    for (int l = 0; l < net->n_layers; ++l) CLAVA_CHECK_ARG(grads[l] && outs[l], "net_input_grad: buffer %d is null", l);
    return net_input_grad(net, M, outs, dlast, grads, alpha, (hipStream_t)stream);
}

long long clava_chain_workspace_floats(const clava_chain* c, int B) {
    if (check_chain(c, B)) return -1;
    ChainWs w;
    carve(c, B, nullptr, w);
    return w.total;
}

int clava_sample_chain_f32(const clava_chain* c, int B, double* x, int* idx, float* ws, long long ws_floats,
                           void* stream) {
    int rc = check_chain(c, B);
    if (rc) return rc;
    CLAVA_CHECK_ARG(x || c->n_num == 0, "sample_chain: x is null");
    CLAVA_CHECK_ARG(idx || c->m == 0, "sample_chain: idx is null");
    ChainWs w;
    carve(c, B, ws, w);
    CLAVA_CHECK_ARG(ws && ws_floats >= w.total, "sample_chain: workspace of %lld floats needed, %lld given", w.total, ws_floats);
    hipStream_t st = (hipStream_t)stream;
    const clava_net* d = c->denoiser;
    const int D = d->dims[0], nd = d->n_layers;
    const int emb_w = d->dims[1];

    // denoiser input of the first step: float(x) | log-one-hot(idx)
    rc = clava_launch_chain_input(x, idx, c->offsets, w.x_in, B, c->n_num, c->m, c->sumK, st);
    if (rc) return rc;

    for (int s = 0; s < c->n_steps; ++s) {
        const int t = c->t_start - s;
        const unsigned step_index = (unsigned)(c->sched.T - 1 - t);
        // ---- denoiser forward: hidden activations ping-pong, only the head output is kept
        float* outs[64];
        for (int l = 0; l < nd - 1; ++l) outs[l] = w.den[l & 1];
        outs[nd - 1] = w.model_out;
        rc = net_forward(d, w.x_in, D, B, c->denoiser_emb + (size_t)t * emb_w, 0, outs, st);
        if (rc) return rc;
        // ---- guidance: grad_x log p(y | x_t)
        const float* grad = nullptr;
        if (c->classifier) {
            const clava_net* k = c->classifier;
            const int nk = k->n_layers, C = k->dims[nk];
            rc = net_forward(k, w.x_in, D, B, c->classifier_emb + (size_t)t * k->dims[1], 0, w.cls_out, st);
            if (rc) return rc;
            rc = clava_logsoftmax_select_grad_f32(w.cls_out[nk - 1], C, c->y, w.dlogits, B, C, st);
            if (rc) return rc;
            float* grads[64];
            for (int l = 1; l < nk; ++l) grads[l] = w.cls_grad[l & 1];
            grads[0] = w.gx;
            rc = net_input_grad(k, B, w.cls_out, w.dlogits, grads, c->guidance_scale, st);
            if (rc) return rc;
            grad = w.gx;
        }
        // ---- fused Gaussian step (writes float(x_next) straight into the next denoiser input)
        if (c->n_num > 0) {
            GaussStepArgs g;
            g.s = c->sched; g.B = B; g.n_num = c->n_num;
            g.eps = w.model_out; g.ld_eps = D;
            g.x = x;
            g.grad = grad; g.ld_grad = c->n_num;
            g.noise = c->noise ? c->noise + (size_t)s * B * c->n_num : nullptr;
            g.x_in = w.x_in; g.ld_xin = D;
            g.pred_xstart = nullptr;
            g.t_rows = nullptr; g.t = t;
            g.x_is_f32 = (s == 0 && c->first_step_f32) ? 1 : 0;
            g.seed = c->seed; g.row_offset = c->row_offset; g.step_index = step_index;
            rc = clava_launch_gauss_step(g, st);
            if (rc) return rc;
        }
        // ---- fused categorical step (writes the log-one-hot of the new indices into the next denoiser input)
        if (c->m > 0) {
            MultiStepArgs q;
            q.s = c->sched; q.B = B; q.m = c->m; q.sumK = c->sumK; q.mode = MULTI_MODE_REVERSE;
            q.logits = w.model_out + c->n_num; q.ld_logits = D;
            q.idx_in = idx; q.idx_out = idx;
            q.uniforms = c->uniforms ? c->uniforms + (size_t)s * B * c->sumK : nullptr; q.ld_u = c->sumK;
            q.x_in = w.x_in + c->n_num; q.ld_xin = D;
            q.log_prob = nullptr; q.ld_lp = 0;
            q.offsets = c->offsets; q.log_k = c->log_k;
            q.t_rows = nullptr; q.t = t;
            q.seed = c->seed; q.row_offset = c->row_offset; q.step_index = step_index;
            rc = clava_launch_multinomial_step(q, st);
            if (rc) return rc;
        }
    }
    return CLAVA_OK;
}

long long clava_chain_workspace_bytes_f16x3(const clava_chain* c, const clava_split_net* denoiser,
                                            const clava_split_net* classifier, int B) {
    if (check_split_chain(c, denoiser, classifier, B)) return -1;
    SplitChainWs w;
    carve_split(c, denoiser, classifier, B, nullptr, w);
    return w.total;
}

}  // extern "C"

// The enqueue of one split-f16 chain.  `groups` / `ng` tell the caller which streams were forked off its own, so that it can
// join them back whatever this function returns.
struct ChainGroup { int r0, rows; hipStream_t den, cls; };

static int enqueue_split_chain(const clava_chain* c, const clava_split_net* d, const clava_split_net* k, int B, double* x, int* idx,
                               const SplitChainWs& w, int* overflow, hipStream_t caller, StreamPool* pool, int n_groups, bool side,
                               ChainGroup* groups, int& ng) {
    int rc = CLAVA_OK;
    const int D = d->dims[0], nd = d->n_layers, ldm = model_out_ld(D);
    const int emb_w = d->dims[1];
    const int rows_per = ((B + n_groups - 1) / n_groups + 63) / 64 * 64;     // whole 64-row blocks per group
    // tables with categorical columns and no guidance network: the first layer gathers m weight columns per row
    // (cat_input.hip) and the [B, sum K] log-one-hot image is never written; CLAVA_CAT_GATHER=0 keeps the dense product
    const bool cat_gather = c->m > 0 && !k && d->cat_wt && d->cat_bias && (c->n_num == 0 || d->num_wt) && d->dims[1] % 128 == 0 &&
                            nd >= 2 && env_int("CLAVA_CAT_GATHER", 1) != 0;
    // fused narrow-stack kernels, part by part (CLAVA_FUSED_PARTS: 1 classifier front + back, 2 classifier tail, 8 denoiser
    // front, 16 denoiser head + Gaussian step; the front's kernel does not store the two activations the per-layer
    // input-gradient products would read, so front and back go together).  Default: the classifier's three - the guidance
    // stream is the step's critical path and they take 8 launches off it; the denoiser's two measure slower than the
    // launches they replace.
    const int parts = env_int("CLAVA_FUSED_PARTS", 7);
    const bool den_fusable = !cat_gather && c->n_num > 0 && denoiser_fusable(c, d);
    const bool den_front = den_fusable && (parts & 8), den_head = den_fusable && (parts & 16);
    const bool cls_fusable = classifier_fusable(c, k);
    const bool cls_ends = cls_fusable && (parts & 1), cls_tail = cls_fusable && (parts & 2);
    if (n_groups > 1) CLAVA_CHECK_HIP(hipEventRecord(pool->begin, caller), "sample_chain_f16x3: fork");
    typedef ChainGroup Group;
    for (int g = 0; g < n_groups; ++g) {
        const int r0 = g * rows_per;
        if (r0 >= B) break;
        Group& G = groups[ng++];
        G.r0 = r0; G.rows = (B - r0 < rows_per) ? B - r0 : rows_per;
        G.den = (g == 0) ? caller : pool->den[g];
        G.cls = side ? pool->cls[g] : G.den;
        if (g > 0) CLAVA_CHECK_HIP(hipStreamWaitEvent(G.den, pool->begin, 0), "sample_chain_f16x3: fork");
    }

    auto rowp = [](char* base, long long pitch, int r0) { return base ? base + pitch * r0 : nullptr; };
    for (int gi = 0; gi < ng; ++gi) {
        const Group& G = groups[gi];
        if (cat_gather) {                          // numeric columns only, rows n_num floats apart
            if (c->n_num > 0)
                rc = clava_launch_chain_input(x + (size_t)G.r0 * c->n_num, nullptr, c->offsets, w.x_in + (size_t)G.r0 * c->n_num, G.rows,
                                              c->n_num, 0, 0, G.den);
        } else {
            rc = clava_launch_chain_input(x ? x + (size_t)G.r0 * c->n_num : nullptr, idx ? idx + (size_t)G.r0 * c->m : nullptr, c->offsets,
                                          w.x_in + (size_t)G.r0 * D, G.rows, c->n_num, c->m, c->sumK, G.den);
        }
        if (rc) return rc;
    }

    for (int s = 0; s < c->n_steps; ++s) {
        const int t = c->t_start - s;
        const unsigned step_index = (unsigned)(c->sched.T - 1 - t);
        for (int gi = 0; gi < ng; ++gi) {
            const Group& G = groups[gi];
            const int r0 = G.r0, R = G.rows;
            float* x_in = w.x_in + (size_t)r0 * D;
            char* x_split = rowp(w.x_split, split_pitch(D), r0);
            float* model_out = w.model_out + (size_t)r0 * ldm;
            // ---- denoiser input as split rows: packed from the float32 image once; afterwards the step kernels write the
            // next input's planes themselves (pad columns keep the zeros of this first pack)
            if (s == 0 && !cat_gather) {
                rc = clava_launch_split_pack(x_in, D, R, D, d->shift_in, x_split, split_pitch(D), overflow, G.den);
                if (rc) return rc;
            }
            // Gaussian step inside the guidance's last launch (narrow_split_gauss): the guidance stream, the step's critical
            // path, then runs from step to step without a hand-over; the denoiser stream waits for it once per step
            // (eligibility of the guidance's last product for the narrow kernel, narrow.hip: <= 16 outputs, W within 64 KB of LDS)
            // (or inside the fused back kernel, which ends in the same product)
            const bool fuse_gauss = k && G.cls != G.den && c->n_num > 0 && c->m == 0 && !den_head && !cat_gather &&
                                    k->dims[0] == c->n_num && env_int("CLAVA_FUSE_GAUSS", 1) != 0 &&
                                    (cls_ends || (c->n_num <= 16 && (long long)c->n_num * ((k->dims[1] + 31) & ~31) <= 16384 &&
                                                  env_int("CLAVA_NARROW_SPLIT", 1) != 0));
            if (G.cls != G.den && (s == 0 || !fuse_gauss)) {
                CLAVA_CHECK_HIP(hipEventRecord(pool->fork[gi], G.den), "sample_chain_f16x3: step fork");
                CLAVA_CHECK_HIP(hipStreamWaitEvent(G.cls, pool->fork[gi], 0), "sample_chain_f16x3: step fork");
            }
            // ---- denoiser forward
            const int hw = max_width_dims(d->dims, 1, nd - 1);
            char* outs[64];
            for (int l = 0; l < nd - 1; ++l) outs[l] = rowp(w.den[l & 1], split_pitch(hw), r0);
            outs[nd - 1] = nullptr;
            if (den_front) {
                // proj + first hidden block in one kernel; the head may run fused with the Gaussian step below
                FusedFrontArgs fa;
                fa.B = R; fa.x = x_split; fa.pitch_x = split_pitch(D);
                fa.stream = (const char*)d->fs_front;
                fa.l[0] = fused_layer(d, 0, c->denoiser_emb + (size_t)t * emb_w);
                fa.l[1] = fused_layer(d, 1, nullptr);
                fa.nt_out = d->dims[2] / 32;
                fa.out = outs[1]; fa.pitch_out = split_pitch(d->dims[2]);
                fa.overflow = overflow;
                rc = clava_launch_den_front(fa, D, G.den);
                if (rc) return rc;
                rc = den_head ? split_forward_range(d, 2, nd - 2, outs[1], R, outs, overflow, G.den)
                              : split_net_forward(d, outs[1], R, nullptr, outs, model_out, ldm, overflow, G.den, 2);
                if (rc) return rc;
            } else if (cat_gather) {
                CatInputArgs ca = {};
                ca.xnum = w.x_in + (size_t)r0 * c->n_num; ca.ld_x = c->n_num;
                ca.idx = idx + (size_t)r0 * c->m;
                ca.B = R; ca.n_num = c->n_num; ca.m = c->m; ca.d1 = d->dims[1];
                ca.offsets = c->offsets;
                ca.num_wt = d->num_wt; ca.cat_wt = d->cat_wt; ca.cat_bias = d->cat_bias;
                ca.emb = c->denoiser_emb + (size_t)t * emb_w;
                ca.log_tiny = d->log_tiny; ca.act = d->act[0];
                ca.out = outs[0]; ca.pitch_out = split_pitch(d->dims[1]);
                ca.scale_out = ldexpf(1.f, d->shift_out[0]);
                ca.overflow = overflow;
                rc = clava_launch_cat_input(ca, G.den);
                if (rc) return rc;
                rc = split_net_forward(d, outs[0], R, nullptr, outs, model_out, ldm, overflow, G.den, 1);
                if (rc) return rc;
            } else {                               // (all layers but the head when the head runs fused with the step)
                rc = split_net_forward(d, x_split, R, c->denoiser_emb + (size_t)t * emb_w, outs, model_out, ldm, overflow, G.den, 0,
                                       nullptr, den_head ? nd - 1 : -1);
                if (rc) return rc;
            }
            // ---- the fused Gaussian step's arguments (it writes float(x_next) straight into the next denoiser input)
            GaussStepArgs gs;
            bool gauss_done = false;
            if (c->n_num > 0) {
                gs.s = c->sched; gs.B = R; gs.n_num = c->n_num;
                gs.eps = model_out; gs.ld_eps = ldm;
                gs.x = x + (size_t)r0 * c->n_num;
                gs.grad = k ? w.gx + (size_t)r0 * c->n_num : nullptr; gs.ld_grad = c->n_num;
                gs.noise = c->noise ? c->noise + ((size_t)s * B + r0) * c->n_num : nullptr;
                gs.x_in = nullptr; gs.ld_xin = 0;
                if (cat_gather) {                  // the gather layer reads the float32 numeric columns themselves
                    gs.x_in = w.x_in + (size_t)r0 * c->n_num; gs.ld_xin = c->n_num;
                } else {
                    gs.xs.base = x_split; gs.xs.pitch = split_pitch(D); gs.xs.scale = ldexpf(1.f, d->shift_in); gs.xs.col0 = 0;
                    gs.xs.overflow = overflow;
                }
                gs.pred_xstart = nullptr;
                gs.t_rows = nullptr; gs.t = t;
                gs.x_is_f32 = (s == 0 && c->first_step_f32) ? 1 : 0;
                gs.seed = c->seed; gs.row_offset = c->row_offset + (unsigned long long)r0; gs.step_index = step_index;
            }
            if (fuse_gauss)                        // the denoiser's eps is complete behind this event
                CLAVA_CHECK_HIP(hipEventRecord(pool->fork[gi], G.den), "sample_chain_f16x3: denoiser done");
            // ---- guidance: grad_x log p(y | x_t), on the group's second stream
            if (k) {
                const int nk = k->n_layers, C = k->dims[nk];
                const hipStream_t st = G.cls;
                char* cls_out[64];
                for (int l = 0; l + 1 < nk; ++l) cls_out[l] = rowp(w.cls_out[l], split_pitch(k->dims[l + 1]), r0);
                cls_out[nk - 1] = nullptr;
                float* logits = w.logits + (size_t)r0 * C;
                char* dlogits = rowp(w.dlogits, split_pitch(C), r0);
                float* gx = w.gx + (size_t)r0 * c->n_num;
                const int gw = max_width_dims(k->dims, 1, nk - 1);
                char* grads[64];
                for (int l = 1; l < nk; ++l) grads[l] = rowp(w.cls_grad[l & 1], split_pitch(gw), r0);
                grads[0] = nullptr;
                // front (3 layers) -> 4 wide products -> tail (2 layers, log-softmax gradient, 2 input-gradient products)
                // -> 4 wide input-gradient products -> back (3 input-gradient products): 11 launches instead of 19 with
                // every part fused
                const float* emb = c->classifier_emb + (size_t)t * k->dims[1];
                unsigned char* cls_bits[64] = {};
                for (int l = 0; l + 1 < nk; ++l)
                    cls_bits[l] = w.cls_bits[l] ? w.cls_bits[l] + (size_t)r0 * bits_pitch(k->dims[l + 1]) : nullptr;
                if (cls_ends) cls_bits[2] = nullptr;   // the fused front stores planes only
                // ---- forward up to the tail (or to the logits)
                const int fwd_end = cls_tail ? 7 : nk;
                if (cls_ends) {
                    FusedFrontArgs fa;
                    fa.B = R; fa.x = x_split; fa.pitch_x = split_pitch(D);
                    fa.stream = (const char*)k->fs_front;
                    fa.l[0] = fused_layer(k, 0, emb); fa.l[1] = fused_layer(k, 1, nullptr); fa.l[2] = fused_layer(k, 2, nullptr);
                    fa.nt_out = 8;
                    fa.out = cls_out[2]; fa.pitch_out = split_pitch(256);
                    fa.overflow = overflow;
                    fa.mask1 = reinterpret_cast<unsigned*>(cls_out[1]);       // 16 bytes a row in the unused rows of layer 1's activation
                    rc = clava_launch_cls_front(fa, c->n_num, st);
                    if (rc) return rc;
                    rc = split_net_forward(k, cls_out[2], R, nullptr, cls_out, logits, C, overflow, st, 3, cls_bits, fwd_end);
                } else {
                    rc = split_net_forward(k, x_split, R, emb, cls_out, logits, C, overflow, st, 0, cls_bits, fwd_end);
                }
                if (rc) return rc;
                // ---- the stack's end and the first input-gradient products
                const char* gin = dlogits;
                int g_from = nk - 1;
                if (cls_tail) {
                    FusedTailArgs ta;
                    ta.B = R; ta.C = C;
                    ta.a6 = cls_out[6]; ta.pitch_a6 = split_pitch(256);
                    ta.y = c->y + r0;
                    ta.l7 = fused_layer(k, 7, nullptr); ta.l8 = fused_layer(k, 8, nullptr);
                    ta.stream = (const char*)k->fs_tail;
                    ta.scale_dlast = ldexpf(1.f, k->shift_dlast);
                    ta.scale_acc_g8 = ldexpf(1.f, -(k->shift_dlast + k->shift_w[8]));
                    ta.scale_out_g8 = ldexpf(1.f, k->shift_grad[8]); ta.slope7 = act_slope(k->act[7]);
                    ta.scale_acc_g7 = ldexpf(1.f, -(k->shift_grad[8] + k->shift_w[7]));
                    ta.scale_out_g7 = ldexpf(1.f, k->shift_grad[7]); ta.slope6 = act_slope(k->act[6]);
                    ta.out = grads[7]; ta.pitch_out = split_pitch(256);
                    ta.overflow = overflow;
                    rc = clava_launch_cls_tail(ta, st);
                    if (rc) return rc;
                    gin = grads[7];
                    g_from = 6;
                } else {
                    ProfScope prof(PROF_SELECT_GRAD, 0.0, 8.0 * R * C, st);
                    hipLaunchKernelGGL(logsoftmax_select_grad_split_kernel, dim3((R + 3) / 4), dim3(256), 0, st, logits, C, c->y + r0,
                                       dlogits, split_pitch(C), ldexpf(1.f, k->shift_dlast), R, C);
                    CLAVA_CHECK_LAUNCH("clava_logsoftmax_select_grad_split");
                }
                // ---- the remaining input-gradient products
                if (cls_ends) {
                    rc = split_grad_range(k, g_from, 3, R, cls_out, gin, grads, overflow, st, cls_bits);
                    if (rc) return rc;
                    FusedBackArgs ba;
                    ba.B = R; ba.n_num = c->n_num;
                    ba.g3 = grads[3]; ba.pitch_g3 = split_pitch(256);
                    ba.mask1 = reinterpret_cast<const unsigned*>(cls_out[1]);
                    ba.stream = (const char*)k->fs_back;
                    ba.scale_acc_g2 = ldexpf(1.f, -(k->shift_grad[3] + k->shift_w[2]));
                    ba.scale_out_g2 = ldexpf(1.f, k->shift_grad[2]); ba.slope1 = act_slope(k->act[1]);
                    ba.scale_acc_g1 = ldexpf(1.f, -(k->shift_grad[2] + k->shift_w[1]));
                    ba.scale_out_g1 = ldexpf(1.f, k->shift_grad[1]);
                    ba.scale_acc_g0 = c->guidance_scale * ldexpf(1.f, -(k->shift_grad[1] + k->shift_w[0]));
                    ba.gx = gx; ba.ld_gx = c->n_num;
                    ba.overflow = overflow;
                    if (fuse_gauss) {              // the denoiser's eps is complete behind fork[gi]
                        CLAVA_CHECK_HIP(hipStreamWaitEvent(st, pool->fork[gi], 0), "guidance: wait for the denoiser");
                        ba.gauss = 1; ba.g = gs;
                        gauss_done = true;
                    }
                    rc = clava_launch_cls_back(ba, st);
                } else {
                    rc = split_net_input_grad(k, R, cls_out, gin, grads, gx, c->n_num, c->guidance_scale, overflow, st, cls_bits,
                                              fuse_gauss ? &gs : nullptr, fuse_gauss ? pool->fork[gi] : nullptr, &gauss_done, g_from);
                }
                if (rc) return rc;
                if (G.cls != G.den) {
                    CLAVA_CHECK_HIP(hipEventRecord(pool->join[gi], G.cls), "sample_chain_f16x3: step join");
                    CLAVA_CHECK_HIP(hipStreamWaitEvent(G.den, pool->join[gi], 0), "sample_chain_f16x3: step join");
                }
            }
            // ---- fused Gaussian step (unless the guidance's last launch did it)
            if (fuse_gauss && !gauss_done) {   // the step's stream order above relied on it
                clava_set_error("sample_chain_f16x3: the guidance's last product did not take the fused Gaussian step (CLAVA_FUSE_GAUSS=0 avoids it)");
                return CLAVA_ERR_ARG;
            }
            if (c->n_num > 0 && !gauss_done) {
                const GaussStepArgs& g = gs;
                if (den_head) {                // the head's eps goes from the accumulators straight into the step
                    FusedHeadArgs ha;
                    ha.stream = (const char*)d->fs_head;
                    ha.h = outs[nd - 2]; ha.pitch_h = split_pitch(d->dims[nd - 1]);
                    ha.ks = d->dims[nd - 1] / 16;
                    ha.l = fused_layer(d, nd - 1, nullptr);
                    ha.g = g;
                    ha.g.eps = nullptr; ha.g.ld_eps = 0;
                    rc = clava_launch_den_head_gauss(ha, G.den);
                } else {
                    rc = clava_launch_gauss_step(g, G.den);
                }
                if (rc) return rc;
            }
            // ---- fused categorical step (writes the log-one-hot of the new indices into the next denoiser input)
            if (c->m > 0) {
                MultiStepArgs q;
                q.s = c->sched; q.B = R; q.m = c->m; q.sumK = c->sumK; q.mode = MULTI_MODE_REVERSE;
                q.logits = model_out + c->n_num; q.ld_logits = ldm;
                q.idx_in = idx + (size_t)r0 * c->m; q.idx_out = idx + (size_t)r0 * c->m;
                q.uniforms = c->uniforms ? c->uniforms + ((size_t)s * B + r0) * c->sumK : nullptr; q.ld_u = c->sumK;
                q.x_in = nullptr; q.ld_xin = 0;
                if (!cat_gather) {                 // (the gather layer only needs the new indices)
                    q.xs.base = x_split; q.xs.pitch = split_pitch(D); q.xs.scale = ldexpf(1.f, d->shift_in); q.xs.col0 = c->n_num;
                    q.xs.overflow = overflow;
                }
                q.log_prob = nullptr; q.ld_lp = 0;
                q.offsets = c->offsets; q.log_k = c->log_k;
                q.t_rows = nullptr; q.t = t;
                q.seed = c->seed; q.row_offset = c->row_offset + (unsigned long long)r0; q.step_index = step_index;
                rc = clava_launch_multinomial_step(q, G.den);
                if (rc) return rc;
            }
        }
    }
    return CLAVA_OK;
}

extern "C" {

int clava_sample_chain_f16x3(const clava_chain* c, const clava_split_net* denoiser, const clava_split_net* classifier, int B,
                             double* x, int* idx, void* ws, long long ws_bytes, int* overflow, void* stream) {
    int rc = check_split_chain(c, denoiser, classifier, B);
    if (rc) return rc;
    CLAVA_CHECK_ARG(x || c->n_num == 0, "sample_chain_f16x3: x is null");
    CLAVA_CHECK_ARG(idx || c->m == 0, "sample_chain_f16x3: idx is null");
    CLAVA_CHECK_ARG(overflow, "sample_chain_f16x3: the overflow flag is required");
    SplitChainWs w;
    carve_split(c, denoiser, classifier, B, (char*)ws, w);
    CLAVA_CHECK_ARG(ws && ws_bytes >= w.total && ((uintptr_t)ws & 255) == 0,
                    "sample_chain_f16x3: workspace of %lld bytes (256-byte aligned) needed, %lld given", w.total, ws_bytes);
    const hipStream_t caller = (hipStream_t)stream;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    // ---- row groups and their streams (CLAVA_CHAIN_GROUPS / CLAVA_CHAIN_STREAMS override the defaults; 1 / 1 = fully serial).
    // The pool is shared by all host threads of the device: one chain is enqueued at a time.
    std::lock_guard<std::mutex> lock(g_pool_mutex[dev]);
    StreamPool* pool = stream_pool(dev, caller);
    int n_groups = env_int("CLAVA_CHAIN_GROUPS", 1);
    n_groups = n_groups < 1 ? 1 : (n_groups > MAX_GROUPS ? MAX_GROUPS : n_groups);
    const bool side = pool && classifier && env_int("CLAVA_CHAIN_STREAMS", 2) >= 2;
    if (!pool) n_groups = 1;
    ChainGroup groups[MAX_GROUPS];
    int ng = 0;
    rc = enqueue_split_chain(c, denoiser, classifier, B, x, idx, w, overflow, caller, pool, n_groups, side, groups, ng);
    // ---- join every forked stream back into the caller's, on success AND on failure: an error between a step's fork and its
    // join must not leave a side stream running behind the caller's back (the caller frees the workspace next)
    int rc_join = CLAVA_OK;
    for (int gi = 0; gi < ng; ++gi) {
        const ChainGroup& G = groups[gi];
        if (G.den != caller) {
            if (hipEventRecord(pool->done[gi], G.den) != hipSuccess || hipStreamWaitEvent(caller, pool->done[gi], 0) != hipSuccess) rc_join = CLAVA_ERR_LAUNCH;
        }
        if (G.cls != G.den && (rc != CLAVA_OK || G.den != caller)) {
            if (hipEventRecord(pool->done_cls[gi], G.cls) != hipSuccess || hipStreamWaitEvent(caller, pool->done_cls[gi], 0) != hipSuccess) rc_join = CLAVA_ERR_LAUNCH;
        }
    }
    if (rc == CLAVA_OK && rc_join != CLAVA_OK) {
        (void)hipGetLastError();
        clava_set_error("sample_chain_f16x3: joining the side streams failed");
        return rc_join;
    }
    return rc;
}

}  // extern "C"
