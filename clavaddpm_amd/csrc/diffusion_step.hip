// Fused per-step kernels of the reverse (and forward) diffusion.
//
// gauss_step_kernel       = gaussian_p_mean_variance + condition_mean + gaussian_p_sample
//                           (tab_ddpm/gaussian_multinomial_diffsuion.py:200-237, 315-328, 354-383; ~25 ATen launches
//                           and a device->host sync per step in the reference) in one pass over [B, n_num].
// multinomial_step_kernel = predict_start + q_posterior + log_sample_categorical (+ index_to_log_onehot)
//                           (:414-455, 506-516; tab_ddpm/utils.py:127-134), and the same machinery for the initial
//                           draw (:1033-1035) and the forward q_sample (:518-523).  One wavefront owns one
//                           (row, feature) block: lane-strided coalesced reads of the block's logits, wavefront
//                           reductions for the two log-sum-exps and the Gumbel arg-max; per-feature offsets and
//                           log K are staged in LDS.  The chain state is the category INDEX per feature, not the
//                           [B, sum K] log-one-hot of the reference.
//
// Both are HBM-bound elementwise/reduction kernels.  Arithmetic follows the reference operation by operation,
// including its dtype promotions (float64 chain state); the file is compiled with
// -ffp-contract=off so that a*b + c*d rounds like the separate ATen ops do.
#include <cstdlib>
#include <cstring>
#include "rng.h"
#include "step.h"
#include "wave.h"

namespace {

constexpr float LOG_TINY = -69.07755279541016f;   // log(float32(1e-30)): value of log(clamp(onehot, 1e-30)) off the hot class

// ------------------------------------------------------------------------------------------------ Gaussian
__global__ void __launch_bounds__(256) gauss_step_kernel(GaussStepArgs a) {
    const long long n = (long long)a.B * a.n_num;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / a.n_num), col = (int)(i - (long long)row * a.n_num);
        gauss_step_element(a, row, col, a.eps[(size_t)row * a.ld_eps + col]);      // step.h
    }
}

// ------------------------------------------------------------------------------------------------ multinomial

// One (row, feature) block per group of G lanes.  G = 64: one wavefront per block, classes lane-strided, the block's
// size, offset and hot class in scalar registers.  G = 8 / 16 / 32 (features of up to 8 / 16 / 32 classes): 8 / 4 / 2
// blocks of one row per wavefront, one class per lane; the kernel orders the features by width in LDS.  Three streaming passes over the block's logits (maximum, softmax sum,
// draw; the second and third hit L1/L2): HBM sees each logit once.
//
// The posterior's normaliser (sliced_logsumexp, utils.py:151-166) is a per-block constant that cannot move the arg-max;
// it is only evaluated when the caller asks for log p(x_{t-1} | x_t).
//
// In-kernel noise: one Philox call yields the uniforms of FOUR classes of a lane (classes c, c+G, c+2G, c+3G), keyed by
// (seed, global row * m + feature, step, stream, slot = 64 * (c / 256) + (c & 63)) for G = 64: a pure function of
// (row, feature, class), whatever the grid or GPU count.
template <int G, int MODE>
__device__ __forceinline__ void multinomial_block(const MultiStepArgs& a, const int* s_off, const float* s_logk,
                                                  const float* s_side, bool staged, int row, int f, bool active) {
    constexpr int NW = (G == 64) ? 4 : 1;
    const int lane = threadIdx.x & (G - 1);
    const unsigned stream_id = (MODE == MULTI_MODE_INIT) ? clava_rng::STREAM_INIT_CAT : clava_rng::STREAM_CAT;
    const bool side_staged = staged && !a.t_rows;
    if (!active) return;              // whole G-lane groups leave together; the reductions below never cross a group
    {
        int o = staged ? s_off[f] : a.offsets[f];
        int K = (staged ? s_off[f + 1] : a.offsets[f + 1]) - o;
        int hot = (MODE == MULTI_MODE_INIT) ? -1 : a.idx_in[(size_t)row * a.m + f];
        if (G == 64) {
            o = __builtin_amdgcn_readfirstlane(o);
            K = __builtin_amdgcn_readfirstlane(K);
            hot = __builtin_amdgcn_readfirstlane(hot);
        }
        const int t = a.t_rows ? (int)a.t_rows[row] : a.t;
        const float* lg = (MODE == MULTI_MODE_REVERSE) ? a.logits + (size_t)row * a.ld_logits + o : nullptr;
        float lca_p = 0.f, uni_p = 0.f, side_hot = 0.f, side_cold = 0.f;
        if (MODE != MULTI_MODE_INIT) {
            // the x_t-side factor takes two values per block: q_pred_one_timestep of the hot / any other class
            // (:391-401); in QSAMPLE mode the whole log-probability does (q_pred at t, :403-412)
            const float logk = staged ? s_logk[f] : a.log_k[f];
            const int tp = (MODE == MULTI_MODE_QSAMPLE) ? t : max(t - 1, 0);
            lca_p = a.s.log_cumprod_alpha[tp];
            uni_p = a.s.log_1_min_cumprod_alpha[tp] - logk;
            if (side_staged) {
                side_hot = s_side[2 * f];
                side_cold = s_side[2 * f + 1];
            } else {
                const float la = (MODE == MULTI_MODE_QSAMPLE) ? lca_p : a.s.log_alpha[t];
                const float un = (MODE == MULTI_MODE_QSAMPLE) ? uni_p : a.s.log_1_min_alpha[t] - logk;
                side_hot = log_add_exp(0.f + la, un);
                side_cold = log_add_exp(LOG_TINY + la, un);
            }
        }
        const unsigned long long pair_id = (a.row_offset + (unsigned long long)row) * (unsigned long long)a.m + f;
        const unsigned step_key = a.step_index + (a.step_counter ? *a.step_counter : 0u);

        float mx = 0.f, lsum = 0.f, lse = 0.f;
        // un-normalised log posterior of class c with logit v (REVERSE), the q_pred log-probability (QSAMPLE), or 0 (INIT)
        auto unnormed = [&](float v, int c) {
            if (MODE == MULTI_MODE_INIT) return 0.f;
            const float side = (c == hot) ? side_hot : side_cold;
            if (MODE == MULTI_MODE_QSAMPLE) return side;
            const float lp = (v - mx) - lsum;                                      // predict_start (:414-424)
            float ev = lp;
            if (t != 0) ev = log_add_exp(lp + lca_p, uni_p);                        // q_pred at t-1 (:436-443)
            return ev + side;                                                      // + q_pred_one_timestep (:449)
        };
        // Gumbel arg-max (log_sample_categorical, :506-516); ties resolve to the lowest class like torch.argmax
        float best = -INFINITY;
        int best_c = 0;                         // all scores NaN or -inf: class 0, like torch.argmax (and always inside the block)
        auto visit = [&](float v, int c, unsigned word) {
            const float lp = unnormed(v, c) - lse;
            if (a.log_prob) a.log_prob[(size_t)row * a.ld_lp + o + c] = lp;
            const float u = a.uniforms ? a.uniforms[(size_t)row * a.ld_u + o + c] : clava_rng::to_uniform(word);
            const float g = -log_pos(-log_pos(u + 1e-30f) + 1e-30f);
            const float sc = g + lp;
            if (sc > best) { best = sc; best_c = c; }
        };
        auto draws = [&](int c0) {
            if (a.uniforms) return clava_rng::u4{0u, 0u, 0u, 0u};
            return clava_rng::philox4x32_10(a.seed, pair_id, step_key, stream_id | ((unsigned)((c0 >> 8) * 64 + (c0 & 63)) << 3));
        };
        // NW classes per lane at a time: their loads are issued together, and a block of up to NW * G classes (256 for a
        // wavefront) stays in registers through all passes
        auto load4 = [&](int c0, float (&v)[NW]) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int c = c0 + w * G;
                v[w] = (MODE == MULTI_MODE_REVERSE && c < K) ? lg[c] : -INFINITY;
            }
        };
        auto draw4 = [&](int c0, const float (&v)[NW]) {
            const clava_rng::u4 r = draws(c0);
            const unsigned words[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int c = c0 + w * G;
                if (w == 0 || w * G < K) {          // wave-uniform for G = 64
                    if (c < K) visit(v[w], c, words[w]);
                }
            }
        };
        if (K <= NW * G) {
            float v[NW];
            load4(lane, v);
            if (MODE == MULTI_MODE_REVERSE) {
                float m1 = v[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) m1 = fmaxf(m1, v[w]);
                mx = group_max<G>(m1);
                float sm = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    if (w == 0 || w * G < K) sm += exp_nonpos(v[w] - mx);      // exp(-inf) = 0 for the lanes beyond K
                lsum = log_pos(group_sum<G>(sm));
                if (a.log_prob) {
                    float un[NW], m2 = -INFINITY;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const int c = lane + w * G;
                        un[w] = (c < K) ? unnormed(v[w], c) : -INFINITY;
                        m2 = fmaxf(m2, un[w]);
                    }
                    m2 = group_max<G>(m2);
                    float sm2 = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sm2 += exp_nonpos(un[w] - m2);
                    lse = m2 + log_pos(group_sum<G>(sm2));
                }
            }
            draw4(lane, v);
        } else {
            float v[NW];
            if (MODE == MULTI_MODE_REVERSE) {
                float m1 = -INFINITY;
                for (int c0 = lane; c0 < K; c0 += NW * G) {
                    load4(c0, v);
#pragma unroll
                    for (int w = 0; w < NW; ++w) m1 = fmaxf(m1, v[w]);
                }
                mx = group_max<G>(m1);
                float sm = 0.f;
                for (int c0 = lane; c0 < K; c0 += NW * G) {
                    load4(c0, v);
#pragma unroll
                    for (int w = 0; w < NW; ++w) sm += exp_nonpos(v[w] - mx);
                }
                lsum = log_pos(group_sum<G>(sm));
                if (a.log_prob) {
                    float m2 = -INFINITY;
                    for (int c = lane; c < K; c += G) m2 = fmaxf(m2, unnormed(lg[c], c));
                    m2 = group_max<G>(m2);
                    float sm2 = 0.f;
                    for (int c = lane; c < K; c += G) sm2 += exp_nonpos(unnormed(lg[c], c) - m2);
                    lse = m2 + log_pos(group_sum<G>(sm2));
                }
            }
            for (int c0 = lane; c0 < K; c0 += NW * G) {
                load4(c0, v);
                draw4(c0, v);
            }
        }
        const float top = group_max<G>(best);
        best_c = group_min<G>(best == top ? best_c : 0x7fffffff);
        if (lane == 0) a.idx_out[(size_t)row * a.m + f] = best_c;
        if (a.x_in) {                              // index_to_log_onehot for the next denoiser input (utils.py:127-134)
            float* xo = a.x_in + (size_t)row * a.ld_xin + o;
            for (int c = lane; c < K; c += G) xo[c] = (c == best_c) ? 0.f : LOG_TINY;
        }
        if (a.xs.base)
            for (int c = lane; c < K; c += G) store_split(a.xs, row, a.xs.col0 + o + c, (c == best_c) ? 0.f : LOG_TINY);
    }
}

// ---- inverse-CDF form of the reverse step's categorical draw (production path: in-kernel draws only) -------------------
// The Gumbel arg-max above spends 2 exp + 3 log and a quarter of a Philox call PER CLASS.  The distribution it samples,
// p(x_{t-1} = c | x_t, x0_hat) (q_posterior of the predicted x_0, `:426-455`), is known in probability space:
//     w_c = (abar_{t-1} * softmax(logits)_c + (1 - abar_{t-1}) / K) * s_c,      s_c = s_hot if c == x_t else s_cold,
// (s = exp of the two-valued q_pred_one_timestep factor; at t = 0 the mixing is skipped: w_c = softmax_c * s_c, `:437-441`),
// so one uniform u per (row, feature) block and the first class whose running sum reaches u * sum(w) draw from the SAME
// distribution with one exp per class, a DPP prefix sum per 64 classes and one Philox call per block.  The draws differ
// from the Gumbel path's (another use of the random words), like the closed-form q-sample of the training draws; parity
// tests with injected uniforms always take the Gumbel path, and tests/test_gpu_icdf.py holds this path to the exact
// posterior probabilities.
template <int CTRL> __device__ __forceinline__ float dpp_shr_f(float v) {       // row_shr:N, zero fill at the row start
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// inclusive prefix sum over each group of G lanes (lane order)
template <int G> __device__ __forceinline__ float group_scan(float v) {
    constexpr int R = G < 16 ? G : 16;
    const int l = threadIdx.x & (R - 1);
    float t;
    t = dpp_shr_f<0x111>(v); if (l >= 1) v += t;
    t = dpp_shr_f<0x112>(v); if (l >= 2) v += t;
    t = dpp_shr_f<0x114>(v); if (l >= 4) v += t;
    if (R == 16) { t = dpp_shr_f<0x118>(v); if (l >= 8) v += t; }
    if (G == 32) {                      // rows 0,1 / 2,3 of the wavefront form a group: the odd row adds the even row's total
        const float lo = __shfl(v, (threadIdx.x & 32) | 15, 64);
        if (threadIdx.x & 16) v += lo;
    }
    if (G == 64) {
        const int i = __float_as_int(v);
        const float t0 = __int_as_float(__builtin_amdgcn_readlane(i, 15)), t1 = __int_as_float(__builtin_amdgcn_readlane(i, 31)),
                    t2 = __int_as_float(__builtin_amdgcn_readlane(i, 47));
        const int row = (threadIdx.x >> 4) & 3;
        v += (row == 0) ? 0.f : (row == 1) ? t0 : (row == 2) ? (t0 + t1) : ((t0 + t1) + t2);
    }
    return v;
}

// The CDF runs over the classes in LANE-MAJOR order: lane l owns classes l, l + G, l + 2G, ... (Q of them: 1 for the narrow
// lane groups, 2 .. 16 for a 64-lane block of up to 1024 classes; wider blocks go through in 1024-class chunks), i.e. the
// loads stay lane-strided and coalesced, the running sum inside a lane's run needs no cross-lane work, and ONE prefix sum
// over the lane totals orders the runs.  Any fixed class order is a valid inverse CDF; this one costs a scan and three
// reductions per block instead of per 64 classes.
struct IcdfBlock {
    float A, U, s_hot, s_cold;
    int K, hot;
    const float* lg;
};

// probability-space constants of a block (float32 exp of the reference's float32 log tables)
__device__ __forceinline__ void icdf_constants(const ScheduleDev& s, int t, float logk, float& A, float& U, float& s_hot, float& s_cold) {
    const int tp = max(t - 1, 0);
    A = (t != 0) ? expf(s.log_cumprod_alpha[tp]) : 1.f;
    U = (t != 0) ? expf(s.log_1_min_cumprod_alpha[tp] - logk) : 0.f;
    const float la = s.log_alpha[t], un = s.log_1_min_alpha[t] - logk;
    s_hot = expf(log_add_exp(0.f + la, un));
    s_cold = expf(log_add_exp(LOG_TINY + la, un));
}

template <int G, int Q>
__device__ __forceinline__ int icdf_resident(const IcdfBlock& b, float u, int lane) {
    float v[Q];
    float m1 = -INFINITY;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int c = lane + j * G;
        v[j] = (c < b.K) ? b.lg[c] : -INFINITY;
        m1 = fmaxf(m1, v[j]);
    }
    const float mx = group_max<G>(m1);
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < Q; ++j) { v[j] = exp_nonpos(v[j] - mx); sm += v[j]; }       // exp(-inf) = 0 beyond K
    const float rS = b.A * __builtin_amdgcn_rcpf(group_sum<G>(sm));
    float run = 0.f;
#pragma unroll
    for (int j = 0; j < Q; ++j) {                                                    // v[j] <- running sum of the lane's weights
        const int c = lane + j * G;
        const float w = (c < b.K) ? (v[j] * rS + b.U) * (c == b.hot ? b.s_hot : b.s_cold) : 0.f;
        run += w;
        v[j] = run;
    }
    const float incl = group_scan<G>(run);
    const float before = incl - run;
    // total from the scan itself (last lane's inclusive sum), so that target <= the largest running sum up to rounding
    const float total = (G == 64) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63))
                                  : __shfl(incl, (threadIdx.x & 63 & ~(G - 1)) | (G - 1), 64);
    const float target = u * total;
    int key = 0x7fffffff;                                                            // position in CDF order: lane * Q + j
#pragma unroll
    for (int j = Q - 1; j >= 0; --j)
        if (lane + j * G < b.K && before + v[j] >= target) key = lane * Q + j;
    key = group_min<G>(key);
    // no running sum reached the target: u * total beyond the last sum by rounding -> the last class; NaN weights -> class 0
    return key == 0x7fffffff ? (total == total ? b.K - 1 : 0) : (key / Q) + (key % Q) * G;
}

// One block per wavefront (G = 64), up to 64 Q classes.  Three more things make the per-class work ~10 instructions:
//  * the hot class's weight is split into its cold-valued part, which sits at the class's place in the order, and the
//    remainder delta = (p_hot * A + U) * (s_hot - s_cold) >= 0, a piece of its own at the END of the order: the per-class
//    loop then knows one side factor only, and u * total beyond the cold total simply means "the hot class";
//  * the cold weights are affine in the softmax numerators e_c: the running sum at (lane, slot j) is
//    kE * (sum of e up to there) + kU * (number of classes up to there), so ONE prefix sum over the lanes' e totals gives the
//    softmax sum, the total and every lane's start, and the class count needs no scan (lane-major order, ragged last slot);
//  * the lane that holds the draw comes from a ballot over the lanes' end sums, the slot inside it from a count of the
//    running sums below the target, read from that lane.
// exp = one hardware exp2 of fma(x, log2 e, -max log2 e): 1e-6 relative at worst on classes within e^-20 of the largest,
// a sampler's weights, not a value any parity test pins.
template <int Q>
__device__ __forceinline__ int icdf_wave(const IcdfBlock& b, float u, int lane) {
    const float LOG2E = 1.44269504088896341f;
    const int nfull = b.K >> 6, rem = b.K & 63;                 // scalar: whole slots, lanes of the ragged slot
    float e[Q];
    float m1 = -INFINITY;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int c = lane + j * 64;
        e[j] = (c < b.K) ? b.lg[c] : -INFINITY;
        m1 = fmaxf(m1, e[j]);
    }
    const float ml = group_max<64>(m1) * LOG2E;
    const bool hot_ok = b.hot >= 0 && b.hot < b.K;
    const int hot_slot = b.hot >> 6;                             // (scalar)
    float pe = 0.f, eh = 0.f;
#pragma unroll
    for (int j = 0; j < Q; ++j) {                                // e[j] <- running sum of the lane's numerators
        const float ev = __builtin_amdgcn_exp2f(__builtin_fmaf(e[j], LOG2E, -ml));
        eh = (j == hot_slot) ? ev : eh;                          // the hot class's numerator stays in its lane (no second load)
        pe += ev;
        e[j] = pe;
    }
    const float incl = group_scan<64>(pe);
    const float S = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
    const float kE = b.A * b.s_cold * __builtin_amdgcn_rcpf(S), kU = b.U * b.s_cold;
    const float cold_total = kE * S + kU * (float)b.K;
    float delta = 0.f;
    if (hot_ok) {
        const float e_hot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(eh), b.hot & 63));
        delta = (e_hot * (b.A * __builtin_amdgcn_rcpf(S)) + b.U) * (b.s_hot - b.s_cold);
    }
    const float total = cold_total + delta;
    if (!(total == total)) return 0;                             // NaN / infinite scores: class 0 (inside the block)
    const float target = u * total;
    if (hot_ok && target > cold_total) return b.hot;
    const int before_n = lane * nfull + min(lane, rem);          // classes owned by the lanes in front
    const int nv = nfull + (lane < rem ? 1 : 0);
    const float lane_end = __builtin_fmaf(kE, incl, kU * (float)(before_n + nv));
    const unsigned long long hit = __ballot(nv > 0 && lane_end >= target);
    if (hit == 0ull) return b.K - 1;                             // u * total beyond the last sum by rounding
    const int wl = __builtin_ctzll(hit);
    // running sum at slot j = kE * (before_e + e[j]) + kU * (before_n + j + 1) = fma(kE, e[j], base + kU * (j + 1))
    const float base = __builtin_fmaf(kE, incl - pe, kU * (float)before_n);
    const float thr = target - base;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < Q; ++j) cnt += (__builtin_fmaf(kE, e[j], kU * (float)(j + 1)) < thr) ? 1 : 0;
    const int nv_w = nfull + (wl < rem ? 1 : 0);
    const int slot = min(__builtin_amdgcn_readlane(cnt, wl), nv_w - 1);
    return wl + slot * 64;
}

// the block's uniform in (0, 1]: 24 bits of one Philox word keyed by (seed, global row * m + feature, step)
__device__ __forceinline__ float icdf_uniform(const MultiStepArgs& a, int row, int f) {
    const unsigned long long pair_id = (a.row_offset + (unsigned long long)row) * (unsigned long long)a.m + f;
    const unsigned step_key = a.step_index + (a.step_counter ? *a.step_counter : 0u);
    const clava_rng::u4 r = clava_rng::philox4x32_10(a.seed, pair_id, step_key, clava_rng::STREAM_CAT | (0x1fffffu << 3));
    return ((float)(r.x >> 8) + 1.0f) * 5.9604644775390625e-8f;
}
// the drawn class and, if asked for, the log-one-hot image of the next denoiser input (utils.py:127-134)
template <int G>
__device__ __forceinline__ void icdf_write_g(const MultiStepArgs& a, int row, int f, int o, int K, int best_c, int lane) {
    if (lane == 0) a.idx_out[(size_t)row * a.m + f] = best_c;
    if (a.x_in) {
        float* xo = a.x_in + (size_t)row * a.ld_xin + o;
        for (int c = lane; c < K; c += G) xo[c] = (c == best_c) ? 0.f : LOG_TINY;
    }
    if (a.xs.base)
        for (int c = lane; c < K; c += G) store_split(a.xs, row, a.xs.col0 + o + c, (c == best_c) ? 0.f : LOG_TINY);
}

// `s_prob` (staged tables and one t per launch): the block's probability-space constants U, s_hot, s_cold per feature and A
// at [3 m], evaluated once per workgroup instead of once per block (four exponentials and two log-add-exps: more than the
// per-class work of a block of up to 64 classes)
template <int G>
__device__ __forceinline__ void multinomial_block_icdf(const MultiStepArgs& a, const int* s_off, const float* s_logk,
                                                       const float* s_prob, bool staged, int row, int f, bool active, int hot,
                                                       bool have_u = false, float u_in = 0.f) {
    const int lane = threadIdx.x & (G - 1);
    if (!active) return;
    if (G == 64) {      // one block per wavefront: row and feature in scalar registers, so that the addresses, the block's
        row = __builtin_amdgcn_readfirstlane(row);      // constants and the whole Philox call run on the scalar unit
        f = __builtin_amdgcn_readfirstlane(f);
    }
    int o = staged ? s_off[f] : a.offsets[f];
    int K = (staged ? s_off[f + 1] : a.offsets[f + 1]) - o;
    if (G == 64) {
        o = __builtin_amdgcn_readfirstlane(o);
        K = __builtin_amdgcn_readfirstlane(K);
        hot = __builtin_amdgcn_readfirstlane(hot);
    }
    const int t = a.t_rows ? (int)a.t_rows[row] : a.t;
    const float logk = staged ? s_logk[f] : a.log_k[f];
    IcdfBlock b;
    if (s_prob) {
        b.U = s_prob[3 * f]; b.s_hot = s_prob[3 * f + 1]; b.s_cold = s_prob[3 * f + 2]; b.A = s_prob[3 * a.m];
    } else {
        icdf_constants(a.s, t, logk, b.A, b.U, b.s_hot, b.s_cold);
    }
    b.K = K; b.hot = hot;
    b.lg = a.logits + (size_t)row * a.ld_logits + o;
    const float u = have_u ? u_in : icdf_uniform(a, row, f);          // the search looks for the first running sum >= u * total
    int best_c;
    if (G < 64) best_c = icdf_resident<G, 1>(b, u, lane);
    else if (K <= 64) best_c = icdf_wave<1>(b, u, lane);
    else if (K <= 128) best_c = icdf_wave<2>(b, u, lane);
    else if (K <= 256) best_c = icdf_wave<4>(b, u, lane);
    else if (K <= 512) best_c = icdf_wave<8>(b, u, lane);
    else if (K <= 1024) best_c = icdf_wave<16>(b, u, lane);
    else {
        // beyond 1024 classes: 1024-class chunks (16 per lane, lane-strided) streamed three times (maximum, softmax sum, search)
        constexpr int Q = 16, CH = G * Q;
        float m1 = -INFINITY;
        for (int cb = 0; cb < K; cb += CH)
#pragma unroll
            for (int j = 0; j < Q; ++j) { const int c = cb + lane + j * G; if (c < K) m1 = fmaxf(m1, b.lg[c]); }
        const float mx = group_max<G>(m1);
        float sm = 0.f;
        for (int cb = 0; cb < K; cb += CH)
#pragma unroll
            for (int j = 0; j < Q; ++j) { const int c = cb + lane + j * G; if (c < K) sm += exp_nonpos(b.lg[c] - mx); }
        const float rS = b.A * __builtin_amdgcn_rcpf(group_sum<G>(sm));
        const bool hot_ok = hot >= 0 && hot < K;
        const float e_hot = hot_ok ? exp_nonpos(b.lg[hot] - mx) : 0.f;
        const float total = b.s_cold * (b.A + (float)K * b.U) + (hot_ok ? (b.s_hot - b.s_cold) * (e_hot * rS + b.U) : 0.f);
        const float target = u * total;
        float base = 0.f;
        best_c = -1;
        for (int cb = 0; cb < K && best_c < 0; cb += CH) {       // (best_c is wave-uniform: the loop ends at the chunk that holds the draw)
            float run = 0.f, cs[Q];
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const int c = cb + lane + j * G;
                run += (c < K) ? (exp_nonpos(b.lg[c] - mx) * rS + b.U) * (c == hot ? b.s_hot : b.s_cold) : 0.f;
                cs[j] = run;
            }
            const float incl = group_scan<G>(run);
            const float before = base + (incl - run);
            int key = 0x7fffffff;
#pragma unroll
            for (int j = Q - 1; j >= 0; --j)
                if (cb + lane + j * G < K && before + cs[j] >= target) key = lane * Q + j;
            key = group_min<G>(key);
            if (key != 0x7fffffff) best_c = cb + (key / Q) + (key % Q) * G;
            base += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
        }
        if (best_c < 0) best_c = (total == total) ? K - 1 : 0;
    }
    icdf_write_g<G>(a, row, f, o, K, best_c, lane);
}

// All (row, feature) blocks of the `n_feat` features listed in `perm`, 64 / G blocks of one row per wavefront work item.
template <int G, int MODE, bool ICDF>
__device__ __forceinline__ void multinomial_bucket(const MultiStepArgs& a, const int* s_off, const float* s_logk, const float* s_side,
                                                   const int* perm, int n_feat, long long wave0, long long n_waves) {
    // (inverse CDF, one block per wavefront: a work item is FOUR consecutive blocks of a row, which share one Philox call)
    constexpr int PER = (ICDF && G == 64) ? 4 : 64 / G;
    const int W = (n_feat + PER - 1) / PER;                 // work items per row
    if (W == 0) return;
    // a stride coprime with W makes every wavefront walk through ALL the features of the bucket (their costs differ by
    // up to 16x): the few wavefronts beyond the stride sit this bucket out
    auto gcd = [](long long x, long long y) { while (y) { const long long r = x % y; x = y; y = r; } return x; };
    while (n_waves > 1 && gcd(n_waves, W) != 1) --n_waves;
    if (wave0 >= n_waves || wave0 >= (long long)a.B * W) return;
    const int sub = (threadIdx.x & 63) / G;
    int row = (int)(wave0 / W), w = (int)(wave0 - (long long)row * W);
    const int d_row = (int)(n_waves / W), d_w = (int)(n_waves - (long long)d_row * W);
    if constexpr (ICDF && G == 64) {
        // Four blocks per work item: ONE Philox call (its four words are the four blocks' uniforms, keyed by the global row
        // and the item's place among the table's wide features - a function of the table alone, so the draws still do not
        // depend on grid, row cuts or GPU count) instead of one per block: the call is ~100 scalar instructions, and the
        // scalar unit of a CU was as busy as its vector units.  x_t of the next item's four blocks is fetched one item ahead.
        auto load_hots = [&](int r, int wi, int (&h)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = a.idx_in[(size_t)r * a.m + perm[min(wi * 4 + k, n_feat - 1)]];
        };
        int hn[4];
        load_hots(row, w, hn);
        const unsigned step_key = a.step_index + (a.step_counter ? *a.step_counter : 0u);
        while (row < a.B) {
            const int row_s = __builtin_amdgcn_readfirstlane(row), w_s = __builtin_amdgcn_readfirstlane(w);
            const int h0 = __builtin_amdgcn_readfirstlane(hn[0]), h1 = __builtin_amdgcn_readfirstlane(hn[1]);
            const int h2 = __builtin_amdgcn_readfirstlane(hn[2]), h3 = __builtin_amdgcn_readfirstlane(hn[3]);
            row += d_row; w += d_w;
            if (w >= W) { w -= W; ++row; }
            if (row < a.B) load_hots(row, w, hn);
            const unsigned long long item = (a.row_offset + (unsigned long long)row_s) * (unsigned long long)W + w_s;
            const clava_rng::u4 r = clava_rng::philox4x32_10(a.seed, item, step_key, clava_rng::STREAM_CAT | (0x1ffffeu << 3));
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const int fi = w_s * 4 + k;
                if (fi >= n_feat) break;
                const unsigned word = k == 0 ? r.x : k == 1 ? r.y : k == 2 ? r.z : r.w;
                const int hot = k == 0 ? h0 : k == 1 ? h1 : k == 2 ? h2 : h3;
                const float u = ((float)(word >> 8) + 1.0f) * 5.9604644775390625e-8f;      // 24 bits -> (0, 1]
                multinomial_block_icdf<64>(a, s_off, s_logk, a.t_rows ? nullptr : s_side, true, row_s, perm[fi], true, hot, true, u);
            }
        }
        return;
    }
    if constexpr (ICDF) {
        // the block's current class (x_t) is fetched one work item ahead: with one exponential per class left, a work item
        // is a chain of dependent memory round trips (x_t, the logits), and this takes the first one off it
        int fi = w * PER + sub, f = perm[min(fi, n_feat - 1)];
        int hot_next = a.idx_in[(size_t)row * a.m + f];
        while (row < a.B) {
            const int row_c = row, f_c = f, hot = hot_next;
            const bool on = fi < n_feat;
            row += d_row; w += d_w;
            if (w >= W) { w -= W; ++row; }
            if (row < a.B) {
                fi = w * PER + sub; f = perm[min(fi, n_feat - 1)];
                hot_next = a.idx_in[(size_t)row * a.m + f];
            }
            multinomial_block_icdf<G>(a, s_off, s_logk, a.t_rows ? nullptr : s_side, true, row_c, f_c, on, hot);
        }
        return;
    }
    for (; row < a.B; row += d_row, w += d_w) {
        if (w >= W) { w -= W; ++row; if (row >= a.B) break; }
        const int fi = w * PER + sub;
        multinomial_block<G, MODE>(a, s_off, s_logk, s_side, true, row, perm[min(fi, n_feat - 1)], fi < n_feat);
    }
}

// ICDF: the inverse-CDF sampler (REVERSE only) as a kernel of its own - its register budget differs from the Gumbel form's
#ifndef CLAVA_ICDF_WAVES
#define CLAVA_ICDF_WAVES 6          // waves per SIMD the inverse-CDF instantiation is compiled for (experiments: tools/build_variant.py)
#endif
template <int MODE, bool ICDF = false>
__global__ void __launch_bounds__(256, ICDF ? CLAVA_ICDF_WAVES : 1) multinomial_step_kernel(MultiStepArgs a) {
    __shared__ int s_off[MAX_LDS_FEATURES + 1];
    __shared__ float s_logk[MAX_LDS_FEATURES];
    __shared__ float s_side[3 * MAX_LDS_FEATURES + 1];     // Gumbel: log side factors [2 m]; ICDF: U, s_hot, s_cold [3 m] and A
    __shared__ int s_perm[MAX_LDS_FEATURES];       // features ordered by lane-group width
    __shared__ int s_start[5];
    const bool staged = a.m <= MAX_LDS_FEATURES;
    const int wave_lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    if (!staged) {
        // more features than the LDS tables hold: one wavefront per (row, feature) block, tables read from memory
        const long long n_pairs = (long long)a.B * a.m;
        if (wave0 >= n_pairs) return;
        int row = (int)(wave0 / a.m), f = (int)(wave0 - (long long)row * a.m);
        const int d_row = (int)(n_waves / a.m), d_f = (int)(n_waves - (long long)d_row * a.m);
        for (; row < a.B; row += d_row, f += d_f) {
            if (f >= a.m) { f -= a.m; ++row; if (row >= a.B) break; }
            if constexpr (ICDF) multinomial_block_icdf<64>(a, s_off, s_logk, nullptr, false, row, f, true, a.idx_in[(size_t)row * a.m + f]);
            else multinomial_block<64, MODE>(a, s_off, s_logk, s_side, false, row, f, true);
        }
        return;
    }
    for (int i = threadIdx.x; i <= a.m; i += blockDim.x) s_off[i] = a.offsets[i];
    for (int i = threadIdx.x; i < a.m; i += blockDim.x) {
        const float logk = a.log_k[i];
        s_logk[i] = logk;
        if (ICDF && !a.t_rows) {
            float A, U, sh, sc;
            icdf_constants(a.s, a.t, logk, A, U, sh, sc);
            s_side[3 * i] = U; s_side[3 * i + 1] = sh; s_side[3 * i + 2] = sc;
            if (i == 0) s_side[3 * a.m] = A;
        } else if (MODE != MULTI_MODE_INIT && !a.t_rows) {     // one t for the whole launch: the side factors depend on the feature only
            const int tp = (MODE == MULTI_MODE_QSAMPLE) ? a.t : max(a.t - 1, 0);
            const float la = (MODE == MULTI_MODE_QSAMPLE) ? a.s.log_cumprod_alpha[tp] : a.s.log_alpha[a.t];
            const float un = ((MODE == MULTI_MODE_QSAMPLE) ? a.s.log_1_min_cumprod_alpha[tp] : a.s.log_1_min_alpha[a.t]) - logk;
            s_side[2 * i] = log_add_exp(0.f + la, un);
            s_side[2 * i + 1] = log_add_exp(LOG_TINY + la, un);
        }
    }
    __syncthreads();
    // features grouped by lane-group width (their order inside a group does not matter: every draw is keyed by the
    // feature, not by where it runs); a wavefront work item = one 64-lane block, or 2 / 4 / 8 narrower blocks of one row
    order_features_by_width(a.m, s_off, s_perm, s_start);
    const int st1 = s_start[1], st2 = s_start[2], st3 = s_start[3];
    const int n0 = st1, n1 = st2 - st1, n2 = st3 - st2, n3 = a.m - st3;
    // one loop per width (a single loop over mixed items would keep all four variants' registers live at once)
    multinomial_bucket<8, MODE, ICDF>(a, s_off, s_logk, s_side, s_perm, n0, wave0, n_waves);
    multinomial_bucket<16, MODE, ICDF>(a, s_off, s_logk, s_side, s_perm + st1, n1, wave0, n_waves);
    multinomial_bucket<32, MODE, ICDF>(a, s_off, s_logk, s_side, s_perm + st2, n2, wave0, n_waves);
    multinomial_bucket<64, MODE, ICDF>(a, s_off, s_logk, s_side, s_perm + st3, n3, wave0, n_waves);
}

// forward noising of the numeric columns, per-row t (gaussian_q_sample, :172-180)
__global__ void __launch_bounds__(256) gauss_qsample_kernel(ScheduleDev s, int B, int n, const float* __restrict__ x0,
                                                            const float* __restrict__ noise, const long long* __restrict__ t,
                                                            float* __restrict__ out, int ld_out) {
    const long long total = (long long)B * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / n), col = (int)(i - (long long)row * n);
        const int tt = (int)t[row];
        out[(size_t)row * ld_out + col] = s.sqrt_alphas_cumprod[tt] * x0[i] + s.sqrt_one_minus_alphas_cumprod[tt] * noise[i];
    }
}

// One training step's draws and the forward noising of the numeric columns in one launch: t ~ U{0..T-1} per row
// (torch.randint of :590), noise ~ N(0, 1) (torch.randn_like of :647), x_t = gaussian_q_sample(x0, t, noise).  Draws are
// keyed by (seed, step word, row / element): a captured step replays with fresh draws by advancing the device step word.
//
// Categorical columns (c.m > 0): x_t ~ q(x_t | x_0) per (row, feature) (q_sample, :494-498) drawn in CLOSED FORM.  The
// reference perturbs the K log-probabilities q_pred(log one-hot(x_0), t) (:403-412) with Gumbel noise and takes the
// arg-max (:506-516); those log-probabilities take two values -- lh for the clean class, lc for every other -- so the
// arg-max is the clean class with probability p = e^lh / (e^lh + (K-1) e^lc) and otherwise uniform over the K-1 others:
// the same distribution from two uniforms per block instead of K Gumbel draws (K logs and Philox words per block were
// 107 us of the scale-sweep step's critical path; this is one thread per block).  lh, lc as the reference's float32
// log_add_exp gives them.
struct TrainDrawCat {
    int m = 0;                          // categorical features (0: none)
    const int* offsets = nullptr;       // [m + 1] class offsets
    const float* log_k = nullptr;       // [m]
    const float* x_cat = nullptr;       // clean class of feature f of row b = (int) x_cat[b * ldx + f]  (`x_cat.long()`, :650)
    long long ldx = 0;
    int* idx0 = nullptr;                // [B, m] clean classes
    int* idxt = nullptr;                // [B, m] noised classes
};

__global__ void __launch_bounds__(256) train_draw_kernel(ScheduleDev s, int B, int n, const float* __restrict__ x0, long long ldx0,
                                                         unsigned long long seed, unsigned step_index,
                                                         const unsigned* __restrict__ step_counter, long long* __restrict__ t_out,
                                                         float* __restrict__ noise, float* __restrict__ out, int ld_out, TrainDrawCat c) {
    const unsigned step_key = step_index + (step_counter ? *step_counter : 0u);
    const int per_row = n > 0 ? n : 1;
    const long long total = (long long)B * per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / per_row), col = (int)(i - (long long)row * per_row);
        const clava_rng::u4 rt = clava_rng::philox4x32_10(seed, (unsigned long long)row, step_key, clava_rng::STREAM_TRAIN_T);
        const int tt = (int)(((unsigned long long)rt.x * (unsigned long long)s.T) >> 32);      // uniform on 0 .. T-1
        if (col == 0) t_out[row] = tt;
        if (n > 0) {
            const clava_rng::u4 r = clava_rng::philox4x32_10(seed, (unsigned long long)i, step_key, clava_rng::STREAM_TRAIN_NOISE);
            const float z = clava_rng::to_normal(r.x, r.y);
            noise[i] = z;
            out[(size_t)row * ld_out + col] = s.sqrt_alphas_cumprod[tt] * x0[(size_t)row * ldx0 + col] + s.sqrt_one_minus_alphas_cumprod[tt] * z;
        }
    }
    const long long pairs = (long long)B * c.m;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / c.m), f = (int)(i - (long long)row * c.m);
        const clava_rng::u4 rt = clava_rng::philox4x32_10(seed, (unsigned long long)row, step_key, clava_rng::STREAM_TRAIN_T);
        const int tt = (int)(((unsigned long long)rt.x * (unsigned long long)s.T) >> 32);      // the row's t, as above
        const int K = c.offsets[f + 1] - c.offsets[f];
        const int hot = (int)c.x_cat[(size_t)row * c.ldx + f];
        const float logk = c.log_k[f];
        const float lca = s.log_cumprod_alpha[tt], uni = s.log_1_min_cumprod_alpha[tt] - logk;
        const float lh = log_add_exp(0.f + lca, uni), lc = log_add_exp(LOG_TINY + lca, uni);
        const clava_rng::u4 r = clava_rng::philox4x32_10(seed, (unsigned long long)i, step_key, clava_rng::STREAM_TRAIN_CAT);
        int drawn;
        if (hot < 0 || hot >= K) {
            drawn = (int)(((unsigned long long)r.y * (unsigned long long)K) >> 32);            // no clean class: K equal log-probabilities
        } else {
            const float p_hot = 1.0f / (1.0f + (float)(K - 1) * exp_nonpos(fminf(lc - lh, 0.f)));
            const int j = (int)(((unsigned long long)r.y * (unsigned long long)(K - 1)) >> 32);    // uniform on 0 .. K-2
            drawn = (clava_rng::to_uniform(r.x) < p_hot || K < 2) ? hot : j + (j >= hot ? 1 : 0);
        }
        c.idx0[i] = hot;
        c.idxt[i] = drawn;
    }
}

// x_in[b, off_f + k] = 0 at k = idx[b, f], log(1e-30) elsewhere: index_to_log_onehot (tab_ddpm/utils.py:127-134) of the noised
// classes for the dense first layer.  A thread owns one column (its feature found once by bisection) and walks the rows.
__global__ void __launch_bounds__(256) cat_onehot_image_kernel(int B, int m, int sumK, const int* __restrict__ offsets,
                                                               const int* __restrict__ idx, float* __restrict__ x_in, long long ld) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= sumK) return;
    int lo = 0, hi = m - 1;                  // the feature f with offsets[f] <= col < offsets[f + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= col) lo = mid; else hi = mid - 1;
    }
    const int k = col - offsets[lo];
    for (int row = blockIdx.y; row < B; row += gridDim.y) x_in[(size_t)row * ld + col] = (idx[(size_t)row * m + lo] == k) ? 0.f : LOG_TINY;
}

// x_T ~ N(0, I): float32 normal draws (the reference draws torch.randn in float32, :1029) stored in the float64 state
__global__ void __launch_bounds__(256) gauss_init_kernel(double* __restrict__ x, int B, int n, unsigned long long seed,
                                                         unsigned long long row_offset) {
    const long long total = (long long)B * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long elem = row_offset * (unsigned long long)n + (unsigned long long)i;
        const clava_rng::u4 r = clava_rng::philox4x32_10(seed, elem, 0u, clava_rng::STREAM_INIT_GAUSS);
        x[i] = (double)clava_rng::to_normal(r.x, r.y);
    }
}

__global__ void __launch_bounds__(256) chain_input_kernel(const double* __restrict__ x, const int* __restrict__ idx,
                                                         const int* __restrict__ offsets, float* __restrict__ x_in, int B,
                                                         int n_num, int m, int sumK) {
    const int D = n_num + sumK;
    const long long n = (long long)B * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / D), col = (int)(i - (long long)row * D);
        float v;
        if (col < n_num) {
            v = (float)x[(size_t)row * n_num + col];
        } else {
            const int c = col - n_num;
            int lo = 0, hi = m;              // feature f with offsets[f] <= c < offsets[f+1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (offsets[mid] <= c) lo = mid; else hi = mid;
            }
            v = (idx[(size_t)row * m + lo] == c - offsets[lo]) ? 0.f : LOG_TINY;
        }
        x_in[i] = v;
    }
}

}  // namespace

int clava_launch_chain_input(const double* x, const int* idx, const int* offsets, float* x_in, int B, int n_num, int m,
                             int sumK, hipStream_t stream) {
    const long long n = (long long)B * (n_num + sumK);
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(chain_input_kernel, dim3(grid), dim3(256), 0, stream, x, idx, offsets, x_in, B, n_num, m, sumK);
    CLAVA_CHECK_LAUNCH("clava_chain_input");
    return CLAVA_OK;
}

int clava_launch_gauss_step(const GaussStepArgs& a, hipStream_t stream) {
    CLAVA_CHECK_ARG(a.B > 0 && a.n_num > 0, "gauss_step: empty shape");
    CLAVA_CHECK_ARG(a.eps && a.x, "gauss_step: null eps/x");
    CLAVA_CHECK_ARG(a.t_rows || (a.t >= 0 && a.t < a.s.T), "gauss_step: t=%d outside [0,%d)", a.t, a.s.T);
    const long long n = (long long)a.B * a.n_num;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    // algorithmic bytes per element: eps 4 + x 8 in + 8 out + x_in 4 (+ grad 4) (+ injected noise 8)
    ProfScope prof(PROF_GAUSS_STEP, 0.0, (double)n * (24 + (a.grad ? 4 : 0) + (a.noise ? 8 : 0)), stream);
    hipLaunchKernelGGL(gauss_step_kernel, dim3(grid), dim3(256), 0, stream, a);
    CLAVA_CHECK_LAUNCH("clava_gauss_step");
    return CLAVA_OK;
}

// Sampler of the reverse step's in-kernel categorical draws: 0 = Gumbel arg-max (the reference's construction, `:506-516`),
// 1 = inverse CDF (same distribution, ~3x fewer instructions).  Injected uniforms and log-probability requests always take
// the Gumbel path.  Default from CLAVA_CAT_SAMPLER (gumbel / icdf); clava_categorical_sampler_set() overrides it.
static int g_cat_sampler = -1;
static int cat_sampler() {
    if (g_cat_sampler < 0) {
        const char* v = getenv("CLAVA_CAT_SAMPLER");
        g_cat_sampler = (v && (v[0] == 'g' || v[0] == '0')) ? 0 : (v && strstr(v, "always")) ? 2 : 1;
    }
    return g_cat_sampler;
}
extern "C" int clava_categorical_sampler_set(int mode) {
    const int prev = cat_sampler();
    if (mode >= 0 && mode <= 2) g_cat_sampler = mode;
    return prev;
}

int clava_launch_multinomial_step(const MultiStepArgs& a_in, hipStream_t stream) {
    MultiStepArgs a = a_in;
    // (measured: 286 vs 673 us at 20000 rows x 50 features of ~165 classes, 17.8 vs 22.1 us on 8 features of ~5 classes:
    // the inverse CDF draws whenever the kernel makes its own noise; modes 1 and 2 are the same since it wins everywhere)
    a.icdf = (a.mode == MULTI_MODE_REVERSE && !a.uniforms && !a.log_prob && cat_sampler() >= 1) ? 1 : 0;
    CLAVA_CHECK_ARG(a.B > 0 && a.m > 0 && a.sumK > 0, "multinomial_step: empty shape");
    CLAVA_CHECK_ARG(a.idx_out && a.offsets && a.log_k, "multinomial_step: null idx_out/offsets/log_k");
    CLAVA_CHECK_ARG(a.mode == MULTI_MODE_INIT || a.idx_in, "multinomial_step: idx_in required");
    CLAVA_CHECK_ARG(a.mode != MULTI_MODE_REVERSE || a.logits, "multinomial_step: logits required");
    CLAVA_CHECK_ARG(a.t_rows || (a.t >= 0 && a.t < a.s.T), "multinomial_step: t=%d outside [0,%d)", a.t, a.s.T);
    const long long pairs = (long long)a.B * a.m;
    const long long blocks = (pairs + 3) / 4;
    // persistent grid: as many workgroups as the chip holds at once, so that every wavefront averages over many blocks
    static int resident[64][4] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    const int slot = a.icdf ? 3 : a.mode;
    if (!resident[dev][slot]) {
        const void* fn = a.icdf ? (const void*)multinomial_step_kernel<MULTI_MODE_REVERSE, true>
                         : a.mode == MULTI_MODE_REVERSE ? (const void*)multinomial_step_kernel<MULTI_MODE_REVERSE>
                         : a.mode == MULTI_MODE_INIT  ? (const void*)multinomial_step_kernel<MULTI_MODE_INIT>
                                                      : (const void*)multinomial_step_kernel<MULTI_MODE_QSAMPLE>;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const char* mult = getenv("CLAVA_STEP_GRID_MULT");
        resident[dev][slot] = per_cu * cus * (mult ? atoi(mult) : 1);
    }
    const int grid = (int)(blocks < resident[dev][slot] ? blocks : resident[dev][slot]);
    // algorithmic bytes: logits 4/class in, log-one-hot 4/class out (+ injected uniforms 4/class), indices 4+4 per feature
    ProfScope prof(PROF_MULTI_STEP, 0.0, (double)a.B * ((double)a.sumK * ((a.logits ? 4 : 0) + (a.x_in ? 4 : 0) + (a.xs.base ? 4 : 0) + (a.uniforms ? 4 : 0)) + 8.0 * a.m), stream);
    if (a.icdf) hipLaunchKernelGGL((multinomial_step_kernel<MULTI_MODE_REVERSE, true>), dim3(grid), dim3(256), 0, stream, a);
    else if (a.mode == MULTI_MODE_REVERSE) hipLaunchKernelGGL(multinomial_step_kernel<MULTI_MODE_REVERSE>, dim3(grid), dim3(256), 0, stream, a);
    else if (a.mode == MULTI_MODE_INIT) hipLaunchKernelGGL(multinomial_step_kernel<MULTI_MODE_INIT>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(multinomial_step_kernel<MULTI_MODE_QSAMPLE>, dim3(grid), dim3(256), 0, stream, a);
    CLAVA_CHECK_LAUNCH("clava_multinomial_step");
    return CLAVA_OK;
}

extern "C" {

int clava_gauss_step_f32(const clava_schedule* sched, int B, int n_num, const float* eps, int ld_eps, double* x,
                         const float* grad, int ld_grad, const double* noise, float* x_in, int ld_xin,
                         double* pred_xstart, const long long* t_rows, int t, int x_is_f32, unsigned long long seed,
                         unsigned long long row_offset, unsigned step_index, void* stream) {
    CLAVA_CHECK_ARG(sched, "gauss_step: null schedule");
    GaussStepArgs a;
    a.s = *sched; a.B = B; a.n_num = n_num;
    a.eps = eps; a.ld_eps = ld_eps; a.x = x; a.grad = grad; a.ld_grad = ld_grad; a.noise = noise;
    a.x_in = x_in; a.ld_xin = ld_xin; a.pred_xstart = pred_xstart; a.t_rows = t_rows; a.t = t; a.x_is_f32 = x_is_f32;
    a.seed = seed; a.row_offset = row_offset; a.step_index = step_index;
    return clava_launch_gauss_step(a, (hipStream_t)stream);
}

int clava_multinomial_step_f32(const clava_schedule* sched, int mode, int B, int m, int sumK, const int* offsets,
                               const float* log_k, const float* logits, int ld_logits, const int* idx_in,
                               int* idx_out, const float* uniforms, int ld_u, float* x_in, int ld_xin,
                               float* log_prob, int ld_lp, const long long* t_rows, int t, unsigned long long seed,
                               unsigned long long row_offset, unsigned step_index, void* stream) {
    return clava_multinomial_step_counter_f32(sched, mode, B, m, sumK, offsets, log_k, logits, ld_logits, idx_in, idx_out, uniforms, ld_u,
                                              x_in, ld_xin, log_prob, ld_lp, t_rows, t, seed, row_offset, step_index, nullptr, stream);
}

int clava_multinomial_step_counter_f32(const clava_schedule* sched, int mode, int B, int m, int sumK, const int* offsets,
                                       const float* log_k, const float* logits, int ld_logits, const int* idx_in,
                                       int* idx_out, const float* uniforms, int ld_u, float* x_in, int ld_xin,
                                       float* log_prob, int ld_lp, const long long* t_rows, int t, unsigned long long seed,
                                       unsigned long long row_offset, unsigned step_index, const unsigned* step_counter,
                                       void* stream) {
    CLAVA_CHECK_ARG(sched, "multinomial_step: null schedule");
    CLAVA_CHECK_ARG(mode >= MULTI_MODE_REVERSE && mode <= MULTI_MODE_QSAMPLE, "multinomial_step: unknown mode %d", mode);
    MultiStepArgs a;
    a.s = *sched; a.B = B; a.m = m; a.sumK = sumK; a.mode = mode;
    a.logits = logits; a.ld_logits = ld_logits; a.idx_in = idx_in; a.idx_out = idx_out;
    a.uniforms = uniforms; a.ld_u = ld_u; a.x_in = x_in; a.ld_xin = ld_xin; a.log_prob = log_prob; a.ld_lp = ld_lp;
    a.offsets = offsets; a.log_k = log_k; a.t_rows = t_rows; a.t = t;
    a.seed = seed; a.row_offset = row_offset; a.step_index = step_index; a.step_counter = step_counter;
    return clava_launch_multinomial_step(a, (hipStream_t)stream);
}

int clava_gauss_qsample_f32(const clava_schedule* sched, int B, int n_num, const float* x0, const float* noise,
                            const long long* t_rows, float* x_t, int ld_xt, void* stream) {
    CLAVA_CHECK_ARG(sched && x0 && noise && t_rows && x_t && B > 0 && n_num > 0 && ld_xt >= n_num, "gauss_qsample: bad arguments");
    const long long n = (long long)B * n_num;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gauss_qsample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *sched, B, n_num, x0, noise, t_rows,
                       x_t, ld_xt);
    CLAVA_CHECK_LAUNCH("clava_gauss_qsample_f32");
    return CLAVA_OK;
}

int clava_train_draw_qsample_f32(const clava_schedule* sched, int B, int n_num, const float* x0, long long ldx0,
                                 unsigned long long seed, unsigned step_index, const unsigned* step_counter, long long* t_rows,
                                 float* noise, float* x_t, int ld_xt, void* stream) {
    CLAVA_CHECK_ARG(sched && t_rows && B > 0 && n_num >= 0 && sched->T > 0, "train_draw_qsample: bad arguments");
    CLAVA_CHECK_ARG(n_num == 0 || (x0 && noise && x_t && ldx0 >= n_num && ld_xt >= n_num), "train_draw_qsample: numeric operands missing");
    const long long n = (long long)B * (n_num > 0 ? n_num : 1);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(train_draw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *sched, B, n_num, x0, ldx0, seed, step_index,
                       step_counter, t_rows, noise, x_t, ld_xt, TrainDrawCat());
    CLAVA_CHECK_LAUNCH("clava_train_draw_qsample_f32");
    return CLAVA_OK;
}

int clava_train_draw_mixed_f32(const clava_schedule* sched, int B, int n_num, const float* x0, long long ldx0, int m, const int* offsets,
                               const float* log_k, unsigned long long seed, unsigned step_index, const unsigned* step_counter,
                               long long* t_rows, float* noise, float* x_t, int ld_xt, int* idx0, int* idxt, void* stream) {
    CLAVA_CHECK_ARG(sched && t_rows && B > 0 && n_num >= 0 && m > 0 && sched->T > 0, "train_draw_mixed: bad arguments");
    CLAVA_CHECK_ARG(x0 && offsets && log_k && idx0 && idxt && ldx0 >= (long long)n_num + m, "train_draw_mixed: categorical operands missing");
    CLAVA_CHECK_ARG(n_num == 0 || (noise && x_t && ld_xt >= n_num), "train_draw_mixed: numeric operands missing");
    long long n = (long long)B * (n_num > 0 ? n_num : 1);
    if ((long long)B * m > n) n = (long long)B * m;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    TrainDrawCat c;
    c.m = m; c.offsets = offsets; c.log_k = log_k; c.x_cat = x0 + n_num; c.ldx = ldx0; c.idx0 = idx0; c.idxt = idxt;
    hipLaunchKernelGGL(train_draw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *sched, B, n_num, x0, ldx0, seed, step_index,
                       step_counter, t_rows, noise, x_t, ld_xt, c);
    CLAVA_CHECK_LAUNCH("clava_train_draw_mixed_f32");
    return CLAVA_OK;
}

int clava_cat_onehot_image_f32(int B, int m, int sumK, const int* offsets, const int* idx, float* x_in, long long ld, void* stream) {
    CLAVA_CHECK_ARG(B > 0 && m > 0 && sumK > 0 && offsets && idx && x_in && ld >= sumK, "cat_onehot_image: bad arguments");
    const int gy = B < 512 ? B : 512;
    hipLaunchKernelGGL(cat_onehot_image_kernel, dim3((sumK + 255) / 256, gy), dim3(256), 0, (hipStream_t)stream, B, m, sumK, offsets, idx,
                       x_in, ld);
    CLAVA_CHECK_LAUNCH("clava_cat_onehot_image_f32");
    return CLAVA_OK;
}

int clava_gauss_init_f64(double* x, int B, int n_num, unsigned long long seed, unsigned long long row_offset, void* stream) {
    CLAVA_CHECK_ARG(x && B > 0 && n_num > 0, "gauss_init: bad arguments");
    const long long n = (long long)B * n_num;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gauss_init_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, B, n_num, seed, row_offset);
    CLAVA_CHECK_LAUNCH("clava_gauss_init_f64");
    return CLAVA_OK;
}

}  // extern "C"
