// Training-side kernels: fused loss + loss gradient, SiLU forward/backward for the time-embedding branch,
// bias gradients, deterministic slab reduction and the fused AdamW + EMA update.
//
// Replaces, per training step (scripts/train.py:30-40, 69): `_gaussian_loss` 'mse' (tab_ddpm/
// gaussian_multinomial_diffsuion.py:282-300), `_multinomial_loss` = compute_Lt + kl_prior (:544-572, 597-607) with
// their autograd backward down to the denoiser output, torch.optim.AdamW.step (scripts/train.py:17, 38) and
// update_ema (scripts/utils_train.py:20-29).  Every reduction runs in a fixed order: results are bitwise
// reproducible from run to run (no float atomics).
#include <cstdlib>
#include "rng.h"
#include "step.h"
#include "wave.h"

namespace {

constexpr float LOG_TINY = -69.07755279541016f;

// ---------------------------------------------------------------------------------------------- Gaussian MSE loss
// loss_rows[b] = mean_c (noise - out)^2 ; d_out[b, c] = 2 (out - noise) / (n_num * B)     (:287-288 and its backward)
// G lanes per row (G = the power of two covering n_num, at most 64): coalesced row segments, fixed-order lane sum.
__global__ void __launch_bounds__(256) gauss_mse_kernel(int B, int n, int G, const float* __restrict__ out, int ld_out,
                                                        const float* __restrict__ noise, float* __restrict__ loss_rows,
                                                        float* __restrict__ d_out, int ld_d, SplitOut ds) {
    const int sub = threadIdx.x & (G - 1);
    const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (row >= B) return;                          // whole groups leave together (G divides 64)
    const float gscale = 2.0f / ((float)n * (float)B);
    float acc = 0.f;
    for (int c = sub; c < n; c += G) {
        const float d = noise[(size_t)row * n + c] - out[(size_t)row * ld_out + c];
        acc += d * d;
        if (d_out) d_out[(size_t)row * ld_d + c] = -d * gscale;
        if (ds.base) store_split(ds, (int)row, ds.col0 + c, -d * gscale);
    }
    for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0) loss_rows[row] = acc / (float)n;
}

// ---------------------------------------------------------------------------------------------- multinomial loss
struct MLossArgs {
    ScheduleDev s;
    int B, m, sumK;
    const float* logits; int ld_logits;   // denoiser output, categorical columns
    const int* idx0;                      // [B, m] clean categories
    const int* idxt;                      // [B, m] noised categories (q_sample)
    const long long* t_rows;
    const int* offsets; const float* log_k;
    float pt;                             // 1 / T as float32 ('uniform' sample_time, :589-593)
    float* loss_terms;                    // [B, m]: Lt / pt + kl_prior of this (row, feature) block
    float* d_logits; int ld_d;            // gradient of mean_b( sum_f loss_terms / m ) w.r.t. logits
    SplitOut ds;                          // the gradient as split rows (the head's backward operand), column ds.col0 + class column; or null
    int pack;                             // 0: one wavefront per block whatever its width (CLAVA_MLOSS_PACK=0)
};

// Per-block (wave-uniform) quantities.  Everything that depends on a class only through "is it the clean category" /
// "is it the noised category" takes two values per block and is evaluated once here, not once per class.
struct LBlock {
    const float* lg;
    float* d;                      // null: split rows only
    int row, col;                  // of the block's first class in the split gradient
    int K, hot0, hott, t;
    float mx, lsum;
    float lca_p, uni_p;            // log cumprod alpha at t-1; log(1 - cumprod alpha_{t-1}) - log K
    float q1_hot, q1_cold;         // q_pred_one_timestep(x_t) of the noised / any other category
    float evt_hot, evt_cold;       // q_pred(x_0, t-1) (identity at t = 0) of the clean / any other category
    float klp_hot, klp_cold;       // prior-KL terms exp(lq) (lq + log K) of the clean / any other category
    // log true posterior q(x_{t-1} | x_t, x_0) and its exponential: four values per block, by (is the clean class, is the
    // noised class), normalised in closed form by loss_block_wave
    // (lt, exp(lt)) of [hh, hc, ch, cc] in this wavefront's four LDS slots (eight more scalars would not fit the scalar
    // registers: the kernel's arguments alone take sixty)
    const float2* post;
    __device__ __forceinline__ float q1(int c) const { return c == hott ? q1_hot : q1_cold; }
    __device__ __forceinline__ float2 posterior(int c) const { return post[(c == hot0 ? 0 : 2) + (c == hott ? 0 : 1)]; }
    __device__ __forceinline__ float ev(float lstart) const {                  // q_pred(., t-1) or identity at t = 0
        return t == 0 ? lstart : log_add_exp_acc(lstart + lca_p, uni_p);
    }
};

// sum_j exp(x_j - mx) of a block with the terms that ARE the maximum counted instead of added (each is exactly 1), so that
// the normaliser mx + log(count + rest) keeps full relative accuracy when one class dominates (log_sum_counted, wave.h)
struct CountedSum {
    float rest = 0.f, n_max = 0.f;
    __device__ __forceinline__ void add(float x, float mx) {
        const float d = x - mx;
        if (d == 0.f) n_max += 1.0f;
        else rest += exp_nonpos(d);
    }
    __device__ __forceinline__ float log_total() const { return log_sum_counted(group_sum<64>(n_max), group_sum<64>(rest)); }
};

// One block of up to 64 * NCH classes held in registers (logit, ev(log-softmax), true-posterior weight / gradient):
// HBM sees each logit once and each gradient once; ten transcendentals per class.
template <int NCH>
__device__ __forceinline__ void loss_block_cached(const MLossArgs& a, LBlock& b, int lane, size_t pair) {
    float v[NCH], e[NCH], w[NCH];
    const int K = b.K;
    float m1 = -INFINITY;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j == 0 || j * 64 < K) {
            const int c = lane + 64 * j;
            v[j] = (c < K) ? b.lg[c] : -INFINITY;
            m1 = fmaxf(m1, v[j]);
        }
    }
    b.mx = group_max<64>(m1);
    {
        CountedSum sm;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            if ((j == 0 || j * 64 < K) && lane + 64 * j < K) sm.add(v[j], b.mx);
        b.lsum = sm.log_total();
    }
    // normaliser of the model posterior (direct per-block log-sum-exp; the true posterior's is closed form, LBlock)
    float mm = -INFINITY;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j == 0 || j * 64 < K) {
            const int c = lane + 64 * j;
            v[j] = (v[j] - b.mx) - b.lsum;                                       // predict_start
            e[j] = b.ev(v[j]);
            if (c < K) mm = fmaxf(mm, e[j] + b.q1(c));
        }
    }
    mm = group_max<64>(mm);
    // w[j] keeps exp(un_model - mm) of the class (exactly 1 at the maximum) until the backward pass, where it gives the
    // model posterior exp(lm) = w[j] / total without a second exponential
    float tot_m;
    float lse_m;
    {
        float n_max = 0.f, rest = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (j == 0 || j * 64 < K) {
                const int c = lane + 64 * j;
                w[j] = 0.f;
                if (c < K) {
                    const float d = (e[j] + b.q1(c)) - mm;
                    if (d == 0.f) { n_max += 1.0f; w[j] = 1.0f; }
                    else { w[j] = exp_nonpos(d); rest += w[j]; }
                }
            }
        }
        n_max = group_sum<64>(n_max); rest = group_sum<64>(rest);
        lse_m = mm + log_sum_counted(n_max, rest);
        tot_m = n_max + rest;
    }
    const float inv_tot_m = 1.0f / tot_m;
    // loss terms: KL(true || model) (t > 0) or decoder NLL (t == 0) (:555-572), and the prior KL (:544-553)
    float W = 0.f, loss = 0.f, klp = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j == 0 || j * 64 < K) {
            const int c = lane + 64 * j;
            if (c < K) {
                const float lm = (e[j] + b.q1(c)) - lse_m;
                if (b.t == 0) {
                    const float wt = (c == b.hot0) ? 1.0f : 1e-30f;              // exp(log_x_start)
                    loss -= wt * lm;
                    W += wt;
                } else {
                    const float2 p = b.posterior(c);
                    loss += p.y * (p.x - lm);
                    W += p.y;
                }
                klp += (c == b.hot0) ? b.klp_hot : b.klp_cold;
            }
        }
    }
    W = group_sum<64>(W); loss = group_sum<64>(loss); klp = group_sum<64>(klp);
    if (lane == 0) a.loss_terms[pair] = loss / a.pt + klp;
    // backward: d loss / d un_model(j) = -w(j) + W * exp(lm(j)); through ev (sigmoid of the log-add-exp) and log-softmax
    float S = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j == 0 || j * 64 < K) {
            const int c = lane + 64 * j;
            if (c < K) {
                const float wt = b.t == 0 ? ((c == b.hot0) ? 1.0f : 1e-30f) : b.posterior(c).y;
                const float g_un = -wt + W * (w[j] * inv_tot_m);
                w[j] = g_un * (b.t == 0 ? 1.0f : exp_nonpos((v[j] + b.lca_p) - e[j]));
                S += w[j];
            }
        }
    }
    S = group_sum<64>(S);
    const float scale = (1.0f / a.pt) / ((float)a.m * (float)a.B);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j == 0 || j * 64 < K) {
            const int c = lane + 64 * j;
            if (c < K) {
                const float g = (w[j] - exp_nonpos(v[j]) * S) * scale;
                if (b.d) b.d[c] = g;
                if (a.ds.base) store_split(a.ds, b.row, b.col + c, g);
            }
        }
    }
}

// Blocks beyond 1024 classes: the same passes, streaming over the classes and recomputing per pass.
__device__ __forceinline__ void loss_block_streamed(const MLossArgs& a, LBlock& b, int lane, size_t pair) {
    const int K = b.K;
    float m1 = -INFINITY;
    for (int c = lane; c < K; c += 64) m1 = fmaxf(m1, b.lg[c]);
    b.mx = group_max<64>(m1);
    {
        CountedSum sm;
        for (int c = lane; c < K; c += 64) sm.add(b.lg[c], b.mx);
        b.lsum = sm.log_total();
    }
    auto lp = [&](int c) { return (b.lg[c] - b.mx) - b.lsum; };
    auto un_model = [&](int c) { return b.ev(lp(c)) + b.q1(c); };
    float mm = -INFINITY;
    for (int c = lane; c < K; c += 64) mm = fmaxf(mm, un_model(c));
    mm = group_max<64>(mm);
    CountedSum sm_m;
    for (int c = lane; c < K; c += 64) sm_m.add(un_model(c), mm);
    const float lse_m = mm + sm_m.log_total();
    auto weight = [&](int c) { return (b.t == 0) ? ((c == b.hot0) ? 1.0f : 1e-30f) : b.posterior(c).y; };
    float W = 0.f, loss = 0.f, klp = 0.f;
    for (int c = lane; c < K; c += 64) {
        const float lm = un_model(c) - lse_m, w = weight(c);
        if (b.t == 0) loss -= w * lm;
        else loss += w * (b.posterior(c).x - lm);
        W += w;
        klp += (c == b.hot0) ? b.klp_hot : b.klp_cold;
    }
    W = group_sum<64>(W); loss = group_sum<64>(loss); klp = group_sum<64>(klp);
    if (lane == 0) a.loss_terms[pair] = loss / a.pt + klp;
    auto g_lp = [&](int c, float lpc) {
        const float evc = b.ev(lpc);
        const float lm = (evc + b.q1(c)) - lse_m;
        return (-weight(c) + W * exp_nonpos(lm)) * (b.t == 0 ? 1.0f : exp_nonpos((lpc + b.lca_p) - evc));
    };
    float S = 0.f;
    for (int c = lane; c < K; c += 64) S += g_lp(c, lp(c));
    S = group_sum<64>(S);
    const float scale = (1.0f / a.pt) / ((float)a.m * (float)a.B);
    for (int c = lane; c < K; c += 64) {
        const float lpc = lp(c);
        const float g = (g_lp(c, lpc) - exp_nonpos(lpc) * S) * scale;
        if (b.d) b.d[c] = g;
        if (a.ds.base) store_split(a.ds, b.row, b.col + c, g);
    }
}

// 64 / G blocks of up to G classes of ONE row per wavefront, one class per lane: the per-class work, the ten group
// reductions and the three normalisers of a block cost the same ~700 wave instructions whether the wavefront holds one
// block of 3 classes or eight of them, and the kernel is VALU bound (207 M wave instructions at the scale-sweep shape =
// 385 us at one instruction per SIMD per 4 cycles: measured 400).  The same functions on the same arguments as the
// wavefront-wide form below -- the two-valued factors are evaluated per lane instead of per block and selected.
struct RowScalars {                // wave-uniform: every block of a work item belongs to one row
    int t;
    float lca_p, l1mca_p, la_t, l1ma_t, lca_T, l1mca_T;
};

template <int G>
__device__ __forceinline__ void loss_block_packed(const MLossArgs& a, const RowScalars& r, int row, int f, bool active, int o, int K,
                                                  float logk) {
    const int c = (int)(threadIdx.x & (G - 1));
    const size_t pair = (size_t)row * a.m + f;
    const int hot0 = a.idx0[pair], hott = a.idxt[pair];
    const bool valid = c < K;
    float v = valid ? a.logits[(size_t)row * a.ld_logits + o + c] : -INFINITY;
    const float uni_p = r.l1mca_p - logk;
    const float h0 = (c == hot0) ? 0.f : LOG_TINY, ht = (c == hott) ? 0.f : LOG_TINY;
    const float q1 = log_add_exp_acc(ht + r.la_t, r.l1ma_t - logk);                      // q_pred_one_timestep(x_t)
    const float evt = r.t == 0 ? h0 : log_add_exp_acc(h0 + r.lca_p, uni_p);               // q_pred(x_0, t-1), identity at t = 0
    const float lq = log_add_exp_acc(h0 + r.lca_T, r.l1mca_T - logk);                     // prior KL (:544-553)
    const float klp_c = exp_nonpos(lq) * (lq + logk);
    const float un_t = evt + q1;
    auto counted_log_total = [&](float x, float mx) {          // CountedSum over the group's valid lanes
        const float d = x - mx;
        const float n_max = (valid && d == 0.f) ? 1.0f : 0.f;
        const float rest = (valid && d != 0.f) ? exp_nonpos(d) : 0.f;
        return log_sum_counted(group_sum<G>(n_max), group_sum<G>(rest));
    };
    const float mx = group_max<G>(v);
    const float lsum = counted_log_total(v, mx);
    v = (v - mx) - lsum;                                                                 // predict_start
    const float e = r.t == 0 ? v : log_add_exp_acc(v + r.lca_p, uni_p);
    const float un_m = e + q1;
    const float mm = group_max<G>(valid ? un_m : -INFINITY), mt = group_max<G>(valid ? un_t : -INFINITY);
    const float lse_m = mm + counted_log_total(un_m, mm), lse_t = mt + counted_log_total(un_t, mt);
    const float lm = un_m - lse_m;
    float w = 0.f, loss = 0.f;
    if (valid) {
        if (r.t == 0) {
            w = (c == hot0) ? 1.0f : 1e-30f;
            loss = -w * lm;
        } else {
            const float lt = un_t - lse_t;
            w = exp_nonpos(lt);
            loss = w * (lt - lm);
        }
    }
    const float W = group_sum<G>(w);
    loss = group_sum<G>(loss);
    const float klp = group_sum<G>(valid ? klp_c : 0.f);
    if (active && c == 0) a.loss_terms[pair] = loss / a.pt + klp;
    float gl = 0.f;
    if (valid) gl = (-w + W * exp_nonpos(lm)) * (r.t == 0 ? 1.0f : exp_nonpos((v + r.lca_p) - e));
    const float S = group_sum<G>(gl);
    const float scale = (1.0f / a.pt) / ((float)a.m * (float)a.B);
    if (active && valid) {
        const float g = (gl - exp_nonpos(v) * S) * scale;
        if (a.d_logits) a.d_logits[(size_t)row * a.ld_d + o + c] = g;
        if (a.ds.base) store_split(a.ds, row, a.ds.col0 + o + c, g);
    }
}

// One (row, feature) block per wavefront: per-block quantities in scalar registers.
__device__ __forceinline__ void loss_block_wave(const MLossArgs& a, const RowScalars& r, int row, int f, int lane, float2* post) {
    const size_t pair = (size_t)row * a.m + f;
    const int o = __builtin_amdgcn_readfirstlane(a.offsets[f]);
    LBlock b;
    b.K = __builtin_amdgcn_readfirstlane(a.offsets[f + 1]) - o;
    const float logk = a.log_k[f];
    b.t = r.t;
    b.hot0 = __builtin_amdgcn_readfirstlane(a.idx0[pair]);
    b.hott = __builtin_amdgcn_readfirstlane(a.idxt[pair]);
    b.lg = a.logits + (size_t)row * a.ld_logits + o;
    b.d = a.d_logits ? a.d_logits + (size_t)row * a.ld_d + o : nullptr;
    b.row = row;
    b.col = a.ds.col0 + o;
    b.lca_p = r.lca_p;
    b.uni_p = r.l1mca_p - logk;
    const float uni_t = r.l1ma_t - logk;
    // The block's six log-add-exps -- q1 (hot, cold), ev at the clean / any other class, the prior's lq (hot, cold) -- in
    // ONE evaluation: lane j computes the j-th (the same function on the same arguments as six scalar calls: identical
    // bits), then the results are read back by lane.
    {
        const float hot_cold = (lane & 1) ? LOG_TINY : 0.f;
        const int which = lane >> 1;
        const float x = hot_cold + (which == 0 ? r.la_t : which == 1 ? b.lca_p : r.lca_T);
        const float y = which == 0 ? uni_t : which == 1 ? b.uni_p : r.l1mca_T - logk;
        const float lae = log_add_exp_acc(x, y);
        const float kl = exp_nonpos(lae) * (lae + logk);
        auto rl = [](float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); };
        b.q1_hot = rl(lae, 0);
        b.q1_cold = rl(lae, 1);
        b.evt_hot = b.t == 0 ? 0.f : rl(lae, 2);                 // ev is the identity at t = 0
        b.evt_cold = b.t == 0 ? LOG_TINY : rl(lae, 3);
        b.klp_hot = rl(kl, 4);
        b.klp_cold = rl(kl, 5);
        // True posterior (:555-566 via q_posterior, :426-452): its un-normalised logarithm (c == clean ? evt_hot : evt_cold)
        // + q1(c) takes four values -- lane j < 4 holds value j of [hh, hc, ch, cc] -- carried by (1 | 0, 1 | 0, 1 | 0, the
        // rest) of the K classes, so the normaliser is a four-term counted log-sum-exp instead of a pass over the classes
        // (two exponentials per class before), and exp(lt) one evaluation for the four values.
        const int jj = lane & 3;
        const float un = ((jj < 2) ? b.evt_hot : b.evt_cold) + ((jj & 1) ? b.q1_cold : b.q1_hot);
        const bool in0 = b.hot0 >= 0 && b.hot0 < b.K, int_ = b.hott >= 0 && b.hott < b.K;
        const bool same = in0 && b.hot0 == b.hott;
        const int n_hh = same ? 1 : 0, n_hc = (in0 && !same) ? 1 : 0, n_ch = (int_ && !same) ? 1 : 0;
        const int cnt_i = jj == 0 ? n_hh : jj == 1 ? n_hc : jj == 2 ? n_ch : b.K - n_hh - n_hc - n_ch;
        const float cnt = (float)cnt_i;
        const float um = cnt_i > 0 ? un : -INFINITY;
        const float mt = fmaxf(fmaxf(rl(um, 0), rl(um, 1)), fmaxf(rl(um, 2), rl(um, 3)));
        const float d = um - mt;                                  // -inf for an absent value: contributes exp(-inf) * 0
        const float at_max = (cnt_i > 0 && d == 0.f) ? cnt : 0.f;
        const float below = (cnt_i > 0 && d != 0.f) ? cnt * exp_nonpos(d) : 0.f;
        const float n_max = (rl(at_max, 0) + rl(at_max, 1)) + (rl(at_max, 2) + rl(at_max, 3));
        const float rest = (rl(below, 0) + rl(below, 1)) + (rl(below, 2) + rl(below, 3));
        const float lse_t = mt + log_sum_counted(n_max, rest);
        const float ltj = un - lse_t;
        if (lane < 4) post[lane] = make_float2(ltj, exp_nonpos(ltj));       // read back by this wavefront only: LDS keeps its order
        b.post = post;
    }
    if (b.K <= 256) loss_block_cached<4>(a, b, lane, pair);
    else if (b.K <= 1024) loss_block_cached<16>(a, b, lane, pair);
    else loss_block_streamed(a, b, lane, pair);
}

__device__ __forceinline__ RowScalars row_scalars(const MLossArgs& a, int row, float lca_T, float l1mca_T) {
    RowScalars r;
    r.t = __builtin_amdgcn_readfirstlane((int)a.t_rows[row]);
    const int tp = max(r.t - 1, 0);
    r.lca_p = a.s.log_cumprod_alpha[tp];
    r.l1mca_p = a.s.log_1_min_cumprod_alpha[tp];
    r.la_t = a.s.log_alpha[r.t];
    r.l1ma_t = a.s.log_1_min_alpha[r.t];
    r.lca_T = lca_T;
    r.l1mca_T = l1mca_T;
    return r;
}

// All (row, feature) blocks of the `n_feat` features listed in `perm` (null: features 0 .. n_feat-1), 64 / G blocks of one
// row per wavefront work item.
template <int G>
__device__ __forceinline__ void loss_bucket(const MLossArgs& a, const int* s_off, const float* s_logk, const int* perm, int n_feat,
                                            long long wave0, long long n_waves, float2* post) {
    constexpr int PER = 64 / G;
    const int W = (n_feat + PER - 1) / PER;                 // work items per row
    if (W == 0) return;
    // a stride coprime with W makes every wavefront walk through all the features of the bucket (block costs differ by
    // up to 16x)
    auto gcd = [](long long x, long long y) { while (y) { const long long q = x % y; x = y; y = q; } return x; };
    while (n_waves > 1 && gcd(n_waves, W) != 1) --n_waves;
    if (wave0 >= n_waves || wave0 >= (long long)a.B * W) return;
    const int lane = threadIdx.x & 63, sub = lane / G;
    const float lca_T = a.s.log_cumprod_alpha[a.s.T - 1], l1mca_T = a.s.log_1_min_cumprod_alpha[a.s.T - 1];
    // (row, item) advance by n_waves items per iteration without a division in the loop
    int row = (int)(wave0 / W), w = (int)(wave0 - (long long)row * W);
    const int d_row = (int)(n_waves / W), d_w = (int)(n_waves - (long long)d_row * W);
    for (; row < a.B; row += d_row, w += d_w) {
        if (w >= W) { w -= W; ++row; if (row >= a.B) break; }
        const RowScalars r = row_scalars(a, row, lca_T, l1mca_T);
        if (G == 64) {
            loss_block_wave(a, r, row, perm ? perm[w] : w, lane, post);
        } else {
            const int fi = w * PER + sub;
            const int f = perm[min(fi, n_feat - 1)];        // a group beyond the bucket repeats its last feature and stores nothing
            const int o = s_off[f];
            loss_block_packed<G>(a, r, row, f, fi < n_feat, o, s_off[f + 1] - o, s_logk[f]);
        }
    }
}

__global__ void __launch_bounds__(256) multinomial_loss_kernel(MLossArgs a) {
    __shared__ int s_off[MAX_LDS_FEATURES + 1];
    __shared__ float s_logk[MAX_LDS_FEATURES];
    __shared__ int s_perm[MAX_LDS_FEATURES];
    __shared__ int s_start[5];
    __shared__ float2 s_post[4][4];                     // per wavefront: the true posterior's four values (LBlock::post)
    float2* post = s_post[threadIdx.x >> 6];
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    if (a.m > MAX_LDS_FEATURES || !a.pack) {            // more features than the LDS tables hold: one wavefront per block
        loss_bucket<64>(a, nullptr, nullptr, nullptr, a.m, wave0, n_waves, post);
        return;
    }
    for (int i = threadIdx.x; i <= a.m; i += blockDim.x) s_off[i] = a.offsets[i];
    for (int i = threadIdx.x; i < a.m; i += blockDim.x) s_logk[i] = a.log_k[i];
    __syncthreads();
    order_features_by_width(a.m, s_off, s_perm, s_start);
    const int st1 = s_start[1], st2 = s_start[2], st3 = s_start[3];
    // one loop per width (a single loop over mixed items would keep every variant's registers live at once)
    loss_bucket<8>(a, s_off, s_logk, s_perm, st1, wave0, n_waves, post);
    loss_bucket<16>(a, s_off, s_logk, s_perm + st1, st2 - st1, wave0, n_waves, post);
    loss_bucket<32>(a, s_off, s_logk, s_perm + st2, st3 - st2, wave0, n_waves, post);
    loss_bucket<64>(a, s_off, s_logk, s_perm + st3, a.m - st3, wave0, n_waves, post);
}

// ---------------------------------------------------------------------------------------------- reductions
// out[0] = scale * sum(v[0..n)) in a fixed order (one workgroup: per-thread strided partials, then an LDS tree)
__global__ void __launch_bounds__(1024) sum_kernel(const float* __restrict__ v, long long n, float scale, float* __restrict__ out) {
    __shared__ float part[1024];
    float acc = 0.f;
    // eight loads in flight per thread, added in index order (the same sum as one load per iteration, whose 200 dependent
    // round trips took 70 us for the scale-sweep step's 204 800 loss terms)
    long long i = threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = v[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
    }
    for (; i < n; i += 1024) acc += v[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] * scale;
}

// db_slabs[s][n] = sum over the s-th batch range of dy[:, n]: bias gradients, same split scheme as the weight slabs.
// One workgroup owns 32 columns of one batch range: 8 column groups (one float4 each) x 32 row lanes; every thread sums
// its rows with four independent accumulators, the 32 row lanes are combined through LDS in a fixed order (bitwise
// reproducible, no atomics).  HBM-bound: 4*M*N bytes.
template <bool VEC>
__global__ void __launch_bounds__(256) colsum_slabs_kernel(const float* __restrict__ dy, int ld, int M, int N, int splits,
                                                           float* __restrict__ slabs, long long slab_stride) {
    __shared__ float part[32][33];
    const int cg = threadIdx.x & 7, q = threadIdx.x >> 3;       // column group, row lane
    const int col = blockIdx.x * 32 + cg * 4;
    const int rows_per = (M + splits - 1) / splits;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
        if (col < N) {                                           // the row holds the whole float4 (lddy >= round_up(N, 4))
            int r = r0 + q;
            for (; r + 32 < r1; r += 64) {
                const float4 u = *reinterpret_cast<const float4*>(dy + (size_t)r * ld + col);
                const float4 v = *reinterpret_cast<const float4*>(dy + (size_t)(r + 32) * ld + col);
                a0[0] += u.x; a0[1] += u.y; a0[2] += u.z; a0[3] += u.w;
                a1[0] += v.x; a1[1] += v.y; a1[2] += v.z; a1[3] += v.w;
            }
            if (r < r1) {
                const float4 u = *reinterpret_cast<const float4*>(dy + (size_t)r * ld + col);
                a0[0] += u.x; a0[1] += u.y; a0[2] += u.z; a0[3] += u.w;
            }
        }
    } else {
        for (int r = r0 + q; r < r1; r += 32)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (col + c < N) a0[c] += dy[(size_t)r * ld + col + c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) part[q][cg * 4 + c] = a0[c] + a1[c];
    __syncthreads();
    if (threadIdx.x < 32 && blockIdx.x * 32 + threadIdx.x < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += part[i][threadIdx.x];
        slabs[(size_t)blockIdx.y * slab_stride + blockIdx.x * 32 + threadIdx.x] = t;
    }
}

// The same sums over split rows (csrc/split.h): element = (hi + lo) * inv_scale, so the gradient planes the input-gradient
// chain produces feed the bias gradients without a float32 copy.  32 columns = one 128-byte block of a row.
typedef _Float16 colsum_h4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) colsum_slabs_split_kernel(const char* __restrict__ src, long long pitch, int M, int N, int splits,
                                                                 float inv_scale, float* __restrict__ slabs, long long slab_stride) {
    __shared__ float part[32][33];
    const int cg = threadIdx.x & 7, q = threadIdx.x >> 3;       // column group (4 columns), row lane
    const int rows_per = (M + splits - 1) / splits;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    const char* base = src + (long long)blockIdx.x * 128 + cg * 8;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + q; r < r1; r += 32) {
        const colsum_h4 hi = *reinterpret_cast<const colsum_h4*>(base + (long long)r * pitch);
        const colsum_h4 lo = *reinterpret_cast<const colsum_h4*>(base + (long long)r * pitch + 64);
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] += (float)hi[c] + (float)lo[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) part[q][cg * 4 + c] = a[c];
    __syncthreads();
    if (threadIdx.x < 32 && blockIdx.x * 32 + threadIdx.x < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += part[i][threadIdx.x];
        slabs[(size_t)blockIdx.y * slab_stride + blockIdx.x * 32 + threadIdx.x] = t * inv_scale;
    }
}

// out[i] = scale * sum_s slabs[s][i]
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, int S, long long stride, long long n,
                                                          float scale, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float g = 0.f;
        for (int s = 0; s < S; ++s) g += slabs[(size_t)s * stride + i];
        out[i] = g * scale;
    }
}

// ---------------------------------------------------------------------------------------------- SiLU
__global__ void __launch_bounds__(256) silu_kernel(const float* __restrict__ z, float* __restrict__ a, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = z[i];
        a[i] = v / (1.0f + expf(-v));
    }
}
// dz = da * sigmoid(z) * (1 + z * (1 - sigmoid(z)))
__global__ void __launch_bounds__(256) silu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                                       float* __restrict__ dz, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = z[i];
        const float sg = 1.0f / (1.0f + expf(-v));
        dz[i] = da[i] * (sg * (1.0f + v * (1.0f - sg)));
    }
}

// ---------------------------------------------------------------------------------------------- AdamW + EMA
struct AdamArgs {
    float* p; float* ema; float* m; float* v;
    const float* g; int S; long long stride; float gscale;   // gradient = gscale * sum_s g[s * stride + i]
    long long n;
    float lr_wd;       // lr * weight_decay
    float beta1_c;     // 1 - beta1
    float beta2, beta2_c;
    float step_size;   // lr / (1 - beta1^step)
    float bc2_sqrt;    // sqrt(1 - beta2^step)
    float eps;
    float ema_rate, ema_c;
    const float* hyper;   // device copy of the seven step-dependent / tunable scalars (clava_adamw_hyper_f32's layout), or null
};

// torch.optim.AdamW single-tensor update order: decay, lerp first moment, second moment, bias-corrected step; then the EMA
__global__ void __launch_bounds__(256) adamw_ema_kernel(AdamArgs a) {
    if (a.hyper) {        // a captured step: the launch arguments are frozen, the scalars come from memory
        a.lr_wd = a.hyper[0]; a.beta1_c = a.hyper[1]; a.beta2 = a.hyper[2]; a.beta2_c = a.hyper[3];
        a.step_size = a.hyper[4]; a.bc2_sqrt = a.hyper[5]; a.eps = a.hyper[6];
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
        float g = 0.f;
        for (int s = 0; s < a.S; ++s) g += a.g[(size_t)s * a.stride + i];
        g *= a.gscale;
        float p = a.p[i];
        p = p * (1.0f - a.lr_wd);
        float m = a.m[i];
        m = m + a.beta1_c * (g - m);
        float v = a.v[i];
        v = v * a.beta2 + a.beta2_c * (g * g);
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        p = p - a.step_size * (m / denom);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
        if (a.ema) a.ema[i] = a.ema[i] * a.ema_rate + p * a.ema_c;
    }
}

// dst[0..n) = (word & 1 ? p1 : p0)[0..n): the captured step's input fetch.  p0 / p1 are PINNED HOST buffers the host fills
// alternately (step parity), read here straight over the bus: no host-synchronous copy sits between two replays.
__global__ void __launch_bounds__(256) stage_fetch_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                          const unsigned* __restrict__ word, float* __restrict__ dst, long long n) {
    const float* src = (*word & 1u) ? p1 : p0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

inline int grid_for(long long n, int cap = 4096) {
    const long long b = (n + 255) / 256;
    return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace

extern "C" {

// d_out and / or split rows: the gradient of the step's loss w.r.t. the denoiser output, as float32 and / or in the split
// layout the head's backward products read (split.h), `col0` = column of the kernel's first column inside those rows
static bool split_dst_ok(const void* d_split, long long pitch, int shift, int col_end) {
    if (!d_split) return true;
    return pitch >= 4LL * ((col_end + 31) & ~31) && pitch % 16 == 0 && (reinterpret_cast<uintptr_t>(d_split) & 15) == 0 && shift >= -126 && shift <= 126;
}
static SplitOut split_dst(void* d_split, long long pitch, int shift, int col0, int* overflow) {
    SplitOut o;
    o.base = static_cast<char*>(d_split); o.pitch = pitch; o.scale = ldexpf(1.f, shift); o.col0 = col0; o.overflow = overflow;
    return o;
}

int clava_gauss_mse_loss_split_f32(int B, int n_num, const float* model_out, int ld_out, const float* noise, float* loss_rows,
                                   float* d_out, int ld_d, void* d_split, long long pitch, int shift, int* overflow, void* stream) {
    CLAVA_CHECK_ARG(B > 0 && n_num > 0 && model_out && noise && loss_rows && (d_out || d_split), "gauss_mse_loss: bad arguments");
    CLAVA_CHECK_ARG(split_dst_ok(d_split, pitch, shift, n_num), "gauss_mse_loss: split rows must be 16-byte aligned and hold n_num columns");
    ProfScope prof(PROF_LOSS, 0.0, 12.0 * B * n_num, (hipStream_t)stream);
    int G = 1;
    while (G < n_num && G < 64) G <<= 1;
    const long long blocks = ((long long)B * G + 255) / 256;
    hipLaunchKernelGGL(gauss_mse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, B, n_num, G, model_out, ld_out,
                       noise, loss_rows, d_out, ld_d, split_dst(d_split, pitch, shift, 0, overflow));
    CLAVA_CHECK_LAUNCH("clava_gauss_mse_loss_f32");
    return CLAVA_OK;
}

int clava_gauss_mse_loss_f32(int B, int n_num, const float* model_out, int ld_out, const float* noise, float* loss_rows,
                             float* d_out, int ld_d, void* stream) {
    CLAVA_CHECK_ARG(d_out, "gauss_mse_loss: bad arguments");
    return clava_gauss_mse_loss_split_f32(B, n_num, model_out, ld_out, noise, loss_rows, d_out, ld_d, nullptr, 0, 0, nullptr, stream);
}

int clava_multinomial_loss_split_f32(const clava_schedule* sched, int B, int m, int sumK, const int* offsets, const float* log_k,
                                     const float* logits, int ld_logits, const int* idx0, const int* idxt, const long long* t_rows,
                                     float pt, float* loss_terms, float* d_logits, int ld_d, void* d_split, long long pitch, int shift,
                                     int col0, int* overflow, void* stream) {
    CLAVA_CHECK_ARG(sched && B > 0 && m > 0 && sumK > 0 && offsets && log_k && logits && idx0 && idxt && t_rows && loss_terms &&
                        (d_logits || d_split) && col0 >= 0,
                    "multinomial_loss: bad arguments");
    CLAVA_CHECK_ARG(split_dst_ok(d_split, pitch, shift, col0 + sumK), "multinomial_loss: split rows must be 16-byte aligned and hold col0 + sumK columns");
    MLossArgs a;
    a.s = *sched; a.B = B; a.m = m; a.sumK = sumK; a.logits = logits; a.ld_logits = ld_logits; a.idx0 = idx0; a.idxt = idxt;
    a.t_rows = t_rows; a.offsets = offsets; a.log_k = log_k; a.pt = pt; a.loss_terms = loss_terms; a.d_logits = d_logits; a.ld_d = ld_d;
    a.ds = split_dst(d_split, pitch, shift, col0, overflow);
    static const int pack_env = [] { const char* e = getenv("CLAVA_MLOSS_PACK"); return (e && e[0] == '0') ? 0 : 1; }();
    a.pack = pack_env;
    const long long blocks = ((long long)B * m + 3) / 4;
    // persistent grid: as many workgroups as the chip holds at once, so that every wavefront averages over many blocks
    static int resident[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!resident[dev]) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)multinomial_loss_kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident[dev] = per_cu * cus;
    }
    ProfScope prof(PROF_LOSS, 0.0, 8.0 * B * sumK, (hipStream_t)stream);
    hipLaunchKernelGGL(multinomial_loss_kernel, dim3((int)(blocks < resident[dev] ? blocks : resident[dev])), dim3(256), 0, (hipStream_t)stream, a);
    CLAVA_CHECK_LAUNCH("clava_multinomial_loss_f32");
    return CLAVA_OK;
}

int clava_multinomial_loss_f32(const clava_schedule* sched, int B, int m, int sumK, const int* offsets, const float* log_k,
                               const float* logits, int ld_logits, const int* idx0, const int* idxt,
                               const long long* t_rows, float pt, float* loss_terms, float* d_logits, int ld_d,
                               void* stream) {
    CLAVA_CHECK_ARG(d_logits, "multinomial_loss: bad arguments");
    return clava_multinomial_loss_split_f32(sched, B, m, sumK, offsets, log_k, logits, ld_logits, idx0, idxt, t_rows, pt, loss_terms, d_logits,
                                            ld_d, nullptr, 0, 0, 0, nullptr, stream);
}

int clava_sum_f32(const float* v, long long n, float scale, float* out, void* stream) {
    CLAVA_CHECK_ARG(v && out && n > 0, "sum: bad arguments");
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, n, scale, out);
    CLAVA_CHECK_LAUNCH("clava_sum_f32");
    return CLAVA_OK;
}

int clava_colsum_slabs_f32(const float* dy, int lddy, int M, int N, int splits, float* db_slabs, long long slab_stride,
                           void* stream) {
    CLAVA_CHECK_ARG(dy && db_slabs && M > 0 && N > 0 && splits >= 1 && splits <= 64 && lddy >= N, "colsum_slabs: bad arguments");
    ProfScope prof(PROF_MISC, 0.0, 4.0 * M * N, (hipStream_t)stream);
    // float4 loads need 16-byte aligned rows that hold the whole last float4 (columns beyond N inside the row are read and
    // their sums discarded)
    const bool vec = (lddy % 4 == 0) && lddy >= ((N + 3) & ~3) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(colsum_slabs_kernel<true>, dim3((N + 31) / 32, splits), dim3(256), 0, (hipStream_t)stream, dy, lddy, M, N,
                           splits, db_slabs, slab_stride);
    else
        hipLaunchKernelGGL(colsum_slabs_kernel<false>, dim3((N + 31) / 32, splits), dim3(256), 0, (hipStream_t)stream, dy, lddy, M, N,
                           splits, db_slabs, slab_stride);
    CLAVA_CHECK_LAUNCH("clava_colsum_slabs_f32");
    return CLAVA_OK;
}

int clava_colsum_slabs_split(const void* dy_split, long long pitch, int shift, int M, int N, int splits, float* db_slabs,
                             long long slab_stride, void* stream) {
    CLAVA_CHECK_ARG(dy_split && db_slabs && M > 0 && N > 0 && splits >= 1 && splits <= 64, "colsum_slabs_split: bad arguments");
    const int Np = (N + 31) & ~31;
    CLAVA_CHECK_ARG(pitch >= 4LL * Np && pitch % 8 == 0 && (reinterpret_cast<uintptr_t>(dy_split) & 7) == 0 && shift >= -30 && shift <= 30,
                    "colsum_slabs_split: pitch / alignment / shift");
    ProfScope prof(PROF_MISC, 0.0, 4.0 * M * N, (hipStream_t)stream);
    hipLaunchKernelGGL(colsum_slabs_split_kernel, dim3(Np / 32, splits), dim3(256), 0, (hipStream_t)stream, (const char*)dy_split, pitch, M,
                       N, splits, ldexpf(1.f, -shift), db_slabs, slab_stride);
    CLAVA_CHECK_LAUNCH("clava_colsum_slabs_split");
    return CLAVA_OK;
}

int clava_slab_reduce_f32(const float* slabs, int splits, long long slab_stride, long long n, float scale, float* out,
                          void* stream) {
    CLAVA_CHECK_ARG(slabs && out && splits >= 1 && n > 0, "slab_reduce: bad arguments");
    ProfScope prof(PROF_MISC, 0.0, 4.0 * n * (splits + 1), (hipStream_t)stream);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, slabs, splits, slab_stride, n,
                       scale, out);
    CLAVA_CHECK_LAUNCH("clava_slab_reduce_f32");
    return CLAVA_OK;
}

int clava_silu_f32(const float* z, float* a, long long n, void* stream) {
    CLAVA_CHECK_ARG(z && a && n > 0, "silu: bad arguments");
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z, a, n);
    CLAVA_CHECK_LAUNCH("clava_silu_f32");
    return CLAVA_OK;
}

int clava_silu_backward_f32(const float* z, const float* da, float* dz, long long n, void* stream) {
    CLAVA_CHECK_ARG(z && da && dz && n > 0, "silu_backward: bad arguments");
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z, da, dz, n);
    CLAVA_CHECK_LAUNCH("clava_silu_backward_f32");
    return CLAVA_OK;
}

int clava_adamw_ema_f32(float* param, float* ema, float* exp_avg, float* exp_avg_sq, const float* grad_slabs, int splits,
                        long long slab_stride, float grad_scale, long long n, double lr, double beta1, double beta2,
                        double eps, double weight_decay, long long step, double ema_rate, void* stream) {
    CLAVA_CHECK_ARG(param && exp_avg && exp_avg_sq && grad_slabs && splits >= 1 && n > 0 && step >= 1, "adamw_ema: bad arguments");
    AdamArgs a;
    a.p = param; a.ema = ema; a.m = exp_avg; a.v = exp_avg_sq; a.g = grad_slabs; a.S = splits; a.stride = slab_stride;
    a.gscale = grad_scale; a.n = n;
    a.lr_wd = (float)(lr * weight_decay);
    a.beta1_c = (float)(1.0 - beta1);
    a.beta2 = (float)beta2; a.beta2_c = (float)(1.0 - beta2);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = (float)eps;
    a.ema_rate = (float)ema_rate; a.ema_c = (float)(1.0 - ema_rate);
    a.hyper = nullptr;
    ProfScope prof(PROF_ADAMW, 0.0, 4.0 * n * (7 + splits + (ema ? 2 : 0)), (hipStream_t)stream);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, a);
    CLAVA_CHECK_LAUNCH("clava_adamw_ema_f32");
    return CLAVA_OK;
}

int clava_stage_fetch_f32(const float* host0, const float* host1, const unsigned* word, float* dst, long long n, void* stream) {
    CLAVA_CHECK_ARG(host0 && host1 && word && dst && n > 0, "stage_fetch: bad arguments");
    hipLaunchKernelGGL(stage_fetch_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, host0, host1, word, dst, n);
    CLAVA_CHECK_LAUNCH("clava_stage_fetch_f32");
    return CLAVA_OK;
}

int clava_adamw_hyper_f32(double lr, double beta1, double beta2, double eps, double weight_decay, long long step, float* hyper8) {
    CLAVA_CHECK_ARG(hyper8 && step >= 1, "adamw_hyper: bad arguments");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hyper8[0] = (float)(lr * weight_decay);
    hyper8[1] = (float)(1.0 - beta1);
    hyper8[2] = (float)beta2;
    hyper8[3] = (float)(1.0 - beta2);
    hyper8[4] = (float)(lr / bc1);
    hyper8[5] = (float)sqrt(bc2);
    hyper8[6] = (float)eps;
    hyper8[7] = 0.f;
    return CLAVA_OK;
}

int clava_adamw_ema_hyper_f32(float* param, float* ema, float* exp_avg, float* exp_avg_sq, const float* grad_slabs, int splits,
                              long long slab_stride, float grad_scale, long long n, const float* hyper_dev, double ema_rate,
                              void* stream) {
    CLAVA_CHECK_ARG(param && exp_avg && exp_avg_sq && grad_slabs && splits >= 1 && n > 0 && hyper_dev, "adamw_ema_hyper: bad arguments");
    AdamArgs a = {};
    a.p = param; a.ema = ema; a.m = exp_avg; a.v = exp_avg_sq; a.g = grad_slabs; a.S = splits; a.stride = slab_stride;
    a.gscale = grad_scale; a.n = n;
    a.ema_rate = (float)ema_rate; a.ema_c = (float)(1.0 - ema_rate);
    a.hyper = hyper_dev;
    ProfScope prof(PROF_ADAMW, 0.0, 4.0 * n * (7 + splits + (ema ? 2 : 0)), (hipStream_t)stream);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, a);
    CLAVA_CHECK_LAUNCH("clava_adamw_ema_hyper_f32");
    return CLAVA_OK;
}

}  // extern "C"
