/* clava_hip.h - C ABI of libclava_hip.so, the gfx950 (MI355X) kernels behind ClavaDDPM's diffusion hot path.
 *
 * The reference (weipang142857/ClavaDDPM) is pure Python and has no FFI of its own; these are the entry
 * points a maintainer binds with ctypes.  Each declaration cites the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major memory unless the comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only enqueue work;
 *   - return value: 0 on success, negative on error (-1 bad argument, -2 launch failure, -3 size limit);
 *     clava_last_error() returns the message of the calling thread's last failure;
 *   - nothing here allocates or frees device memory; the caller owns every buffer, including workspaces.
 */
#ifndef CLAVA_HIP_H
#define CLAVA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLAVA_HIP_VERSION 110

/* activation codes */
#define CLAVA_ACT_NONE_ 0
#define CLAVA_ACT_RELU_ 1
#define CLAVA_ACT_LEAKY_ 2 /* negative slope 0.01 = nn.LeakyReLU() default (pipeline_utils.py:615) */
#define CLAVA_ACT_SILU_ 3

int clava_version(void);
const char* clava_last_error(void);
int clava_device_count(void); /* host-only probe: number of HIP devices, 0 when there is none */

/* ---- per-launch timing (measurement only; no counterpart in the reference) ------------------------------------
 * clava_prof_enable(n) makes the library bracket each of its next n kernel launches with two hipEvents on the
 * launch's own stream and tag it with its kernel family and algorithmic FLOPs / bytes; clava_prof_collect waits for
 * the recorded events and returns per-family totals (arrays of clava_prof_family_count() entries) and the number of
 * launches recorded.  bench.py uses it for the `roofline` object.  Not thread-safe; host-side only. */
int clava_prof_enable(int max_records);
int clava_prof_disable(void);
int clava_prof_family_count(void);
const char* clava_prof_family_name(int family);
int clava_prof_collect(int* launches, double* total_ms, double* total_flops, double* total_bytes);

/* ---- dense layers -------------------------------------------------------------------------------------------
 * y[M,N] = act( (x*in_scale + in_shift)[M,K] . w[N,K]^T + bias[N] + rowadd[M,N] )
 * Replaces nn.Linear (+ReLU / LeakyReLU / SiLU) of MLPDiffusion (tab_ddpm/modules.py:153-158, 191, 444-449, 459)
 * and Classifier (pipeline_utils.py:589, 597-623).  `rowadd` is the time-embedding term of `proj(x) + emb`
 * (modules.py:459): ld_rowadd == 0 broadcasts one row (batch-constant t while sampling).  `in_scale`/`in_shift`
 * (both or neither) apply an eval-mode BatchNorm1d to the input columns on the fly (pipeline_utils.py:609, 616).
 * bias, rowadd, in_scale, in_shift may be NULL. */
int clava_gemm_bias_act_f32(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowadd,
                            int ld_rowadd, const float* in_scale, const float* in_shift, float* y, int ldy, int M,
                            int N, int K, int act, void* stream);

/* dx[M,K_in] = alpha * col_scale[K_in] * act'(act_out)[M,K_in] * ( dy[M,N_out] . w[N_out,K_in] )
 * Input-gradient of a Linear, with the derivative of the PREVIOUS layer's activation (read off its stored
 * output `act_out`: factor 1 where act_out > 0 else `slope`) and BatchNorm scale fused in.  Replaces autograd's
 * backward through the layers above (scripts/train.py:37; torch.autograd.grad in cond_fn, pipeline_utils.py:864).
 * act_out and col_scale may be NULL. */
int clava_gemm_dgrad_f32(const float* dy, int lddy, const float* w, int ldw, const float* act_out, int ld_act,
                         float slope, const float* col_scale, float alpha, float* dx, int lddx, int M, int N_out,
                         int K_in, void* stream);

/* dw_slabs[s*slab_stride + n*lddw + k] = partial_s( dy[M,N_out]^T . x[M,K_in] ), s < splits: the batch dimension is cut into
 * `splits` ranges so that small weight matrices still fill the chip; the caller sums the slabs in a fixed order
 * (clava_adamw_ema_f32 does).  Replaces the weight-gradient half of loss.backward() (scripts/train.py:37). */
int clava_gemm_wgrad_f32(const float* dy, int lddy, const float* x, int ldx, float* dw_slabs, int lddw,
                         long long slab_stride, int M, int N_out, int K_in, int splits, void* stream);

/* ---- dense layers on the f16 matrix cores at float32 accuracy ("split-f16", clavaddpm_amd/csrc/split.h) ---------
 * Same nn.Linear products as above (tab_ddpm/modules.py:153-158, 459; pipeline_utils.py:597-630; the input gradient
 * behind cond_fn, pipeline_utils.py:864), for the sampling loop where weights are constant over a chain.  A float32
 * value v is stored as two float16 planes hi = f16(v*2^shift), lo = f16(v*2^shift - hi) (22 significand bits); a
 * product is three v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi) into a float32 accumulator.
 * "Split rows": row-major, columns padded to a multiple of 32 (pad = 0), each 32-column block = 32 hi halves then
 * 32 lo halves (128 bytes); `pitch` = bytes per row, >= 4*round_up(cols,32), a multiple of 16.
 *
 * clava_split_pack_f32 / clava_split_unpack_f32 convert float32 [rows, cols] (leading dimension ld) to / from split
 * rows.  *overflow (device int, may be NULL) is set to 1 if a scaled magnitude exceeds the float16 range. */
int clava_split_pack_f32(const float* src, long long ld, long long rows, int cols, int shift, void* dst, long long pitch,
                         int* overflow, void* stream);
int clava_split_unpack_f32(const void* src, long long pitch, long long rows, int cols, int shift, float* dst, long long ld,
                           void* stream);
/* packs the TRANSPOSE of a stored matrix: dst[rows, cols] with element (r, c) = src[c * ld + r] (an nn.Linear weight as the
 * operand of the input-gradient product dy . W, scripts/train.py:37). */
int clava_split_pack_transposed_f32(const float* src, long long ld, long long rows, int cols, int shift, void* dst,
                                    long long pitch, int* overflow, void* stream);

/* One launch for a list of matrices, both orientations: descs_dev[i] (DEVICE array of n descriptors) names a float32
 * matrix src[rows, cols] (leading dimension ld), the factor scale = 2^shift, and where its split rows go: dst
 * [rows, cols] (pitch) and / or dst_t = the split rows of the TRANSPOSE [cols, rows] (pitch_t); either may be NULL.
 * max_tiles >= the largest ceil(rows/64) * ceil(cols/64) of the list.  The training step re-packs every hidden weight
 * (forward operand) and its transpose (input-gradient operand, scripts/train.py:37) after each optimizer step. */
typedef struct {
    const float* src;
    long long ld;
    int rows, cols;
    float scale;
    void* dst;
    long long pitch;
    void* dst_t;
    long long pitch_t;
} clava_pack_desc;
int clava_split_pack_multi_f32(const clava_pack_desc* descs_dev, int n, int max_tiles, int* overflow, void* stream);

/* y[M,N] = alpha * act'(act_out)[M,N] * act( x[M,K] . w[N,K]^T + bias[N] + rowadd[N] )
 * x, w: split rows with shifts shift_x, shift_w; y_split (shift_y) and/or y_f32 (ldy) receive the result; act_out
 * (split rows [M,N], any shift; or NULL) supplies the activation-derivative factor (1 where > 0 else `slope`) of the
 * input-gradient chain; rowadd is the broadcast time-embedding row.  tile_config < 0 picks the tile for the shape
 * (0 .. clava_gemm_f16x3_config_count_()-1 force one: tools/bench_f16x3.py). */
int clava_gemm_f16x3(const void* x, long long pitch_x, int shift_x, const void* w, long long pitch_w, int shift_w,
                     const float* bias, const float* rowadd, int act, const void* act_out, long long pitch_act, float slope,
                     float alpha, void* y_split, long long pitch_y, int shift_y, float* y_f32, int ldy, int M, int N, int K,
                     int* overflow, int tile_config, void* stream);
int clava_gemm_f16x3_config_count_(void);

/* dst[cols, rows] = src[rows, cols]^T, both split rows (the float16 planes are moved, not re-rounded; the shift is the
 * source's).  Pad rows / columns of dst are zeros.  Pitches are multiples of 16 bytes. */
int clava_split_transpose(const void* src, long long pitch_src, long long rows, int cols, void* dst, long long pitch_dst,
                          void* stream);

/* clava_gemm_wgrad_f32 on the f16 matrix cores: dw_slabs[s*slab_stride + n*lddw + k] = partial_s( dy[M,N_out]^T . x[M,K_in] ).
 * dyt = dy^T as split rows [N_out, M] (shift_dy), xt = x^T as split rows [K_in, M] (shift_x): the batch is the contraction
 * axis, cut into `splits` ranges of whole 32-row blocks (every range non-empty).  Replaces the weight-gradient half of
 * loss.backward() (scripts/train.py:37) for the hidden blocks. */
int clava_gemm_wgrad_f16x3(const void* dyt, long long pitch_dyt, int shift_dy, const void* xt, long long pitch_xt, int shift_x,
                           float* dw_slabs, int lddw, long long slab_stride, int M, int N_out, int K_in, int splits,
                           int tile_config, void* stream);

/* The same weight-gradient slabs from the batch-major operands themselves: dy[M, N_out] (shift_dy) and x[M, K_in]
 * (shift_x) as the forward / input-gradient kernels wrote them; no transposed copies (the kernel reads its LDS tiles
 * transposed, ds_read_b64_tr_b16).  M must be a multiple of 32 (otherwise use clava_gemm_wgrad_f16x3).  db_slabs
 * (optional) receives the bias-gradient slabs db_slabs[s*slab_stride + n] = partial_s( sum_b dy[b, n] ) from the same
 * launch (clava_colsum_slabs_split otherwise). */
int clava_gemm_wgrad_f16x3_tn(const void* dy, long long pitch_dy, int shift_dy, const void* x, long long pitch_x, int shift_x,
                              float* dw_slabs, int lddw, long long slab_stride, float* db_slabs, int M, int N_out, int K_in,
                              int splits, void* stream);

/* ---- schedule tables ---------------------------------------------------------------------------------------
 * Device-resident tables of length T, built by the host exactly as GaussianMultinomialDiffusion.__init__ builds
 * them (tab_ddpm/gaussian_multinomial_diffsuion.py:105-156); model_variance is the "fixed-large" variance the
 * reference rebuilds every step at :209-211 and model_sigma = exp(0.5*log(model_variance)) (:382), both float64. */
typedef struct clava_schedule {
    int T;
    const float* sqrt_recip_alphas_cumprod;
    const float* sqrt_recipm1_alphas_cumprod;
    const float* posterior_mean_coef1;
    const float* posterior_mean_coef2;
    const double* model_variance;
    const double* model_sigma;
    const float* log_alpha;
    const float* log_1_min_alpha;
    const float* log_cumprod_alpha;
    const float* log_1_min_cumprod_alpha;
    const float* sqrt_alphas_cumprod;
    const float* sqrt_one_minus_alphas_cumprod;
} clava_schedule;

/* ---- fused diffusion steps ---------------------------------------------------------------------------------
 * One Gaussian reverse step on x[B,n_num] (float64 chain state, updated in place):
 *   x <- mean(eps, x, t) [+ variance_t * grad] + 1[t != 0] * sigma_t * z
 * Replaces gaussian_p_mean_variance + condition_mean + gaussian_p_sample (:200-237, :315-328, :354-383).
 * eps = numeric columns of the denoiser output (float32, leading dimension ld_eps); grad = cond_fn output or NULL;
 * noise = injected z [B,n_num] float64, or NULL for the counter-based generator keyed by
 * (seed, step_index, row_offset + row, column); t_rows = per-row timesteps (int64) or NULL to use `t` for every
 * row; x_is_f32 != 0 reproduces the float32 arithmetic of a chain's first step (x is then the float32 draw of
 * :1029).  x_in (optional) receives float(x_next) for the next denoiser call; pred_xstart is optional. */
int clava_gauss_step_f32(const clava_schedule* sched, int B, int n_num, const float* eps, int ld_eps, double* x,
                         const float* grad, int ld_grad, const double* noise, float* x_in, int ld_xin,
                         double* pred_xstart, const long long* t_rows, int t, int x_is_f32, unsigned long long seed,
                         unsigned long long row_offset, unsigned step_index, void* stream);

/* One categorical step on index state idx[B,m] (m features, feature f owns classes offsets[f]..offsets[f+1]).
 *   mode 0 (reverse): idx_out ~ p(x_{t-1} | x_t = idx_in, logits)   = p_sample (:468-472): predict_start (:414-424),
 *                     q_posterior (:426-455), log_sample_categorical (:506-516)
 *   mode 1 (init):    idx_out ~ uniform categorical                  = the initial draw of `sample` (:1033-1035)
 *   mode 2 (q_sample):idx_out ~ q(x_t | x_0 = idx_in)                = q_sample (:518-523)
 * uniforms = injected U[0,1) draws [B,sumK] (feature blocks side by side, the reference's draw order) or NULL for
 * the counter-based generator.  x_in (optional) receives the log-one-hot of idx_out (index_to_log_onehot,
 * tab_ddpm/utils.py:127-134); log_prob (optional, mode 0) receives the normalised log p(x_{t-1}|x_t) of p_pred
 * (:457-466).  log_k[f] = log(float(K_f)) in float32.  idx_in may alias idx_out. */
int clava_multinomial_step_f32(const clava_schedule* sched, int mode, int B, int m, int sumK, const int* offsets,
                               const float* log_k, const float* logits, int ld_logits, const int* idx_in,
                               int* idx_out, const float* uniforms, int ld_u, float* x_in, int ld_xin,
                               float* log_prob, int ld_lp, const long long* t_rows, int t, unsigned long long seed,
                               unsigned long long row_offset, unsigned step_index, void* stream);

/* clava_multinomial_step_f32 whose generator key also takes a DEVICE word: the draws are keyed by
 * step_index + *step_counter (step_counter may be NULL).  A training step captured as a HIP graph keeps seed and
 * step_index as baked-in arguments and advances the word on the device, so every replay draws fresh uniforms for
 * q_sample (:518-523) without a host-side argument update. */
int clava_multinomial_step_counter_f32(const clava_schedule* sched, int mode, int B, int m, int sumK, const int* offsets,
                                       const float* log_k, const float* logits, int ld_logits, const int* idx_in,
                                       int* idx_out, const float* uniforms, int ld_u, float* x_in, int ld_xin,
                                       float* log_prob, int ld_lp, const long long* t_rows, int t, unsigned long long seed,
                                       unsigned long long row_offset, unsigned step_index, const unsigned* step_counter,
                                       void* stream);

/* Sampler of the reverse step's categorical draws when the kernel draws its own noise (uniforms == NULL, log_prob == NULL):
 * 0 = Gumbel arg-max over the K posterior log-probabilities, the reference's construction (log_sample_categorical,
 * tab_ddpm/gaussian_multinomial_diffsuion.py:506-516, called from p_sample :468-472); 1 = inverse CDF of the same posterior
 * p(x_{t-1} | x_t, x0_hat) (:426-455) evaluated in probability space, one uniform per (row, feature) block (default; 2 is
 * accepted as a synonym; same distribution, other draws).  Injected uniforms always take the Gumbel form.  Returns the
 * previous setting; any other
 * `mode` only queries.  Process-wide; CLAVA_CAT_SAMPLER=gumbel|icdf sets the initial value. */
int clava_categorical_sampler_set(int mode);

/* x_t[B,n_num] = sqrt_alphas_cumprod[t] * x0 + sqrt_one_minus_alphas_cumprod[t] * noise, per-row t (int64):
 * gaussian_q_sample (:172-180), float32. */
int clava_gauss_qsample_f32(const clava_schedule* sched, int B, int n_num, const float* x0, const float* noise,
                            const long long* t_rows, float* x_t, int ld_xt, void* stream);

/* One training step's draws and forward noising in one launch (mixed_loss, :644-655): t_rows[B] (int64) uniform on
 * 0..T-1 (torch.randint of :590), noise[B,n_num] standard normal (torch.randn_like of :647), x_t = gaussian_q_sample of
 * x0 (row stride ldx0).  The counter-based generator is keyed by (seed, step_index + *step_counter, row / element):
 * a captured step replays with fresh draws when the caller advances the device word `step_counter` (may be NULL).
 * n_num = 0 draws t only. */
int clava_train_draw_qsample_f32(const clava_schedule* sched, int B, int n_num, const float* x0, long long ldx0,
                                 unsigned long long seed, unsigned step_index, const unsigned* step_counter, long long* t_rows,
                                 float* noise, float* x_t, int ld_xt, void* stream);

/* The same launch for a table with categorical columns: x0 rows hold [n_num numeric | m class codes as floats] (the
 * training loop's `x`, :644-652); also writes idx0[B,m] (`x_cat.long()`) and idxt[B,m], the classes of
 * x_t ~ q(x_t | x_0) (q_sample, :494-498).  The draw is the reference's Gumbel arg-max over q_pred's K log-probabilities
 * (:403-412, :506-516) in closed form: those take two values (the clean class / any other), so x_t is the clean class with
 * probability e^lh / (e^lh + (K-1) e^lc) and uniform over the others otherwise -- the same distribution from two uniforms
 * per (row, feature) instead of K.  clava_multinomial_step_counter_f32 (mode 2) remains the per-class Gumbel form. */
int clava_train_draw_mixed_f32(const clava_schedule* sched, int B, int n_num, const float* x0, long long ldx0, int m, const int* offsets,
                               const float* log_k, unsigned long long seed, unsigned step_index, const unsigned* step_counter,
                               long long* t_rows, float* noise, float* x_t, int ld_xt, int* idx0, int* idxt, void* stream);

/* x_in[b, offsets[f] + k] = 0 at k = idx[b,f], log(1e-30) elsewhere: index_to_log_onehot (tab_ddpm/utils.py:127-134) of
 * class indices [B,m], the categorical columns of the denoiser input (row stride ld, in floats). */
int clava_cat_onehot_image_f32(int B, int m, int sumK, const int* offsets, const int* idx, float* x_in, long long ld, void* stream);

/* x[B,n_num] (float64) <- standard normal draws, generated in float32 like torch.randn of :1029 / :990, from the
 * counter-based generator keyed by (seed, row_offset + row, column). */
int clava_gauss_init_f64(double* x, int B, int n_num, unsigned long long seed, unsigned long long row_offset,
                         void* stream);

/* ---- layer stacks --------------------------------------------------------------------------------------------
 * A stack of Linear layers: layer l maps dims[l] -> dims[l+1] with weight w[l] ([dims[l+1], dims[l]], row-major),
 * bias b[l], activation act[l]; bn_scale[l]/bn_shift[l] (entries or the arrays themselves may be NULL) are an
 * eval-mode BatchNorm1d applied to layer l's activated output.  All arrays are HOST arrays; w/b/bn_* entries are
 * device pointers.  MLPDiffusion (tab_ddpm/modules.py:425-460) = proj, 6 ReLU blocks, head; Classifier
 * (pipeline_utils.py:584-630) = proj, Linear+ReLU+BN, 6 x Linear+LeakyReLU+BN, Linear. */
typedef struct clava_net {
    int n_layers;
    const int* dims;
    const float* const* w;
    const float* const* b;
    const int* act;
    const float* const* bn_scale;
    const float* const* bn_shift;
} clava_net;

/* outs[l] (host array of device pointers) receives layer l's activated, pre-BatchNorm output [M, dims[l+1]].
 * rowadd0 is added to layer 0's pre-activation (the time embedding of `proj(x) + emb`, modules.py:459;
 * pipeline_utils.py:627); ld_rowadd0 == 0 broadcasts one row.  Replaces MLPDiffusion.forward / Classifier.forward. */
int clava_net_forward_f32(const clava_net* net, const float* x, int ldx, int M, const float* rowadd0, int ld_rowadd0,
                          float* const* outs, void* stream);

/* Input gradient of the stack: given the forward's `outs` and dlast = d(loss)/d(last output) [M, dims[n_layers]],
 * computes grads[l] = d(loss)/d(pre-activation of layer l-1) for l = n_layers-1..1 and grads[0] = alpha *
 * d(loss)/d(x) [M, dims[0]] (grads = host array of n_layers device pointers, grads[l] is [M, dims[l]]).
 * Replaces the torch.autograd.grad call of cond_fn (pipeline_utils.py:864) below the log-softmax. */
int clava_net_input_grad_f32(const clava_net* net, int M, float* const* outs, const float* dlast,
                             float* const* grads, float alpha, void* stream);

/* dlogits[b,c] = 1[c == y[b]] - softmax(logits[b,:])[c]: gradient of sum_b log_softmax(logits)[b, y_b]
 * (cond_fn, pipeline_utils.py:862-864). */
int clava_logsoftmax_select_grad_f32(const float* logits, int ld, const int* y, float* dlogits, int B, int C,
                                     void* stream);

/* emb[i, 0:dim] = [cos(t_i * freqs), sin(t_i * freqs)] (timestep_embedding, tab_ddpm/modules.py:20-38);
 * freqs [dim/2] float32 is computed by the host (exp in float32 as torch does); t is int64; dim must be even. */
int clava_timestep_embedding_f32(const long long* t, int n, const float* freqs, int dim, float* emb, void* stream);

/* ---- whole reverse chain -----------------------------------------------------------------------------------
 * Runs n_steps reverse steps t = t_start, t_start-1, ... on B rows: denoiser forward, optional classifier guidance
 * (forward + log-softmax-select + input gradient), fused Gaussian step, fused categorical step.  Replaces the
 * loop body of `sample` (:1043-1062) and `conditional_sample` (:995-1014).  Nothing is copied to the host. */
typedef struct clava_chain {
    clava_schedule sched;
    int t_start;                 /* first timestep to process (T-1 for a fresh chain) */
    int n_steps;                 /* number of steps t = t_start, t_start-1, ... to run (T for a full chain) */
    int n_num, m, sumK;          /* numeric columns, categorical features, total classes (0, 0 when none) */
    const int* offsets;          /* device [m+1] */
    const float* log_k;          /* device [m] */
    const clava_net* denoiser;   /* dims[0] == n_num + sumK == dims[n_layers] */
    const float* denoiser_emb;   /* device [T, dims[1]]: time_embed(timestep_embedding(t)) for every t */
    const clava_net* classifier; /* NULL for unguided sampling; dims[0] == n_num */
    const float* classifier_emb; /* device [T, dims[1]] */
    const int* y;                /* device [B] cluster label per row (guidance target) */
    float guidance_scale;        /* classifier_scale (pipeline_utils.py:864) */
    unsigned long long seed, row_offset;
    const double* noise;         /* device [n_steps, B, n_num] injected Gaussian noise, or NULL */
    const float* uniforms;       /* device [n_steps, B, sumK] injected uniforms, or NULL */
    int first_step_f32;          /* x holds float32 values on entry (fresh chain) */
} clava_chain;

/* number of floats of workspace clava_sample_chain_f32 needs for B rows */
long long clava_chain_workspace_floats(const clava_chain* c, int B);

/* x [B,n_num] float64 (in: x at t_start, out: x after n_steps steps), idx [B,m] int32 likewise; ws = device workspace. */
int clava_sample_chain_f32(const clava_chain* c, int B, double* x, int* idx, float* ws, long long ws_floats,
                           void* stream);

/* ---- whole reverse chain on the f16 matrix cores (split-f16 layers, float32-level accuracy) --------------------
 * A Linear stack prepared for clava_gemm_f16x3: w[l] = layer l's weight as split rows [dims[l+1], dims[l]] with
 * shift shift_w[l]; an eval-mode BatchNorm below a layer (pipeline_utils.py:609, 616) is folded into that layer's
 * weight and bias by the host (W' = W diag(scale), b' = b + W shift).  wt[l] = the same matrix transposed (split rows
 * [dims[l], dims[l+1]], same shift) for the input-gradient chain; wt and shift_grad may be NULL for a stack that is
 * only run forward.  shift_in / shift_out[l] / shift_grad[l] / shift_dlast are the power-of-two scales of the input,
 * of layer l's activation planes, of the gradient planes d/d(pre-activation of layer l-1) and of the d/d(logits)
 * planes, chosen by the host from a float32 calibration pass so that magnitudes sit near 2^6.  All arrays are HOST
 * arrays of device pointers / ints. */
typedef struct clava_split_net {
    int n_layers;
    const int* dims;
    const void* const* w;
    const void* const* wt;
    const float* const* b;
    const int* act;
    const int* shift_w;
    int shift_in;
    const int* shift_out;
    const int* shift_grad;
    int shift_dlast;
    /* Optional operands of the fused narrow-stack kernels (csrc/fused_narrow.hip; NULL: the chain runs those layers as
     * separate products).  Each fs_* is ONE device buffer: the weights of all layers of a fused kernel as ready-made
     * v_mfma_f32_32x32x16_f16 fragments in the order the kernel consumes them - per layer and group of <= 4 row tiles
     * [16-deep k-step][row tile of 32][hi, lo][lane][8 halves], element j of lane half h of step s holding column
     * 16 s + 8 (j >> 2) + 4 h + (j & 3), zero padded to whole 32 KB chunks.  Denoiser: fs_front = proj, first hidden block;
     * fs_head = head.  Classifier: fs_front = proj, blocks 1-2; fs_tail = block 7, logits layer, their transposes;
     * fs_back = transposes of block 2, block 1, proj (the derivative of block 1 comes as bits the front kernel stored).
     * bf[l] = b[l] zero padded to a multiple of 32 floats (64 for the logits layer), host array of n_layers device pointers.
     * clavaddpm_amd/split_nets.py (SplitNet, pack_fused / fused_stream) builds them; CLAVA_FUSED_PARTS selects which of the
     * kernels a chain uses (default: the classifier's three). */
    const void* fs_front;
    const void* fs_tail;
    const void* fs_back;
    const void* fs_head;
    const float* const* bf;
    /* Optional (denoiser of a table with categorical columns; NULL: layer 0 runs as a dense product over the
     * [B, n_num + sum K] input): layer 0 = MLPDiffusion.proj (tab_ddpm/modules.py:459) evaluated from the numeric columns
     * and the category INDICES - the log-one-hot columns (tab_ddpm/utils.py:127-134) are 0 at the hot class and
     * log_tiny = log(1e-30) elsewhere, so proj(x) = W_num x + cat_bias - log_tiny * sum_f W[:, n_num + off_f + idx_f].
     * num_wt = numeric columns of w[0] transposed, float32 [n_num][dims[1]]; cat_wt = categorical columns transposed,
     * float32 [sum K][dims[1]]; cat_bias[j] = b[0][j] + log_tiny * sum_c W[j][n_num + c] (device arrays; dims[1] a multiple
     * of 128).  Used by clava_sample_chain_f16x3 when no classifier is given; the step kernels then write no input image. */
    const float* num_wt;
    const float* cat_wt;
    const float* cat_bias;
    float log_tiny;
} clava_split_net;

/* Same chain as clava_sample_chain_f32 (same step kernels, same noise keys) with every nn.Linear product on
 * clava_gemm_f16x3; c->denoiser / c->classifier are ignored (c->classifier_emb, c->y, c->guidance_scale are used when
 * `classifier` is not NULL).  *overflow (device int, zeroed by the caller) becomes 1 if any intermediate left the
 * float16 range: the result is then invalid and the caller reruns the chain with clava_sample_chain_f32. */
long long clava_chain_workspace_bytes_f16x3(const clava_chain* c, const clava_split_net* denoiser,
                                            const clava_split_net* classifier, int B);
int clava_sample_chain_f16x3(const clava_chain* c, const clava_split_net* denoiser, const clava_split_net* classifier, int B,
                             double* x, int* idx, void* ws, long long ws_bytes, int* overflow, void* stream);

/* ---- training step ------------------------------------------------------------------------------------------
 * Gaussian 'mse' loss and its gradient: loss_rows[b] = mean_c (noise - model_out)^2, d_out[b,c] = 2 (model_out -
 * noise) / (n_num * B), i.e. the gradient of loss_rows.mean().  Replaces _gaussian_loss (:282-300, 'mse' branch) and
 * its autograd backward. */
int clava_gauss_mse_loss_f32(int B, int n_num, const float* model_out, int ld_out, const float* noise, float* loss_rows,
                             float* d_out, int ld_d, void* stream);

/* Multinomial variational loss per (row, feature) block and its gradient w.r.t. the categorical logits:
 *   loss_terms[b,f] = (t_b == 0 ? decoder NLL : KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t))) / pt + KL prior
 *   d_logits       = gradient of mean_b( sum_f loss_terms[b,f] / m )
 * idx0 / idxt are the clean and noised category indices [B,m] (int32), pt = 1/T as float32.  Replaces
 * _multinomial_loss 'vb_stochastic' = compute_Lt + kl_prior (:544-572, :597-607), multinomial_kl (:387-389),
 * log_categorical (tab_ddpm/utils.py:124-125) and their autograd backward. */
int clava_multinomial_loss_f32(const clava_schedule* sched, int B, int m, int sumK, const int* offsets, const float* log_k,
                               const float* logits, int ld_logits, const int* idx0, const int* idxt,
                               const long long* t_rows, float pt, float* loss_terms, float* d_logits, int ld_d,
                               void* stream);

/* The two loss kernels with the gradient ALSO (or, with d_out / d_logits null, ONLY) written as split rows -- the operand
 * of the head's backward products (clava_gemm_f16x3, clava_gemm_wgrad_f16x3_tn) at scale 2^shift, 128 bytes per 32-column
 * block as clava_split_pack_f32 writes them; `col0` = column of the first categorical column inside those rows (n_num).
 * The same loss.backward() down to the denoiser output (scripts/train.py:37) without the float32 image and the pack
 * launch between it and the products; columns the two kernels do not own (the pad beyond n_num + sumK) are left as they
 * are (the caller zeroes the rows once).  *overflow is set to 1 when a scaled magnitude leaves the float16 range. */
int clava_gauss_mse_loss_split_f32(int B, int n_num, const float* model_out, int ld_out, const float* noise, float* loss_rows,
                                   float* d_out, int ld_d, void* d_split, long long pitch, int shift, int* overflow, void* stream);
int clava_multinomial_loss_split_f32(const clava_schedule* sched, int B, int m, int sumK, const int* offsets, const float* log_k,
                                     const float* logits, int ld_logits, const int* idx0, const int* idxt, const long long* t_rows,
                                     float pt, float* loss_terms, float* d_logits, int ld_d, void* d_split, long long pitch, int shift,
                                     int col0, int* overflow, void* stream);

/* out[0] = scale * sum(v[0..n)), fixed summation order (the .mean() of the loss terms, :675). */
int clava_sum_f32(const float* v, long long n, float scale, float* out, void* stream);

/* db_slabs[s][0..N) = column sums of dy over the s-th of `splits` batch ranges (bias gradients; same slab scheme as
 * clava_gemm_wgrad_f32). */
int clava_colsum_slabs_f32(const float* dy, int lddy, int M, int N, int splits, float* db_slabs, long long slab_stride,
                           void* stream);
/* the same column sums over split rows (dy as float16 planes with shift `shift`, pitch bytes per row): the bias gradient
 * straight from the planes the split-f16 input-gradient chain produces, no float32 copy of dy. */
int clava_colsum_slabs_split(const void* dy_split, long long pitch, int shift, int M, int N, int splits, float* db_slabs,
                             long long slab_stride, void* stream);

/* out[i] = scale * sum_s slabs[s*slab_stride + i], fixed order (flat gradient for autograd / the RCCL all-reduce). */
int clava_slab_reduce_f32(const float* slabs, int splits, long long slab_stride, long long n, float scale, float* out,
                          void* stream);

/* SiLU and its backward for the time-embedding branch (nn.SiLU in time_embed, tab_ddpm/modules.py:445-449):
 * a = z * sigmoid(z);  dz = da * sigmoid(z) * (1 + z * (1 - sigmoid(z))). */
int clava_silu_f32(const float* z, float* a, long long n, void* stream);
int clava_silu_backward_f32(const float* z, const float* da, float* dz, long long n, void* stream);

/* Front of MLPDiffusion.forward in training (per-row timesteps) as one launch: te = timestep_embedding(t)
 * (tab_ddpm/modules.py:20-38, freqs = the dim_t/2 frequencies as float32), z1 = te W0^T + b0, a1 = SiLU(z1),
 * emb = a1 W2^T + b2 (time_embed, modules.py:445-449, 452), h = x_in Wp^T + bp + emb (modules.py:459).  te, z1, a1
 * ([B, dim_t] each) are stored for the backward; h goes to `h` (float32 [B, dim_t], may be NULL) and / or to `h_split`
 * (split rows with `shift`, `pitch` bytes per row, may be NULL; *overflow as in clava_split_pack_f32).  dim_t must be 128
 * (every config's value); other widths use the separate kernels.  d_in = 0 leaves proj out: h = emb (tables whose first
 * layer runs in gather form, clava_cat_input_f32, which adds its per-row embedding). */
int clava_train_front_f32(const long long* t, int B, const float* freqs, int dim_t, const float* x_in, long long ldx, int d_in,
                          const float* w0, const float* b0, const float* w2, const float* b2, const float* wp, const float* bp,
                          float* te, float* z1, float* a1, float* h, void* h_split, long long pitch, int shift, int* overflow,
                          void* stream);

/* Fused optimizer step over n contiguous parameters: gradient = grad_scale * sum_s grad_slabs[s*slab_stride + i]
 * (fixed order), then torch.optim.AdamW's update (decoupled decay, bias-corrected moments; scripts/train.py:17, 38)
 * and, if ema != NULL, update_ema: ema = ema*ema_rate + param*(1-ema_rate) (scripts/utils_train.py:20-29).
 * `step` is the 1-based optimizer step count. */
int clava_adamw_ema_f32(float* param, float* ema, float* exp_avg, float* exp_avg_sq, const float* grad_slabs, int splits,
                        long long slab_stride, float grad_scale, long long n, double lr, double beta1, double beta2,
                        double eps, double weight_decay, long long step, double ema_rate, void* stream);

/* d_z1 = SiLU'(z1) * (g W2): the backward from the time embedding (g = d loss / d emb, [B, dim_t]) through time_embed[2] and the
 * SiLU in one launch (autograd of tab_ddpm/modules.py:445-449); W2 is time_embed[2].weight [dim_t, dim_t]; dim_t must be 128. */
int clava_train_time_bwd_f32(const float* g, long long ldg, int B, int dim_t, const float* w2, const float* z1, float* d_z1,
                             void* stream);

/* ---- narrow end layers of the denoiser in training (csrc/narrow.hip): D <= 16 columns on one side -----------------
 * Memory-bound replacements of the tiled products for the head Linear(K, D) and proj Linear(D, J) of numeric tables.
 * clava_narrow_head_f32: out[b][d] = bias[d] + sum_k h[b][k] W[d][k] (tab_ddpm/modules.py:460); K % 4 == 0, rows of h
 *   16-byte aligned, D * K <= 16384.
 * clava_narrow_dgrad_f32: g[b][j] = act'(act[b][j]) * sum_d dy[b][d] W[d][j] (W is [D][J]; act may be NULL = no mask,
 *   else 1 where act > 0 and `slope` elsewhere), written as float32 (g_f32, may be NULL) and / or as split rows
 *   (g_split with shift / pitch as clava_split_pack_f32, pad columns zero, may be NULL): the input gradient of the head
 *   with the last block's ReLU' and the split pack folded in (backward of modules.py:460, :174-180).
 * clava_narrow_wgrad_f32: per batch range s (rows split as clava_colsum_slabs_f32 splits them),
 *   dw_slabs[s*slab_stride + d*stride_d + j*stride_j] = sum_b A[b][d] Bm[b][j]   (A [B, D] narrow, Bm [B, J]);
 *   db_narrow_slabs[s*slab_stride + d] = sum_b A[b][d], db_wide_slabs[s*slab_stride + j] = sum_b Bm[b][j] (either may
 *   be NULL).  Head: A = dy, Bm = h, strides (K, 1), bias = narrow sums.  proj: A = x, Bm = g, strides (1, D), bias =
 *   wide sums (weight half of loss.backward(), scripts/train.py:37). */
int clava_narrow_head_f32(const float* h, long long ldh, int B, int K, const float* W, const float* bias, int D, float* out, int ld_out,
                          void* stream);
int clava_narrow_dgrad_f32(const float* dy, int ld_dy, int B, int D, const float* W, int J, const float* act, long long ld_act,
                           float slope, float* g_f32, long long ldg, void* g_split, long long pitch, int shift, int* overflow,
                           void* stream);
int clava_narrow_wgrad_f32(const float* A, long long lda, int D, const float* Bm, long long ldb, int J, int B, int splits,
                           float* dw_slabs, long long stride_d, long long stride_j, float* db_narrow_slabs, float* db_wide_slabs,
                           long long slab_stride, void* stream);

/* dst[0..n) = (*word & 1 ? host1 : host0)[0..n): input fetch of a captured training step.  host0 / host1 are pinned,
 * device-visible HOST buffers (hipHostMalloc / torch pin_memory) that the caller fills alternately, by the parity of the
 * device step word `word`; the kernel reads them over the bus, so no host-synchronous copy separates two replays.  The
 * caller must not refill a buffer before the replay that reads it has finished (an event two steps back). */
int clava_stage_fetch_f32(const float* host0, const float* host1, const unsigned* word, float* dst, long long n, void* stream);

/* ---- first layer of a table with categorical columns in TRAINING, without the [B, sum K] log-one-hot image -------------
 * clava_cat_input_f32: h = W_num x_num + (b + L S) + emb - L sum_f W[:, n_num + off_f + idx_f] (csrc/cat_input.hip; proj of
 *   tab_ddpm/modules.py:459 on the input `index_to_log_onehot` builds, utils.py:127-134), from the TRANSPOSED weight
 *   (num_wt [n_num][d1], cat_wt [sum K][d1]) and cat_bias = b + L * row sums of the categorical columns; emb is one
 *   broadcast row (ld_emb = 0) or a row per batch row; output as split rows and optionally float32.  d1 % 128 == 0.
 * clava_cat_wgrad_f32: the categorical columns of that layer's weight gradient per batch range,
 *   dw_slabs[s][c][col0 + off_f + k] = L * (sum_b g[b][c] - sum_{b: idx[b][f] = k} g[b][c]) (csrc/cat_wgrad.hip; weight half
 *   of loss.backward(), scripts/train.py:37), no atomics; g_colsum_slabs[s * slab_stride + c] = sum_b g[b][c] over batch range s (the
 *   layer's bias-gradient slabs, computed first on the same stream).  wg_feature / wg_k0 (DEVICE arrays of n_workgroups ints) assign a
 *   (feature, first class) to every workgroup: clava_cat_wgrad_workgroups counts them from the HOST offsets,
 *   clava_cat_wgrad_plan fills host arrays the caller uploads once per table.  d1 must be 128. */
int clava_cat_input_f32(const float* xnum, int ld_x, const int* idx, int B, int n_num, int m, int d1, const int* offsets,
                        const float* num_wt, const float* cat_wt, const float* cat_bias, const float* emb, long long ld_emb,
                        float log_tiny, int act, void* out_split, long long pitch, int shift, float* out_f32, long long ld_f32,
                        int* overflow, void* stream);

/* The weight operands of clava_cat_input_f32 from the parameters of MLPDiffusion.proj (tab_ddpm/modules.py:444, applied at
 * :459; training: rebuilt every step, after scripts/train.py:38 moved them): wt[D][128] = w^T and
 * cat_bias[c] = b[c] + log_tiny * sum_{k >= n_num} w[c][k] (float64 sums in a fixed order).  `partial` is scratch of
 * clava_cat_input_weights_tiles(D) * 128 doubles. */
int clava_cat_input_weights_tiles(int D);
int clava_cat_input_weights_f32(const float* w, long long ldw, int d1, int D, int n_num, const float* b, double log_tiny,
                                float* wt, double* partial, float* cat_bias, void* stream);
int clava_cat_wgrad_workgroups(const int* offsets_host, int m);
int clava_cat_wgrad_plan(const int* offsets_host, int m, int* wg_feature_host, int* wg_k0_host);
int clava_cat_wgrad_f32(const float* g, long long ldg, const float* g_colsum_slabs, int B, int d1, const int* idx, int m,
                        const int* offsets, const int* wg_feature, const int* wg_k0, int n_workgroups, int splits, float log_tiny,
                        float* dw_slabs, long long ldw, int col0, long long slab_stride, void* stream);

/* The same update with the step-dependent scalars read from device memory, for a step captured in a hipGraph (kernel
 * arguments are frozen at capture): clava_adamw_hyper_f32 fills hyper8[0..7] on the HOST with exactly the values
 * clava_adamw_ema_f32 derives from (lr, betas, eps, weight_decay, step); the caller copies them to the device before each
 * replay and passes that device pointer as hyper_dev. */
int clava_adamw_hyper_f32(double lr, double beta1, double beta2, double eps, double weight_decay, long long step, float* hyper8);
int clava_adamw_ema_hyper_f32(float* param, float* ema, float* exp_avg, float* exp_avg_sq, const float* grad_slabs, int splits,
                              long long slab_stride, float grad_scale, long long n, const float* hyper_dev, double ema_rate,
                              void* stream);

/* ---- guidance-classifier training (train_classifier / numerical_forward_backward_log) --------
 * One hidden block of Classifier (pipeline_utils.py:607-617) in train mode is Linear -> ReLU/LeakyReLU -> BatchNorm1d
 * -> Dropout.  The Linear + activation is clava_gemm_bias_act_f32; these entry points are the rest.
 *
 * clava_bn_dropout_train_f32: h = dropout(BatchNorm_train(a)) for a[B,C] (the activation output).  Batch mean and
 * biased variance per channel (nn.BatchNorm1d in training mode, eps inside the square root), running statistics
 * updated in place with `momentum` and the unbiased variance (pass NULL for both to skip), save_mean / save_invstd [C]
 * kept for the backward.  Dropout keeps an element with probability 1-p and scales kept ones by 1/(1-p) (nn.Dropout);
 * the keep decision comes from keep_mask[B,C] bytes (parity tests) or, when NULL, from the counter-based generator
 * keyed by (seed, step, layer < 32, row, column); when step_dev != NULL the step index is read from that DEVICE word
 * instead, so that a captured graph can advance it.  workspace holds clava_bn_workspace_floats(C) floats. */
long long clava_bn_workspace_floats(int C);
int clava_bn_dropout_train_f32(const float* a, int lda, int B, int C, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, float p, const unsigned char* keep_mask,
                               int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev, unsigned int layer,
                               float* workspace,
                               float* save_mean, float* save_invstd, float* h, int ldh, void* stream);
/* Backward of the block above, from dh (gradient w.r.t. the dropout output) to dz (gradient w.r.t. the Linear output):
 * dy = dh * keep/(1-p); dbeta = sum_b dy, dgamma = sum_b dy*xhat (summed in a fixed order through `workspace`,
 * clava_bn_workspace_floats(C) floats) land in slab 0 of dbeta_slabs / dgamma_slabs, slabs 1..splits-1 (at
 * + s*slab_stride, the layout clava_adamw_ema_f32 sums) are zeroed; dz = act'(a) * gamma*invstd *
 * (dy - dbeta/B - xhat*dgamma/B) with act'(a) = a > 0 ? 1 : slope (act = CLAVA_ACT_NONE: no activation factor).
 * Replaces autograd's backward through model.{j+1..j+3} of the reference Classifier. */
int clava_bn_dropout_backward_f32(const float* dh, int lddh, const float* a, int lda, int B, int C, const float* save_mean,
                                  const float* save_invstd, const float* gamma, float p, const unsigned char* keep_mask,
                                  int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev, unsigned int layer,
                                  int act, float slope, float* workspace,
                                  int splits, float* dgamma_slabs, float* dbeta_slabs, long long slab_stride, float* dz,
                                  int lddz, void* stream);
/* The two entry points above with a second copy of their output as split rows (the number format of clava_gemm_f16x3:
 * planes scaled by 2^split_shift, rows split_pitch bytes apart, *overflow raised when a scaled magnitude leaves the float16
 * range), so that the wide blocks' split-f16 products (pipeline_utils.py:607-617 forward; autograd's backward through them)
 * read their operands without a packing launch in between.  h_split / dz_split == NULL: exactly the calls above. */
int clava_bn_dropout_train_split_f32(const float* a, int lda, int B, int C, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float momentum, float eps, float p, const unsigned char* keep_mask,
                                     int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev,
                                     unsigned int layer, float* workspace, float* save_mean, float* save_invstd, float* h, int ldh,
                                     void* h_split, long long split_pitch, int split_shift, int* overflow, void* stream);
int clava_bn_dropout_backward_split_f32(const float* dh, int lddh, const float* a, int lda, int B, int C, const float* save_mean,
                                        const float* save_invstd, const float* gamma, float p, const unsigned char* keep_mask,
                                        int ldmask, unsigned long long seed, unsigned int step, const unsigned int* step_dev,
                                        unsigned int layer, int act, float slope, float* workspace, int splits, float* dgamma_slabs,
                                        float* dbeta_slabs, long long slab_stride, float* dz, int lddz, void* dz_split,
                                        long long split_pitch, int split_shift, int* overflow, void* stream);
/* F.cross_entropy(logits, labels, reduction='none') per row (pipeline_utils.py:683) with the row's logsumexp and the
 * acc@1 / acc@5 indicators of compute_top_k (pipeline_utils.py:640-645; either may be NULL). */
int clava_cross_entropy_rows_f32(const float* logits, int ld, const long long* labels, int B, int C, float* loss_rows, float* acc1,
                                 float* acc5, float* lse, void* stream);
/* dlogits = factor * scale[0] * (softmax(logits) - onehot(labels)).  `scale` is a DEVICE scalar or NULL: the reference
 * back-propagates loss.backward(loss) (pipeline_utils.py:700), i.e. the batch-mean loss multiplies its own gradient;
 * passing the device-resident mean keeps the step free of host synchronisation. */
int clava_cross_entropy_grad_f32(const float* logits, int ld, const long long* labels, const float* lse, int B, int C,
                                 const float* scale, float factor, float* dlogits, int ldd, void* stream);

/* ---- GMM E-step (parent-child clustering) ----------------------------------------------------------------
 * Diagonal-covariance mixture E-step in float64 for X[N,d] (row-major, leading dimension ldx) and k <= 64 components:
 *   log_prob_norm[N], log_resp[N,k] (may be NULL), labels[N] = argmax_c weighted log-probability (may be NULL).
 * Replaces the scikit-learn arithmetic behind GaussianMixture(covariance_type='diag').fit / .predict at
 * pipeline_modules.py:268-277: sklearn/mixture/_gaussian_mixture.py `_estimate_log_gaussian_prob` ('diag' branch)
 * and _base.py `_estimate_weighted_log_prob`, `_estimate_log_prob_resp`, `predict` (scikit-learn 1.4.1.post1 pinned
 * upstream, requirements.txt:199).  weights[k], means[k,d], precisions_chol[k,d] are device arrays; workspace holds
 * clava_gmm_workspace_doubles(k, d) doubles.  Returns -3 when k exceeds 64. */
long long clava_gmm_workspace_doubles(int k, int d);
int clava_gmm_estep_f64(const double* X, long long ldx, long long N, int d, int k, const double* weights,
                        const double* means, const double* precisions_chol, double* workspace, double* log_prob_norm,
                        double* log_resp, int* labels, void* stream);

/* clava_gmm_estep_f64 for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that may
 * not call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it) and therefore cannot
 * provide the workspace either.  Same arguments and results as clava_gmm_estep_f64 minus `workspace`: the derived
 * parameter tables go to a static device array inside the library (clava_gmm_noalloc_max_doubles() doubles = 513.5 KB,
 * held by every GPU the library runs on until the process ends).  Limit: k <= 64 and 2*k*d + 3*k <= that size
 * (d <= 512 at k = 64); beyond it returns -3 with a message.  Calls no hipMalloc / hipFree or other allocating routine.
 * All calls on one GPU share the array: calls that overlap on one GPU (two streams, two host threads) race. */
long long clava_gmm_noalloc_max_doubles(void);
int clava_gmm_estep_noalloc_f64(const double* X, long long ldx, long long N, int d, int k, const double* weights,
                                const double* means, const double* precisions_chol, double* log_prob_norm, double* log_resp,
                                int* labels, void* stream);

/* M-step of the same mixture: scikit-learn's GaussianMixture._m_step for covariance_type='diag'
 * (sklearn/mixture/_gaussian_mixture.py `_estimate_gaussian_parameters`, `_estimate_gaussian_covariances_diag`,
 * `_compute_precision_cholesky`; called once per EM iteration from BaseMixture.fit_predict, _base.py) with
 * resp = exp(log_resp[N,k]):  nk = sum_n resp + 10 eps;  means = resp^T X / nk;
 * covariances = resp^T X^2 / nk - means^2 + reg_covar;  weights = nk / sum(nk);  precisions_chol = 1/sqrt(covariances).
 * stats[0] receives mean_n log_prob_norm[n] (the EM lower bound of `_compute_lower_bound`; 0 when log_prob_norm is
 * NULL) and stats[1] is set to 1 when a covariance is not positive (scikit-learn raises there).  weights[k],
 * means / covariances / precisions_chol [k,d] and stats[2] are device arrays; workspace holds
 * clava_gmm_mstep_workspace_doubles(k, d) doubles.  All sums have a fixed order.  Returns -3 when k exceeds 64. */
long long clava_gmm_mstep_workspace_doubles(int k, int d);
int clava_gmm_mstep_f64(const double* X, long long ldx, long long N, int d, int k, const double* log_resp,
                        const double* log_prob_norm, double reg_covar, double* workspace, double* weights, double* means,
                        double* covariances, double* precisions_chol, double* stats, void* stream);

/* The same M-step in two halves, for mixtures whose rows are sharded over GPUs ("GMM M-step sufficient
 * statistics all-reduce"; scikit-learn's `_estimate_gaussian_parameters` sums over ALL rows).  `partial` writes this
 * shard's sums to totals[clava_gmm_mstep_totals_doubles(k, d)] (layout with KP = round_up(k,16), DP = round_up(d,32):
 * sum_n resp*x at [c*DP + j], sum_n resp*x^2 at [(KP + c)*DP + j], sum_n resp at [2*KP*DP + c], sum_n log_prob_norm at
 * [2*KP*DP + KP]; padding is zero); the caller adds the totals of all shards (one RCCL sum all-reduce) and `finalize`
 * turns them into parameters with N = the row count of ALL shards.  clava_gmm_mstep_f64 is partial + finalize.
 * `workspace` holds 512 * clava_gmm_mstep_totals_doubles(k, d) doubles for `partial`. */
long long clava_gmm_mstep_totals_doubles(int k, int d);
int clava_gmm_mstep_partial_f64(const double* X, long long ldx, long long N, int d, int k, const double* log_resp,
                                const double* log_prob_norm, double* workspace, double* totals, void* stream);
int clava_gmm_mstep_finalize_f64(const double* totals, long long N, int d, int k, double reg_covar, double* weights,
                                 double* means, double* covariances, double* precisions_chol, double* stats, void* stream);

/* ---- multi-parent row matching ------------------------------------------------------------------
 * The search inside match_tables (pipeline_utils.py:1147-1205; there a faiss.IndexIVFFlat with 25 lists and nprobe 1, i.e.
 * an APPROXIMATE search in a third-party library that is not part of the reference tree): here EXACT.  For every query row
 * A[q, :d] the 4 nearest rows of B[nb, d] whose removed[] byte is 0 (removed may be NULL), by squared L2 distance
 * accumulated column by column in float32, ties to the lower row index: out_idx[q*4 + k] (-1 when fewer than k+1 rows are
  * left), out_dist[q*4 + k], ascending.  Any d (tables wider than 64 columns are reduced in 64-column chunks).  workspace: clava_nn_topk_workspace_bytes(nq, nb) bytes. */
long long clava_nn_topk_workspace_bytes(int nq, int nb);
int clava_nn_topk_f32(const float* A, long long lda, int nq, const float* B, long long ldb, int nb, int d,
                      const unsigned char* removed, void* workspace, int* out_idx, float* out_dist, void* stream);

/* ---- k-means++ seeding on the device ------------------------------------------------------------------
 * The N-sized arithmetic of scikit-learn's `_kmeans_plusplus` (sklearn/cluster/_kmeans.py), which seeds
 * GaussianMixture(init_params='k-means++') (pipeline_modules.py:269-277); the random draws stay with the caller.
 *   clava_kmpp_row_norms_f64    xsq[n] = sum_j X[n,j]^2                              (row_norms(X, squared=True))
 *   clava_kmpp_candidates_f64   for the T <= 16 candidate rows candidates[t] (device int64):
 *                               new_closest[t][n] = min(closest[n], max(0, -2 X[n].X[c] + xsq[c] + xsq[n])),
 *                               potentials[t] = sum_n new_closest[t][n]; closest == NULL: no minimum (first centre)
 *                               (`_euclidean_distances(..., squared=True)`, `np.minimum`, `@ sample_weight`)
 *   clava_kmpp_pick_f64         ids[t] = searchsorted(cumsum(closest), rand01[t] * sum(closest)), clipped to N-1;
 *                               *total = sum(closest)        (`stable_cumsum`, `np.searchsorted`, `np.clip`)
 * workspace holds clava_kmpp_workspace_doubles() doubles; every sum has a fixed order.  -3 beyond 16 candidates. */
long long clava_kmpp_workspace_doubles(void);
int clava_kmpp_row_norms_f64(const double* X, long long ldx, long long N, int d, double* xsq, void* stream);
int clava_kmpp_candidates_f64(const double* X, long long ldx, long long N, int d, const double* xsq, const double* closest,
                              const long long* candidates, int T, double* workspace, double* new_closest, double* potentials,
                              void* stream);
int clava_kmpp_pick_f64(const double* closest, long long N, const double* rand01, int T, double* workspace, long long* ids,
                        double* total, void* stream);

/* ---- majority vote of child-row cluster labels per parent group ----------------------------------------
 * The loop of pipeline_modules.py:300-321 and the (cluster, group length) frequency table of :332-340.  Group g owns the
 * rows [group_offsets[g], group_offsets[g+1]) of labels (int32 in [0, n_labels), n_labels <= 1024; groups are
 * contiguous, as `sorted_child_data` is).  group_label[g] = np.argmax(np.bincount(group)) (lowest label among the most
 * frequent), top_count[g] = np.max(np.bincount(group)) (agree rate = top_count / length, divided by the host),
 * assignment[i] = the label of row i's group (np.repeat, :321; may be NULL), size_hist[label * (max_len + 1) + length] += 1
 * per group (int64, zeroed by the caller; may be NULL).  *bad_label_flag (device int, may be NULL) is set to 1 when a
 * label falls outside [0, n_labels). */
int clava_group_vote_i32(const int* labels, const long long* group_offsets, long long n_groups, int n_labels, int* group_label,
                         int* top_count, int* assignment, long long* size_hist, int max_len, int* bad_label_flag, void* stream);

/* ---- post-sampling decoding of numeric columns ------------------------------------------------------
 * out[N,d] = QuantileTransformer(output_distribution='normal').inverse_transform(x[N,d]) in float64: the
 * `dataset.num_transform.inverse_transform(...)` of pipeline_utils.py:913 (transformer built at lib/data.py:217-223;
 * arithmetic: sklearn/preprocessing/_data.py `_transform_col`, inverse branch = scipy norm.cdf, np.interp over
 * (references_, quantiles_[:, col]), the 1e-7 bound clamps).  quantiles is sklearn's quantiles_ [n_quantiles, d]
 * row-major, references its references_ [n_quantiles]; out may alias x.  NaN in, NaN out. */
int clava_quantile_inverse_f64(const double* x, long long ldx, long long N, int d, const double* quantiles, const double* references,
                               int n_quantiles, double* out, long long ldo, void* stream);
/* x[:, col] <- nearest of values[n_values] (sorted unique training values), distance sqrt((x-v)^2), first minimum wins:
 * one column of lib.round_columns (lib/data.py:627-632; scipy cdist + argmin), in place. */
int clava_round_to_values_f64(double* x, long long ldx, long long N, int col, const double* values, int n_values, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLAVA_HIP_H */
