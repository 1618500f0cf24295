// Argument blocks of the fused diffusion-step kernels (diffusion_step.hip), shared with the chain driver.
#pragma once
#include "common.h"

#include "clava_hip.h"
typedef clava_schedule ScheduleDev;

// Optional split-row destination of a step kernel's "next denoiser input" (csrc/split.h: hi = f16(v * scale),
// lo = f16(v * scale - hi); 32-column blocks of 64 B hi | 64 B lo).  base == null: not written.
struct SplitOut {
    char* base = nullptr;
    long long pitch = 0;
    float scale = 1.f;
    int col0 = 0;                      // column of the kernel's first column inside the denoiser input
    int* overflow = nullptr;           // set to 1 when a scaled magnitude leaves the float16 range
};

struct GaussStepArgs {
    ScheduleDev s;
    int B, n_num;
    const float* eps;  int ld_eps;     // model output, numeric columns
    double* x;                         // [B, n_num] chain state, updated in place
    const float* grad; int ld_grad;    // guidance gradient or null
    const double* noise;               // [B, n_num] injected noise or null (Philox)
    float* x_in; int ld_xin;           // float(x_next) written here for the next denoiser call, or null
    SplitOut xs;                       // ... and / or straight into the denoiser's split-row input (split.h format)
    double* pred_xstart;               // optional [B, n_num]
    const long long* t_rows; int t;    // per-row timestep, or null and one t for all rows
    int x_is_f32;                      // state currently holds float32 values: do the float32 arithmetic of the first step
    unsigned long long seed, row_offset;
    unsigned step_index;
};

enum { MULTI_MODE_REVERSE = 0, MULTI_MODE_INIT = 1, MULTI_MODE_QSAMPLE = 2 };

struct MultiStepArgs {
    ScheduleDev s;
    int B, m, sumK, mode;
    const float* logits; int ld_logits;   // model output, categorical columns (REVERSE only)
    const int* idx_in;                    // [B, m] x_t (REVERSE) or x_0 (QSAMPLE); unused for INIT
    int* idx_out;                         // [B, m]
    const float* uniforms; int ld_u;      // [B, sumK] injected uniforms or null (Philox)
    float* x_in; int ld_xin;              // log-one-hot of idx_out for the next denoiser call, or null
    SplitOut xs;                          // ... and / or straight into the denoiser's split-row input (columns col0 + offset + class)
    float* log_prob; int ld_lp;           // optional normalised log p(x_{t-1} | x_t) (REVERSE)
    const int* offsets;                   // device [m + 1]
    const float* log_k;                   // device [m]: log(K_f) in float32
    const long long* t_rows; int t;
    unsigned long long seed, row_offset;
    unsigned step_index;
    const unsigned* step_counter = nullptr;   // optional device word added to step_index (a graph replay draws afresh)
    int icdf = 0;                             // REVERSE with in-kernel draws: sample by inverse CDF (one uniform per block)
};

// first denoiser layer from (numeric columns, category indices): cat_input.hip
struct CatInputArgs {
    const float* xnum; int ld_x;         // [B, n_num] float32 numeric columns of the denoiser input
    const int* idx;                       // [B, m] current categories
    int B, n_num, m, d1;                  // d1 = width of the layer (a multiple of 128)
    const int* offsets;                   // device [m + 1]
    const float* num_wt;                  // [n_num][d1]: numeric columns of the layer's weight, transposed
    const float* cat_wt;                  // [sumK][d1]: categorical columns, transposed (one gathered row per feature)
    const float* cat_bias;                // [d1]: b + log(1e-30) * (row sums of the categorical columns)
    const float* emb;                     // time embedding or null: one broadcast row [d1] (ld_emb = 0, the sampling chain:
    long long ld_emb;                     //   one t per launch) or a row per batch row (ld_emb = its stride; training)
    float* out_f32; long long ld_f32;     // optional float32 copy of the layer output [B, d1] (null: split rows only)
    float log_tiny;                       // log(1e-30) as float32 (tab_ddpm/utils.py:131)
    int act;
    char* out; long long pitch_out;       // split rows [B, d1]
    float scale_out;
    int* overflow;
};
int clava_launch_cat_input(const CatInputArgs& a, hipStream_t stream);

#ifdef __HIPCC__
#include "rng.h"
// one element of a split row (split.h): both planes, 2-byte stores
__device__ __forceinline__ void store_split(const SplitOut& o, int row, int col, float v) {
    const float sc = v * o.scale;
    if (!(fabsf(sc) <= 65504.f) && o.overflow) *o.overflow = 1;
    const _Float16 hi = (_Float16)sc;
    const _Float16 lo = (_Float16)(sc - (float)hi);
    _Float16* p = reinterpret_cast<_Float16*>(o.base + (long long)row * o.pitch + (long long)(col >> 5) * 128) + (col & 31);
    p[0] = hi;
    p[32] = lo;
}

// One element (row, col) of the fused Gaussian reverse step given the denoiser's eps for it: gaussian_p_mean_variance +
// condition_mean + gaussian_p_sample (tab_ddpm/gaussian_multinomial_diffsuion.py:200-237, 315-328, 354-383), operation by
// operation and promotion by promotion (float64 chain state, float32 tables, `mean.float()` inside condition_mean).
// Shared by gauss_step_kernel (diffusion_step.hip) and the fused denoiser-head kernel (fused_narrow.hip).  Files that use
// it are compiled with -ffp-contract=off so that a*b + c*d rounds like the separate ATen ops do.
// In three pieces so that a caller with several elements per thread can fetch everything first and then do the arithmetic
// (fused_narrow.hip: cls_back): the row's table entries, the element's inputs, the step itself.  gauss_step_element is the
// three in a row; the arithmetic is the same statements either way.
struct GaussRow {              // what depends on the row's t only
    int t;
    float sr, srm1, c1, c2;
    double var, sigma_keep;
};
__device__ __forceinline__ GaussRow gauss_row(const GaussStepArgs& a, int row) {
    GaussRow k;
    k.t = a.t_rows ? (int)a.t_rows[row] : a.t;
    k.sr = a.s.sqrt_recip_alphas_cumprod[k.t]; k.srm1 = a.s.sqrt_recipm1_alphas_cumprod[k.t];
    k.c1 = a.s.posterior_mean_coef1[k.t]; k.c2 = a.s.posterior_mean_coef2[k.t];
    k.var = a.grad ? a.s.model_variance[k.t] : 0.0;
    const double keep = (k.t != 0) ? 1.0 : 0.0;            // nonzero_mask (:373-375)
    k.sigma_keep = keep * a.s.model_sigma[k.t];
    return k;
}
struct GaussIn { double xd, z; };
// the element's standard normal when the kernel draws it itself (a.noise == null)
__device__ __forceinline__ double gauss_draw(const GaussStepArgs& a, int row, int col) {
    const unsigned long long elem = (a.row_offset + (unsigned long long)row) * (unsigned long long)a.n_num + col;
    const clava_rng::u4 r = clava_rng::philox4x32_10(a.seed, elem, a.step_index, clava_rng::STREAM_GAUSS);
    return (double)clava_rng::to_normal(r.x, r.y);
}
__device__ __forceinline__ GaussIn gauss_in(const GaussStepArgs& a, int row, int col) {
    const long long i = (long long)row * a.n_num + col;
    GaussIn v;
    v.xd = a.x[i];
    v.z = a.noise ? a.noise[i] : gauss_draw(a, row, col);
    return v;
}
// e = the denoiser's eps, g = the guidance gradient (ignored when a.grad is null)
__device__ __forceinline__ void gauss_apply(const GaussStepArgs& a, const GaussRow& k, const GaussIn& v, int row, int col, float e, float g) {
    const long long i = (long long)row * a.n_num + col;
    const double xd = v.xd;
    double pred, mean;
    if (a.x_is_f32) {          // first step of a chain: x is the float32 draw, the reference computes in float32 (:1029)
        const float xf = (float)xd;
        const float pf = k.sr * xf - k.srm1 * e;
        pred = (double)pf;
        mean = (double)(k.c1 * pf + k.c2 * xf);
    } else {                   // afterwards x is float64; float32 tables promote, eps * table stays a float32 product (:305-306)
        pred = (double)k.sr * xd - (double)(k.srm1 * e);
        mean = (double)k.c1 * pred + (double)k.c2 * xd;
    }
    if (a.grad)                // condition_mean: mean.float() + variance * gradient.float()  (:325-327)
        mean = (double)(float)mean + k.var * (double)g;
    const double out = mean + k.sigma_keep * v.z;
    a.x[i] = out;
    if (a.pred_xstart) a.pred_xstart[i] = pred;
    if (a.x_in) a.x_in[(size_t)row * a.ld_xin + col] = (float)out;
    if (a.xs.base) store_split(a.xs, row, a.xs.col0 + col, (float)out);
}
__device__ __forceinline__ void gauss_step_element(const GaussStepArgs& a, int row, int col, float e) {
    const GaussRow k = gauss_row(a, row);
    const GaussIn v = gauss_in(a, row, col);
    const float g = a.grad ? a.grad[(size_t)row * a.ld_grad + col] : 0.f;
    gauss_apply(a, k, v, row, col, e, g);
}
#endif

int clava_launch_gauss_step(const GaussStepArgs& a, hipStream_t stream);
int clava_launch_multinomial_step(const MultiStepArgs& a, hipStream_t stream);

// x_in[B, n_num + sumK] = float(x) | log-one-hot(idx): the denoiser input torch.cat([z_norm, log_z]).float()
// (tab_ddpm/gaussian_multinomial_diffsuion.py:1047) at the start of a chain.
int clava_launch_chain_input(const double* x, const int* idx, const int* offsets, float* x_in, int B, int n_num, int m,
                             int sumK, hipStream_t stream);
