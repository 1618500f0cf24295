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
};

int clava_launch_gauss_step(const GaussStepArgs& a, hipStream_t stream);
int clava_launch_multinomial_step(const MultiStepArgs& a, hipStream_t stream);

// x_in[B, n_num + sumK] = float(x) | log-one-hot(idx): the denoiser input torch.cat([z_norm, log_z]).float()
// (tab_ddpm/gaussian_multinomial_diffsuion.py:1047) at the start of a chain.
int clava_launch_chain_input(const double* x, const int* idx, const int* offsets, float* x_in, int B, int n_num, int m,
                             int sumK, hipStream_t stream);
