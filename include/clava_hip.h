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

#define CLAVA_HIP_VERSION 100

/* activation codes */
#define CLAVA_ACT_NONE_ 0
#define CLAVA_ACT_RELU_ 1
#define CLAVA_ACT_LEAKY_ 2 /* negative slope 0.01 = nn.LeakyReLU() default (pipeline_utils.py:615) */
#define CLAVA_ACT_SILU_ 3

int clava_version(void);
const char* clava_last_error(void);
int clava_device_count(void); /* host-only probe: number of HIP devices, 0 when there is none */

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

/* dw_slabs[s][N_out,K_in] = partial_s( dy[M,N_out]^T . x[M,K_in] ), s < splits: the batch dimension is cut into
 * `splits` ranges so that small weight matrices still fill the chip; the caller sums the slabs in a fixed order
 * (clava_adamw_ema_f32 does).  Replaces the weight-gradient half of loss.backward() (scripts/train.py:37). */
int clava_gemm_wgrad_f32(const float* dy, int lddy, const float* x, int ldx, float* dw_slabs, int lddw, int M,
                         int N_out, int K_in, int splits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLAVA_HIP_H */
