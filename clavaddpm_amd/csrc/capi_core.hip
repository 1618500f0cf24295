// C-ABI entry points: library bookkeeping + the GEMM family (declared in include/clava_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "clava_hip.h"
#include "common.h"

static thread_local char g_err[512] = "";

void clava_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int clava_version(void) { return CLAVA_HIP_VERSION; }

const char* clava_last_error(void) { return g_err; }

int clava_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int clava_gemm_bias_act_f32(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowadd,
                            int ld_rowadd, const float* in_scale, const float* in_shift, float* y, int ldy, int M,
                            int N, int K, int act, void* stream) {
    CLAVA_CHECK_ARG(act >= CLAVA_ACT_NONE && act <= CLAVA_ACT_SILU, "gemm_bias_act: unknown activation %d", act);
    CLAVA_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "gemm_bias_act: in_scale/in_shift must come together");
    GemmEpi e = gemm_epi_plain();
    e.bias = bias;
    e.rowadd = rowadd;
    e.ldr = ld_rowadd;
    e.act = act;
    e.a_scale = in_scale;
    e.a_shift = in_shift;
    return clava_launch_gemm_f32(x, ldx, true, w, ldw, true, y, ldy, M, N, K, e, 1, 0, (hipStream_t)stream);
}

int clava_gemm_dgrad_f32(const float* dy, int lddy, const float* w, int ldw, const float* act_out, int ld_act,
                         float slope, const float* col_scale, float alpha, float* dx, int lddx, int M, int N_out,
                         int K_in, void* stream) {
    GemmEpi e = gemm_epi_plain();
    e.mask_src = act_out;
    e.ldm = ld_act;
    e.mask_slope = slope;
    e.col_scale = col_scale;
    e.alpha = alpha;
    // dx[M, K_in] = dy[M, N_out] * w[N_out, K_in]: contraction over N_out, w is "K x N" row-major for this product
    return clava_launch_gemm_f32(dy, lddy, true, w, ldw, false, dx, lddx, M, K_in, N_out, e, 1, 0, (hipStream_t)stream);
}

int clava_gemm_wgrad_f32(const float* dy, int lddy, const float* x, int ldx, float* dw_slabs, int lddw, int M,
                         int N_out, int K_in, int splits, void* stream) {
    GemmEpi e = gemm_epi_plain();
    // dw[N_out, K_in] = dy^T[N_out, M] * x[M, K_in]: contraction over the batch M; both operands MN-major
    return clava_launch_gemm_f32(dy, lddy, false, x, ldx, false, dw_slabs, lddw, N_out, K_in, M, e, splits,
                                 (long)N_out * lddw, (hipStream_t)stream);
}

}  // extern "C"
