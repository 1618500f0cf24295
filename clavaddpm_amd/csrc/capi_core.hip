// C-ABI entry points: library bookkeeping + the GEMM family (declared in include/clava_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "clava_hip.h"
#include "common.h"
#include "split.h"

static thread_local char g_err[512] = "";

void clava_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-launch timing ------------------------------------------------------------------------------------------------
bool g_clava_prof_on = false;
namespace {
struct ProfRec { hipEvent_t a, b; int family; double flops, bytes; };
std::vector<ProfRec> g_prof;
size_t g_prof_used = 0;
bool g_prof_open = false;
const char* const kFamilyNames[PROF_N_FAMILIES] = {
    "gemm_f32_nt_128x128", "gemm_f32_nt_64x64", "gemm_f32_nt_128x128_bn", "gemm_f32_nt_64x64_bn", "gemm_f32_nn_128x128",
    "gemm_f32_nn_64x64", "gemm_f32_tn_128x128", "gemm_f32_tn_64x64", "gauss_step", "multinomial_step", "logsoftmax_select_grad", "misc", "mixed_loss", "adamw_ema",
    "gmm_estep", "batchnorm_dropout", "gemm_f16x3_256x320", "gemm_f16x3_128x96", "gemm_f16x3_256x160", "gemm_f16x3_256x80", "gemm_f16x3_128x320",
    "gemm_f16x3_128x160", "gemm_f16x3_128x80", "gemm_f16x3_64x128", "gemm_f16x3_128x80_4stage", "gemm_f16x3_64x128_4stage", "gemm_f16x3_128x160_4w",
    "gemm_f16x3_128x320_4w", "gemm_f16x3_256x160_4w", "gemm_f16x3_128x128", "gemm_f16x3_128x128_4stage", "fused_narrow", "gemm_f16x3_tn", "cat_input_layer"};
}  // namespace

void clava_prof_begin_(int family, double flops, double bytes, hipStream_t stream) {
    g_prof_open = false;
    if (g_prof_used >= g_prof.size()) return;
    ProfRec& r = g_prof[g_prof_used];
    r.family = family; r.flops = flops; r.bytes = bytes;
    if (hipEventRecord(r.a, stream) == hipSuccess) g_prof_open = true;
}
void clava_prof_end_(hipStream_t stream) {
    if (!g_prof_open) return;
    if (hipEventRecord(g_prof[g_prof_used].b, stream) == hipSuccess) ++g_prof_used;
    g_prof_open = false;
}

extern "C" {

int clava_prof_enable(int max_records) {
    CLAVA_CHECK_ARG(max_records > 0 && max_records <= (1 << 20), "prof_enable: max_records out of range");
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof.resize(max_records);
    for (auto& r : g_prof) {
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
            clava_set_error("prof_enable: hipEventCreate failed");
            return CLAVA_ERR_LAUNCH;
        }
    }
    g_prof_used = 0;
    g_clava_prof_on = true;
    return CLAVA_OK;
}

int clava_prof_disable(void) {
    g_clava_prof_on = false;
    return CLAVA_OK;
}

int clava_prof_family_count(void) { return PROF_N_FAMILIES; }

const char* clava_prof_family_name(int family) {
    return (family >= 0 && family < PROF_N_FAMILIES) ? kFamilyNames[family] : "";
}

int clava_prof_collect(int* launches, double* total_ms, double* total_flops, double* total_bytes) {
    CLAVA_CHECK_ARG(launches && total_ms && total_flops && total_bytes, "prof_collect: null output");
    for (int f = 0; f < PROF_N_FAMILIES; ++f) { launches[f] = 0; total_ms[f] = total_flops[f] = total_bytes[f] = 0.0; }
    for (size_t i = 0; i < g_prof_used; ++i) {
        const ProfRec& r = g_prof[i];
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        launches[r.family] += 1;
        total_ms[r.family] += ms;
        total_flops[r.family] += r.flops;
        total_bytes[r.family] += r.bytes;
    }
    return (int)g_prof_used;
}

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

int clava_gemm_wgrad_f32(const float* dy, int lddy, const float* x, int ldx, float* dw_slabs, int lddw,
                         long long slab_stride, int M, int N_out, int K_in, int splits, void* stream) {
    CLAVA_CHECK_ARG(splits == 1 || slab_stride >= (long long)(N_out - 1) * lddw + K_in, "gemm_wgrad: slab_stride too small");
    GemmEpi e = gemm_epi_plain();
    // dw[N_out, K_in] = dy^T[N_out, M] * x[M, K_in]: contraction over the batch M; both operands MN-major
    return clava_launch_gemm_f32(dy, lddy, false, x, ldx, false, dw_slabs, lddw, N_out, K_in, M, e, splits,
                                 (long)slab_stride, (hipStream_t)stream);
}

// ---- split-f16 GEMM family (split.h) ------------------------------------------------------------------------------------
int clava_split_pack_f32(const float* src, long long ld, long long rows, int cols, int shift, void* dst, long long pitch,
                         int* overflow, void* stream) {
    return clava_launch_split_pack(src, ld, rows, cols, shift, (char*)dst, pitch, overflow, (hipStream_t)stream);
}

int clava_split_pack_transposed_f32(const float* src, long long ld, long long rows, int cols, int shift, void* dst, long long pitch,
                                    int* overflow, void* stream) {
    return clava_launch_split_pack(src, ld, rows, cols, shift, (char*)dst, pitch, overflow, (hipStream_t)stream, true);
}

int clava_split_unpack_f32(const void* src, long long pitch, long long rows, int cols, int shift, float* dst, long long ld,
                           void* stream) {
    return clava_launch_split_unpack((const char*)src, pitch, rows, cols, shift, dst, ld, (hipStream_t)stream);
}

int clava_split_transpose(const void* src, long long pitch_src, long long rows, int cols, void* dst, long long pitch_dst, void* stream) {
    return clava_launch_split_transpose((const char*)src, pitch_src, rows, cols, (char*)dst, pitch_dst, (hipStream_t)stream);
}

int clava_gemm_wgrad_f16x3(const void* dyt, long long pitch_dyt, int shift_dy, const void* xt, long long pitch_xt, int shift_x,
                           float* dw_slabs, int lddw, long long slab_stride, int M, int N_out, int K_in, int splits,
                           int tile_config, void* stream) {
    CLAVA_CHECK_ARG(M > 0 && N_out > 0 && K_in > 0 && splits >= 1 && dw_slabs, "gemm_wgrad_f16x3: bad arguments");
    CLAVA_CHECK_ARG(shift_dy + shift_x >= -60 && shift_dy + shift_x <= 60, "gemm_wgrad_f16x3: shift out of range");
    SplitGemmArgs g;
    g.A = (const char*)dyt; g.W = (const char*)xt;
    g.M = N_out; g.N = K_in; g.Kp = (M + 31) & ~31; g.K = M;
    g.pitchA = pitch_dyt; g.pitchW = pitch_xt;
    g.bias = nullptr; g.bias2 = nullptr; g.act = CLAVA_ACT_NONE;
    g.mask = nullptr; g.pitchMask = 0; g.mask_slope = 0.f;
    g.scale_acc = ldexpf(1.f, -(shift_dy + shift_x));
    g.alpha = 1.f; g.scale_out = 1.f;
    g.out_split = nullptr; g.pitchOut = 0;
    g.out_f32 = dw_slabs; g.ldc = lddw;
    g.overflow = nullptr;
    g.tiles_m = g.tiles_n = 0;
    const int nkb = g.Kp >> 5;
    g.kb_per_slab = (nkb + splits - 1) / splits;
    g.slabs = splits;
    CLAVA_CHECK_ARG((long long)(splits - 1) * g.kb_per_slab < nkb, "gemm_wgrad_f16x3: %d splits leave empty slabs at batch %d", splits, M);
    g.slab_stride = slab_stride;
    if (splits == 1) { g.slabs = 1; g.kb_per_slab = 0; }
    return clava_launch_gemm_f16x3(g, tile_config, (hipStream_t)stream);
}

int clava_gemm_wgrad_f16x3_tn(const void* dy, long long pitch_dy, int shift_dy, const void* x, long long pitch_x, int shift_x,
                              float* dw_slabs, int lddw, long long slab_stride, float* db_slabs, int M, int N_out, int K_in,
                              int splits, void* stream) {
    CLAVA_CHECK_ARG(M > 0 && N_out > 0 && K_in > 0 && splits >= 1 && dw_slabs, "gemm_wgrad_f16x3_tn: bad arguments");
    CLAVA_CHECK_ARG(M % 32 == 0, "gemm_wgrad_f16x3_tn: batch %d is not a multiple of 32 (use clava_gemm_wgrad_f16x3 on transposed copies)", M);
    CLAVA_CHECK_ARG(shift_dy + shift_x >= -60 && shift_dy + shift_x <= 60, "gemm_wgrad_f16x3_tn: shift out of range");
    SplitGemmArgs g;
    g.A = (const char*)dy; g.W = (const char*)x;
    g.M = N_out; g.N = K_in; g.Kp = M; g.K = M;
    g.pitchA = pitch_dy; g.pitchW = pitch_x;
    g.bias = nullptr; g.bias2 = nullptr; g.act = CLAVA_ACT_NONE;
    g.mask = nullptr; g.pitchMask = 0; g.mask_slope = 0.f;
    g.scale_acc = ldexpf(1.f, -(shift_dy + shift_x));
    g.alpha = 1.f; g.scale_out = 1.f;
    g.out_split = nullptr; g.pitchOut = 0;
    g.out_f32 = dw_slabs; g.ldc = lddw;
    g.overflow = nullptr;
    g.tiles_m = g.tiles_n = 0;
    const int nkb = M >> 5;
    g.kb_per_slab = (nkb + splits - 1) / splits;
    g.slabs = splits;
    CLAVA_CHECK_ARG((long long)(splits - 1) * g.kb_per_slab < nkb, "gemm_wgrad_f16x3_tn: %d splits leave empty slabs at batch %d", splits, M);
    g.slab_stride = slab_stride;
    g.out_colsum = db_slabs;
    g.scale_colsum = ldexpf(1.f, -shift_dy);
    if (splits == 1) { g.slabs = 1; g.kb_per_slab = 0; }
    return clava_launch_gemm_f16x3_tn(g, (hipStream_t)stream);
}

int clava_gemm_f16x3_config_count_(void) { return clava_gemm_f16x3_config_count(); }

int clava_gemm_f16x3(const void* x, long long pitch_x, int shift_x, const void* w, long long pitch_w, int shift_w,
                     const float* bias, const float* rowadd, int act, const void* act_out, long long pitch_act, float slope,
                     float alpha, void* y_split, long long pitch_y, int shift_y, float* y_f32, int ldy, int M, int N, int K,
                     int* overflow, int tile_config, void* stream) {
    CLAVA_CHECK_ARG(K > 0, "gemm_f16x3: K must be positive");
    CLAVA_CHECK_ARG(shift_x + shift_w >= -60 && shift_x + shift_w <= 60 && shift_y >= -30 && shift_y <= 30, "gemm_f16x3: shift out of range");
    SplitGemmArgs g;
    g.A = (const char*)x; g.W = (const char*)w;
    g.M = M; g.N = N; g.Kp = (K + 31) & ~31; g.K = K;
    g.pitchA = pitch_x; g.pitchW = pitch_w;
    g.bias = bias; g.bias2 = rowadd; g.act = act;
    g.mask = (const char*)act_out; g.pitchMask = pitch_act; g.mask_slope = slope;
    g.scale_acc = ldexpf(1.f, -(shift_x + shift_w));
    g.alpha = alpha;
    g.scale_out = ldexpf(1.f, shift_y);
    g.out_split = (char*)y_split; g.pitchOut = pitch_y;
    g.out_f32 = y_f32; g.ldc = ldy;
    g.overflow = overflow;
    g.tiles_m = g.tiles_n = 0;
    return clava_launch_gemm_f16x3(g, tile_config, (hipStream_t)stream);
}

}  // extern "C"
