// Wavefront-group reductions on the DPP path and the lean exp / log used by the categorical kernels
// (diffusion_step.hip, train.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// Reductions over a group of G lanes without LDS traffic: xor-1 / xor-2 quad permutes and the two row mirrors on the
// DPP path (16 lanes), then one shuffle (G = 32) or four lane reads into scalar registers (G = 64, result wave-uniform).
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

template <int G> __device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    if (G >= 16) v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    if (G == 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (G == 64) {
        const int i = __float_as_int(v);
        v = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
                  fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
    }
    return v;
}
template <int G> __device__ __forceinline__ float group_sum(float v) {       // fixed order: deterministic
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    if (G >= 16) v += dpp_f<DPP_ROW_MIRROR>(v);
    if (G == 32) v += __shfl_xor(v, 16, 64);
    if (G == 64) {
        const int i = __float_as_int(v);
        v = (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
            (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
    }
    return v;
}
template <int G> __device__ __forceinline__ int group_min(int v) {
    v = min(v, dpp_i<DPP_XOR1>(v));
    v = min(v, dpp_i<DPP_XOR2>(v));
    v = min(v, dpp_i<DPP_HALF_MIRROR>(v));
    if (G >= 16) v = min(v, dpp_i<DPP_ROW_MIRROR>(v));
    if (G == 32) v = min(v, __shfl_xor(v, 16, 64));
    if (G == 64)
        v = min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
    return v;
}

// e^x for x <= 0: the library routine's two-term argument reduction without its overflow / denormal-result paths
// (arguments here are differences to a maximum); below -200 the result is 0 like the exact value in float32.
__device__ __forceinline__ float exp_nonpos(float x) {
    x = fmaxf(x, -200.0f);
    const float log2e = __int_as_float(0x3fb8aa3b);
    const float ph = x * log2e;
    float pl = __builtin_fmaf(x, log2e, -ph);
    pl = __builtin_fmaf(x, __int_as_float(0x32a5705f), pl);
    const float r = __builtin_amdgcn_exp2f(ph);
    return __builtin_fmaf(r, pl * 0.693147182f, r);
}
// log(x) for normal, finite x > 0: bit-identical to logf there (same hardware log2 and two-term product), without the
// denormal pre-scaling and the infinity select
__device__ __forceinline__ float log_pos(float x) {
    const float l = __builtin_amdgcn_logf(x);
    const float ln2 = __int_as_float(0x3f317217);
    const float ph = l * ln2;
    float pl = __builtin_fmaf(l, ln2, -ph);
    pl = __builtin_fmaf(l, __int_as_float(0x3377d1cf), pl);
    return ph + pl;
}
// log(1 + x) for x >= 0, accurate RELATIVE to the result.  log_pos(1 + x) loses that near x = 0: the sum rounds to
// float32 (6e-8 absolute on a result of size x) and the hardware log2 is only accurate in absolute terms around 1, so
// a normaliser of a concentrated distribution (log of 1 + 4e-5) came out 2 % off - harmless for an arg-max, but the
// training loss differentiates -w + exp(lm) with both terms within 1e-4 of one.  Below 1/4: 2 atanh(x / (2 + x)) by its
// odd series (five terms, truncation < 1e-10 relative at z <= 1/9); above: the plain logarithm (result >= 0.22).
__device__ __forceinline__ float log1p_acc(float x) {
    if (x > 0.25f) return log_pos(1.0f + x);
    const float z = x * __builtin_amdgcn_rcpf(2.0f + x), z2 = z * z;       // 1-ulp reciprocal: the series needs ~1e-7
    float p = __builtin_fmaf(z2, 1.0f / 9.0f, 1.0f / 7.0f);
    p = __builtin_fmaf(z2, p, 0.2f);
    p = __builtin_fmaf(z2, p, 1.0f / 3.0f);
    p = __builtin_fmaf(z2 * z, p, z);
    return 2.0f * p;
}
// log of (n_max + s): a log-sum-exp whose n_max largest terms (each exactly exp(0) = 1) were counted instead of summed and
// whose other terms add up to s
__device__ __forceinline__ float log_sum_counted(float n_max, float s) {
    return n_max == 1.0f ? log1p_acc(s) : log_pos(n_max + s);
}
// log_add_exp with the accurate log1p (per-block scalars of the loss kernel, where the larger argument can be ~0)
__device__ __forceinline__ float log_add_exp_acc(float x, float y) {
    return fmaxf(x, y) + log1p_acc(exp_nonpos(-fabsf(x - y)));
}
// log_add_exp (tab_ddpm/utils.py:104-106): one of the two exponentials is exp(0) = 1 exactly
__device__ __forceinline__ float log_add_exp(float x, float y) {
    const float m = fmaxf(x, y);
    return m + log_pos(1.0f + exp_nonpos(-fabsf(x - y)));
}

// ---- (row, feature) blocks packed into lane groups (categorical step kernel, multinomial loss kernel) ----
constexpr int MAX_LDS_FEATURES = 1024;

// Lane-group width of a feature: the smallest of 8 / 16 / 32 / 64 lanes that holds one class per lane (64 beyond).
__device__ __forceinline__ int feature_bucket(int K) { return K <= 8 ? 0 : K <= 16 ? 1 : K <= 32 ? 2 : 3; }

// Features ordered by lane-group width: s_perm lists them bucket by bucket, s_start[b] .. s_start[b + 1] is bucket b.
// Called by every thread of the workgroup once s_off (the m + 1 class offsets, m <= MAX_LDS_FEATURES) is filled and
// synchronised; returns synchronised.  The order must be THE SAME in every workgroup -- a work item is "features
// perm[8w .. 8w+7] of row r" and the items of a row are spread over the whole grid -- so one wavefront ranks the features
// in index order with ballots (a stable counting sort; positions handed out by LDS atomics depend on which wavefront
// gets there first once m > 64).
__device__ __forceinline__ void order_features_by_width(int m, const int* s_off, int* s_perm, int* s_start) {
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        int cnt[4] = {0, 0, 0, 0};
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            const int bk = (i < m) ? feature_bucket(s_off[i + 1] - s_off[i]) : -1;
#pragma unroll
            for (int b = 0; b < 4; ++b) cnt[b] += __popcll(__ballot(bk == b));
        }
        int base[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
        if (lane == 0) { s_start[0] = 0; s_start[1] = base[1]; s_start[2] = base[2]; s_start[3] = base[3]; s_start[4] = m; }
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            const int bk = (i < m) ? feature_bucket(s_off[i + 1] - s_off[i]) : -1;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned long long mask = __ballot(bk == b);
                if (bk == b) s_perm[base[b] + __popcll(mask & ((1ull << lane) - 1ull))] = i;
                base[b] += __popcll(mask);
            }
        }
    }
    __syncthreads();
}

}  // namespace
