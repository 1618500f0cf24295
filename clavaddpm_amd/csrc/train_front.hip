// Front of the denoiser's TRAINING forward as one kernel (per-row timesteps):
//
//   te  = [cos(t f_k) | sin(t f_k)]                        timestep_embedding      (tab_ddpm/modules.py:20-38)
//   z1  = te W0^T + b0 ;  a1 = SiLU(z1) ;  emb = a1 W2^T + b2      time_embed      (modules.py:445-449, 452)
//   h   = x_in Wp^T + bp + emb                               proj(x) + emb          (modules.py:459)
//
// and h written as the split rows (split.h) the first hidden block's product reads.  The step used six launches for this
// (embedding, product, SiLU, product, product, split pack), ~35 us of kernels plus a launch gap each, all of it on the
// step's critical path in front of the first hidden block; the arithmetic is 0.27 GFLOP.  te, z1 and a1 are stored because
// the backward reads them (time_embed.0 / .2 weight gradients, SiLU').
//
// Plain float32 FMAs in ascending k (the accuracy class of the float32 MFMA kernels it replaces).  A workgroup of 256
// threads carries 16 rows; thread (c = tid & 127, half = tid >> 7) owns output column c of rows 8 half .. 8 half + 7.  A
// weight matrix is staged once per workgroup into LDS as [128 outputs][132] floats (16-byte reads at lane stride 132
// floats = 4 banks: conflict free), the activations of the 16 rows as [16][132] read as wave-uniform broadcasts.  dim_t is
// 128 (every config; other widths take the unfused kernels); x_in may have any number of columns (128 per pass).
#include "clava_hip.h"
#include "common.h"

typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DT = 128, ROWS = 16, LDW = DT + 4;

struct FrontArgs {
    const long long* t; const float* freqs;
    const float* x; long long ldx; int D;
    const float *w0, *b0, *w2, *b2, *wp, *bp;
    float *te, *z1, *a1, *h;          // [B][128] each; h may be null
    char* hs; long long pitch; float scale; int* overflow;      // split rows of h; hs may be null
    int B;
};

// W[0..127][k0 .. k0+kc) (row stride ldw floats) -> registers; idx = tid + 256 j covers (row = idx >> 5, 4 columns at 4 (idx & 31))
__device__ __forceinline__ void fetch_w(f4 (&pre)[16], const float* __restrict__ W, long long ldw, int k0, int kc, int tid) {
    const bool vec = (ldw & 3) == 0 && (k0 & 3) == 0 && ((reinterpret_cast<unsigned long long>(W) & 15) == 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int idx = tid + 256 * j, r = idx >> 5, k = (idx & 31) << 2;
        const float* p = W + r * ldw + k0 + k;
        if (vec && k + 3 < kc) pre[j] = *reinterpret_cast<const f4*>(p);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) pre[j][e] = (k + e < kc) ? p[e] : 0.f;
        }
    }
}
__device__ __forceinline__ void commit_w(const f4 (&pre)[16], float* wl, int tid) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int idx = tid + 256 * j;
        *reinterpret_cast<f4*>(wl + (idx >> 5) * LDW + ((idx & 31) << 2)) = pre[j];
    }
}

// acc[i] += sum_k act[8 half + i][k] * wl[c][k] over k < kq * 4
__device__ __forceinline__ void fma_rows(float (&acc)[8], const float* act, const float* wl, int c, int half, int kq) {
    const float* wr = wl + c * LDW;
    const float* ar = act + 8 * half * LDW;
#pragma unroll 2
    for (int q = 0; q < kq; ++q) {
        const f4 w = *reinterpret_cast<const f4*>(wr + 4 * q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f4 a = *reinterpret_cast<const f4*>(ar + i * LDW + 4 * q);
            acc[i] = fmaf(a[0], w[0], acc[i]);
            acc[i] = fmaf(a[1], w[1], acc[i]);
            acc[i] = fmaf(a[2], w[2], acc[i]);
            acc[i] = fmaf(a[3], w[3], acc[i]);
        }
    }
}

__global__ void __launch_bounds__(256) train_front_kernel(FrontArgs a) {
    __shared__ __attribute__((aligned(16))) float wl[DT * LDW];
    __shared__ __attribute__((aligned(16))) float act0[ROWS * LDW], act1[ROWS * LDW];
    const int tid = threadIdx.x, c = tid & 127, half = tid >> 7;
    const long long row0 = (long long)blockIdx.x * ROWS;
    f4 pre[16];

    // ---- h = x_in Wp^T, 128 input columns per pass
    float hacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hacc[i] = 0.f;
    for (int k0 = 0; k0 < a.D; k0 += DT) {
        const int kc = min(DT, a.D - k0);
        fetch_w(pre, a.wp, a.D, k0, kc, tid);
        if (k0) __syncthreads();                              // the previous pass is done with wl / act0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, r = idx >> 7, k = idx & 127;
            const long long row = min(row0 + r, (long long)a.B - 1);
            act0[r * LDW + k] = (k < kc) ? a.x[row * a.ldx + k0 + k] : 0.f;
        }
        commit_w(pre, wl, tid);
        __syncthreads();
        fma_rows(hacc, act0, wl, c, half, (kc + 3) >> 2);
    }
    fetch_w(pre, a.w0, DT, 0, DT, tid);                       // in flight while the embedding is computed
    __syncthreads();

    // ---- te -> act0 and memory: 16 rows x 64 frequencies, 4 (row, k) pairs per thread
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j, r = idx >> 6, k = idx & 63;
        const long long row = row0 + r;
        const float arg = (float)a.t[min(row, (long long)a.B - 1)] * a.freqs[k];
        const float cs = cosf(arg), sn = sinf(arg);
        act0[r * LDW + k] = cs;
        act0[r * LDW + 64 + k] = sn;
        if (row < a.B) {
            a.te[row * DT + k] = cs;
            a.te[row * DT + 64 + k] = sn;
        }
    }
    commit_w(pre, wl, tid);
    __syncthreads();

    // ---- z1 = te W0^T + b0, a1 = SiLU(z1) -> act1 and memory
    float acc[8];
    {
        const float b = a.b0[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = b;
    }
    fetch_w(pre, a.w2, DT, 0, DT, tid);                       // in flight during the product
    fma_rows(acc, act0, wl, c, half, DT / 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = 8 * half + i;
        const float z = acc[i], s = z / (1.0f + expf(-z));
        act1[r * LDW + c] = s;
        if (row0 + r < a.B) {
            a.z1[(row0 + r) * DT + c] = z;
            a.a1[(row0 + r) * DT + c] = s;
        }
    }
    __syncthreads();                                          // every thread is done with W0; a1 is complete
    commit_w(pre, wl, tid);
    __syncthreads();

    // ---- emb = a1 W2^T + b2 ;  h = (x Wp^T + bp) + emb
    {
        const float b = a.b2[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = b;
    }
    fma_rows(acc, act1, wl, c, half, DT / 4);
    const float bp = a.bp ? a.bp[c] : 0.f;                    // no proj: h = emb (the caller's first layer adds the rest)
    bool ovf = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long row = row0 + 8 * half + i;
        if (row >= a.B) continue;
        const float h = (hacc[i] + bp) + acc[i];
        if (a.h) a.h[row * DT + c] = h;
        if (a.hs) {
            const float s = h * a.scale;
            ovf |= !(fabsf(s) <= 65504.f);
            const _Float16 hi = (_Float16)s, lo = (_Float16)(s - (float)hi);
            char* p = a.hs + row * a.pitch + (c >> 5) * 128 + ((c & 31) << 1);
            *reinterpret_cast<_Float16*>(p) = hi;
            *reinterpret_cast<_Float16*>(p + 64) = lo;
        }
    }
    if (ovf && a.overflow) *a.overflow = 1;
}

// Backward through time_embed[2] and the SiLU in one launch: d_z1 = SiLU'(z1) * (g W2), g = d loss / d emb [B, 128]
// (the autograd backward of tab_ddpm/modules.py:445-449 between emb and z1).  Same tiling as the front: 16 rows per
// workgroup, W2 staged TRANSPOSED in LDS ([input k][132]) so that thread k of a row contracts over the output index.
struct TimeBwdArgs {
    const float* g; long long ldg;
    const float* w2;                  // [128 out][128 in]
    const float* z1; float* dz1;      // [B, 128]
    int B;
};

__global__ void __launch_bounds__(256) train_time_bwd_kernel(TimeBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float wl[DT * LDW];
    __shared__ __attribute__((aligned(16))) float act0[ROWS * LDW];
    const int tid = threadIdx.x, c = tid & 127, half = tid >> 7;
    const long long row0 = (long long)blockIdx.x * ROWS;
    f4 pre[16];
    fetch_w(pre, a.w2, DT, 0, DT, tid);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j, r = idx >> 7, k = idx & 127;
        act0[r * LDW + k] = a.g[min(row0 + r, (long long)a.B - 1) * a.ldg + k];
    }
    // wl[k][o] = W2[o][k]: element e of pre[j] is W2[row = idx >> 5][4 (idx & 31) + e]
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int idx = tid + 256 * j, o = idx >> 5, k = (idx & 31) << 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) wl[(k + e) * LDW + o] = pre[j][e];
    }
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    fma_rows(acc, act0, wl, c, half, DT / 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long row = row0 + 8 * half + i;
        if (row >= a.B) continue;
        const float v = a.z1[row * DT + c];
        const float sg = 1.0f / (1.0f + expf(-v));
        a.dz1[row * DT + c] = acc[i] * (sg * (1.0f + v * (1.0f - sg)));      // silu_bwd_kernel's formula (train.hip)
    }
}

}  // namespace

extern "C" int clava_train_front_f32(const long long* t, int B, const float* freqs, int dim_t, const float* x_in, long long ldx, int d_in,
                                     const float* w0, const float* b0, const float* w2, const float* b2, const float* wp,
                                     const float* bp, float* te, float* z1, float* a1, float* h, void* h_split, long long pitch,
                                     int shift, int* overflow, void* stream) {
    CLAVA_CHECK_ARG(dim_t == DT, "train_front: dim_t %d not supported (the fused front is built for 128)", dim_t);
    CLAVA_CHECK_ARG(t && freqs && w0 && b0 && w2 && b2 && te && z1 && a1 && (h || h_split), "train_front: null operand");
    CLAVA_CHECK_ARG(B > 0 && d_in >= 0 && (d_in == 0 || (x_in && wp && bp && ldx >= d_in)), "train_front: bad shape B=%d d_in=%d ldx=%lld", B,
                    d_in, ldx);
    if (d_in == 0) wp = bp = nullptr;
    if (h_split) {
        CLAVA_CHECK_ARG(pitch >= 4LL * DT && pitch % 8 == 0 && (reinterpret_cast<unsigned long long>(h_split) & 7) == 0,
                        "train_front: split pitch %lld too small or misaligned", pitch);
        CLAVA_CHECK_ARG(shift >= -24 && shift <= 24, "train_front: shift %d out of range", shift);
    }
    FrontArgs a;
    a.t = t; a.freqs = freqs; a.x = x_in; a.ldx = ldx; a.D = d_in;
    a.w0 = w0; a.b0 = b0; a.w2 = w2; a.b2 = b2; a.wp = wp; a.bp = bp;
    a.te = te; a.z1 = z1; a.a1 = a1; a.h = h;
    a.hs = (char*)h_split; a.pitch = pitch; a.scale = ldexpf(1.f, shift); a.overflow = overflow;
    a.B = B;
    const double flops = 2.0 * B * DT * (2.0 * DT + d_in);
    ProfScope prof(PROF_MISC, flops, 4.0 * B * (4.0 * DT + d_in), (hipStream_t)stream);
    hipLaunchKernelGGL(train_front_kernel, dim3((B + ROWS - 1) / ROWS), dim3(256), 0, (hipStream_t)stream, a);
    CLAVA_CHECK_LAUNCH("clava_train_front_f32");
    return CLAVA_OK;
}

extern "C" int clava_train_time_bwd_f32(const float* g, long long ldg, int B, int dim_t, const float* w2, const float* z1, float* d_z1,
                                        void* stream) {
    CLAVA_CHECK_ARG(dim_t == DT, "train_time_bwd: dim_t %d not supported (built for 128)", dim_t);
    CLAVA_CHECK_ARG(g && w2 && z1 && d_z1 && B > 0 && ldg >= DT, "train_time_bwd: bad arguments");
    TimeBwdArgs a;
    a.g = g; a.ldg = ldg; a.w2 = w2; a.z1 = z1; a.dz1 = d_z1; a.B = B;
    ProfScope prof(PROF_MISC, 2.0 * B * DT * DT, 4.0 * B * 3.0 * DT, (hipStream_t)stream);
    hipLaunchKernelGGL(train_time_bwd_kernel, dim3((B + ROWS - 1) / ROWS), dim3(256), 0, (hipStream_t)stream, a);
    CLAVA_CHECK_LAUNCH("clava_train_time_bwd_f32");
    return CLAVA_OK;
}
