// Weight gradient of the categorical columns of the denoiser's first layer (proj) without the [B, sum K] log-one-hot image.
//
// The input columns of feature f are index_to_log_onehot(idx_f) (tab_ddpm/utils.py:127-134): 0 at the row's class, L =
// log(1e-30) elsewhere.  With g = d loss / d h ([B, d1]) the weight half of loss.backward() (scripts/train.py:37) is
//     dW[c][n_num + off_f + k] = sum_b g[b][c] x[b][off_f + k] = L * (G[c] - H_f[k][c]),
//     G[c] = sum_b g[b][c],   H_f[k][c] = sum over the rows b whose class of feature f is k of g[b][c]
// i.e. one add of a 128-float row of g per (row, feature) instead of a [d1 x B] x [B x sum K] product: at the scale-sweep
// shape 4096 * 50 row adds against a 9 GFLOP product that took 190 us on the matrix cores, and the product's operand --
// the split image of x_in, 137 MB per step -- is never built (its forward counterpart: cat_input.hip).
//
// A workgroup owns (feature, batch range, range of <= 128 classes); thread c owns column c of an LDS table [classes][129]
// and walks the rows of the batch range in order, adding g[b][c] to the row of b's class (a plain LDS read-add-write;
// ds_add_f32 was tried and is 2x slower here: LDS float adds retire at ~600 cycles per wave instruction).  Only thread c
// ever touches column c and a wave's LDS operations execute in order, so every sum has a fixed order: slabs are bitwise
// reproducible, no atomics.  G comes from the bias-gradient slabs of the same layer (the caller computes them first).
// Written out with the lanes along the classes (contiguous in dW's rows).
#include "clava_hip.h"
#include "common.h"

namespace {

constexpr int D1 = 128, KR = 128, LDT = D1 + 1, RCH = 512;      // 129 rows x 129 floats = 66.6 KB of LDS: two workgroups per CU

struct CatWgradArgs {
    const float* g; long long ldg;       // [B, 128]
    const float* G;                      // bias gradient slabs: G[s * slab_stride + c] = sum of g[:, c] over batch range s
    const int* idx; int m;               // [B, m] classes
    const int* offsets;                  // [m + 1]
    const int* wg_feature; const int* wg_k0;      // per workgroup (blockIdx.x): feature and first class of its range
    int B, splits;
    float log_tiny;
    float* dw; long long ldw; int col0; long long slab_stride;       // dW slabs [splits][128][ldw], categorical columns from col0
};

__global__ void __launch_bounds__(128) cat_wgrad_kernel(CatWgradArgs a) {
    extern __shared__ float tab[];                   // [kr + 1][LDT]: H of this (feature, class range, batch range) + a dump row
    __shared__ __attribute__((aligned(16))) int s_idx[RCH];      // LDT * (class - k0), or LDT * kr (the dump row) out of range
    const int c = threadIdx.x;
    const int f = a.wg_feature[blockIdx.x], k0 = a.wg_k0[blockIdx.x];
    const int o = a.offsets[f], K = a.offsets[f + 1] - o;
    const int kr = min(KR, K - k0);
    for (int i = threadIdx.x; i < (kr + 1) * LDT; i += 128) tab[i] = 0.f;
    const int rows_per = (a.B + a.splits - 1) / a.splits;
    const int r0 = blockIdx.y * rows_per, r1 = min(a.B, r0 + rows_per);
    // The row loop is INSTRUCTION bound (every workgroup of a batch range walks all of its rows for its <= 128 classes; a
    // first version spent ~60 instructions per row on 64-bit addresses, a branch and an LDS read per row and ran 150 us
    // whatever the memory system did): 16 rows per step, branch-free -- rows of other classes add into a dump row --, four
    // class offsets per LDS read, one 32-bit address add per load, two register sets used in turn.
    float* const col = tab + c;
    for (int rb = r0; rb < r1; rb += RCH) {
        const int nr = min(RCH, r1 - rb);
        __syncthreads();
        for (int i = threadIdx.x; i < RCH; i += 128) {
            const int k = (i < nr) ? a.idx[(long long)(rb + i) * a.m + f] - k0 : -1;
            s_idx[i] = LDT * ((k >= 0 && k < kr) ? k : kr);
        }
        __syncthreads();
        constexpr int RB = 16;
        const char* gp = reinterpret_cast<const char*>(a.g + (long long)rb * a.ldg + c);
        const unsigned step = (unsigned)a.ldg * 4u;
        const int nfull = nr / RB;
        float va[RB], vb[RB];
        auto fetch = [&](float (&v)[RB], int bi) {
            const char* p = gp + (size_t)bi * RB * step;
#pragma unroll
            for (int u = 0; u < RB; ++u) v[u] = *reinterpret_cast<const float*>(p + u * step);
        };
        auto process = [&](const float (&v)[RB], int bi) {
#pragma unroll
            for (int q = 0; q < RB / 4; ++q) {
                const int4 k4 = *reinterpret_cast<const int4*>(s_idx + bi * RB + 4 * q);      // workgroup-uniform
                col[k4.x] += v[4 * q];
                col[k4.y] += v[4 * q + 1];
                col[k4.z] += v[4 * q + 2];
                col[k4.w] += v[4 * q + 3];
            }
        };
        if (nfull > 0) fetch(va, 0);
        for (int bi = 0; bi < nfull; bi += 2) {
            if (bi + 1 < nfull) fetch(vb, bi + 1);
            process(va, bi);
            if (bi + 2 < nfull) fetch(va, bi + 2);
            if (bi + 1 < nfull) process(vb, bi + 1);
        }
        for (int r = nfull * RB; r < nr; ++r) col[s_idx[r]] += *reinterpret_cast<const float*>(gp + (size_t)r * step);
    }
    __syncthreads();
    // dW[cc][col0 + o + k0 + k] = L * (G[cc] - H[k][cc]); lanes along k
    float* slab = a.dw + (long long)blockIdx.y * a.slab_stride + a.col0 + o + k0;
    const float* Gs = a.G + (long long)blockIdx.y * a.slab_stride;
    for (int cc = 0; cc < D1; ++cc) {
        const float Gc = Gs[cc];
        for (int k = threadIdx.x; k < kr; k += 128) slab[(long long)cc * a.ldw + k] = a.log_tiny * (Gc - tab[k * LDT + cc]);
    }
}

}  // namespace

extern "C" int clava_cat_wgrad_workgroups(const int* offsets_host, int m) {
    if (!offsets_host || m <= 0) return 0;
    int n = 0;
    for (int f = 0; f < m; ++f) n += (offsets_host[f + 1] - offsets_host[f] + KR - 1) / KR;
    return n;
}

extern "C" int clava_cat_wgrad_plan(const int* offsets_host, int m, int* wg_feature_host, int* wg_k0_host) {
    CLAVA_CHECK_ARG(offsets_host && m > 0 && wg_feature_host && wg_k0_host, "cat_wgrad_plan: bad arguments");
    int n = 0;
    for (int f = 0; f < m; ++f)
        for (int k0 = 0; k0 < offsets_host[f + 1] - offsets_host[f]; k0 += KR) { wg_feature_host[n] = f; wg_k0_host[n] = k0; ++n; }
    return CLAVA_OK;
}

extern "C" int clava_cat_wgrad_f32(const float* g, long long ldg, const float* g_colsum_slabs, int B, int d1, const int* idx, int m,
                                   const int* offsets, const int* wg_feature, const int* wg_k0, int n_workgroups, int splits,
                                   float log_tiny, float* dw_slabs, long long ldw, int col0, long long slab_stride, void* stream) {
    CLAVA_CHECK_ARG(d1 == D1, "cat_wgrad: layer width %d not supported (built for 128, dim_t of every config)", d1);
    CLAVA_CHECK_ARG(g && g_colsum_slabs && idx && offsets && wg_feature && wg_k0 && dw_slabs && B > 0 && m > 0 && n_workgroups > 0 && splits >= 1 &&
                        splits <= 65535 && ldg >= d1 && col0 >= 0 && ldw > col0,
                    "cat_wgrad: bad arguments");
    CatWgradArgs a;
    a.g = g; a.ldg = ldg; a.G = g_colsum_slabs; a.idx = idx; a.m = m; a.offsets = offsets; a.wg_feature = wg_feature; a.wg_k0 = wg_k0;
    a.B = B; a.splits = splits; a.log_tiny = log_tiny; a.dw = dw_slabs; a.ldw = ldw; a.col0 = col0; a.slab_stride = slab_stride;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    const int lds = (KR + 1) * LDT * 4;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cat_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            clava_set_error("cat_wgrad: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
            return CLAVA_ERR_LAUNCH;
        }
        attr_set[dev] = true;
    }
    ProfScope prof(PROF_MISC, 2.0 * B * (double)d1 * m, 4.0 * ((double)B * d1 * n_workgroups + (double)splits * d1 * ldw), (hipStream_t)stream);
    hipLaunchKernelGGL(cat_wgrad_kernel, dim3(n_workgroups, splits), dim3(128), lds, (hipStream_t)stream, a);
    CLAVA_CHECK_LAUNCH("clava_cat_wgrad_f32");
    return CLAVA_OK;
}
