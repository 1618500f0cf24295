// Shared declarations for the gfx950 kernels behind the C-ABI (include/clava_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CLAVA_OK 0
#define CLAVA_ERR_ARG (-1)      // bad shape / null pointer / unsupported flag
#define CLAVA_ERR_LAUNCH (-2)   // hipGetLastError() after a launch
#define CLAVA_ERR_LIMIT (-3)    // size beyond a stated maximum

void clava_set_error(const char* fmt, ...);

#define CLAVA_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            clava_set_error(__VA_ARGS__);          \
            return CLAVA_ERR_ARG;                  \
        }                                          \
    } while (0)

#define CLAVA_CHECK_LAUNCH(name)                                                         \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            clava_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return CLAVA_ERR_LAUNCH;                                                     \
        }                                                                                \
    } while (0)

// ---- optional per-launch timing (clava_prof_* in include/clava_hip.h) -----------------------------------------------
// When enabled, every kernel launch of the library is bracketed by two hipEvents on its own stream and tagged with
// its kernel family and algorithmic work; bench.py reads the totals for the roofline line.  Disabled: one branch.
enum { PROF_GEMM_F16X3_CFGS = 15 };
enum ProfFamily {
    PROF_GEMM_NT_128 = 0, PROF_GEMM_NT_64, PROF_GEMM_NT_128_BN, PROF_GEMM_NT_64_BN, PROF_GEMM_NN_128, PROF_GEMM_NN_64,
    PROF_GEMM_TN_128, PROF_GEMM_TN_64,
    PROF_GAUSS_STEP, PROF_MULTI_STEP, PROF_SELECT_GRAD, PROF_MISC, PROF_LOSS, PROF_ADAMW, PROF_GMM, PROF_BN,
    PROF_GEMM_F16X3,                                   // first of PROF_GEMM_F16X3_CFGS families, one per tile configuration
    PROF_FUSED_NARROW = PROF_GEMM_F16X3 + PROF_GEMM_F16X3_CFGS,   // fused_narrow.hip
    PROF_GEMM_F16X3_TN,                                // gemm_f16x3_tn.hip
    PROF_CAT_INPUT,                                    // cat_input.hip
    PROF_N_FAMILIES
};
extern bool g_clava_prof_on;
void clava_prof_begin_(int family, double flops, double bytes, hipStream_t stream);
void clava_prof_end_(hipStream_t stream);
struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(int family, double flops, double bytes, hipStream_t stream) : s(stream), on(g_clava_prof_on) {
        if (on) clava_prof_begin_(family, flops, bytes, stream);
    }
    ~ProfScope() {
        if (on) clava_prof_end_(s);
    }
};

// activation codes shared by the GEMM epilogue and the host side
enum { CLAVA_ACT_NONE = 0, CLAVA_ACT_RELU = 1, CLAVA_ACT_LEAKY = 2, CLAVA_ACT_SILU = 3 };

// Epilogue / prologue description of one GEMM launch (see gemm_f32.hip).
struct GemmEpi {
    const float* bias;       // [N] added to the accumulator, or null
    const float* rowadd;     // [M, N] (ldr) added after the bias, ldr == 0 broadcasts one row; or null
    int ldr;
    int act;                 // CLAVA_ACT_*
    const float* mask_src;   // [M, N] (ldm): multiply by (mask_src > 0 ? 1 : mask_slope); or null
    int ldm;
    float mask_slope;
    const float* col_scale;  // [N] multiply after the mask (BatchNorm scale on the way back); or null
    float alpha;             // final scale
    const float* a_scale;    // [K] A-operand prologue x*a_scale[k] + a_shift[k] (eval BatchNorm), K-major A only; or null
    const float* a_shift;
};

static inline GemmEpi gemm_epi_plain() {
    GemmEpi e;
    e.bias = nullptr; e.rowadd = nullptr; e.ldr = 0; e.act = CLAVA_ACT_NONE;
    e.mask_src = nullptr; e.ldm = 0; e.mask_slope = 0.f; e.col_scale = nullptr; e.alpha = 1.f;
    e.a_scale = nullptr; e.a_shift = nullptr;
    return e;
}

// C[M,N] = epi( op(A)[M,K] * op(B)[K,N] ).
//   a_kmajor: A stored [M,K] (lda) else stored [K,M] (lda)
//   b_kmajor: B stored [N,K] (ldb) (an nn.Linear weight) else stored [K,N] (ldb)
//   splitk > 1: K is cut into `splitk` ranges, range z writes the plain partial product to C + z*slab
int clava_launch_gemm_f32(const float* A, int lda, bool a_kmajor, const float* B, int ldb, bool b_kmajor,
                          float* C, int ldc, int M, int N, int K, const GemmEpi& epi, int splitk, long slab,
                          hipStream_t stream);
