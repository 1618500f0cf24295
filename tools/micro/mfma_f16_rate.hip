// v_mfma_f32_16x16x32_f16 ceiling under the split-GEMM's instruction mix (developer tool): 8 waves per CU, each wave
// owns 40 accumulators; per "k block" 120 MFMAs and, optionally, 26 ds_read_b128 of fresh operands from LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool LDS, bool BAR>
__global__ void __launch_bounds__(512) k(float* out, int iters, const _Float16* init) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 72 * 1024 / 2; i += 512) reinterpret_cast<_Float16*>(smem)[i] = init[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[8][5];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = {0, 0, 0, 0};
    const char* base = smem + wave * 8192 + lane * 16;
    h8 ah[5], al[5], wh, wl;
    for (int j = 0; j < 5; ++j) { ah[j] = *(const h8*)(base + j * 1024); al[j] = *(const h8*)(base + j * 1024 + 512); }
    wh = *(const h8*)(base + 6000 * 0 + 5120); wl = *(const h8*)(base + 5632);
    for (int it = 0; it < iters; ++it) {
        const int off = LDS ? ((it & 1) * 1024) : 0;
        if (LDS) for (int j = 0; j < 5; ++j) { ah[j] = *(const h8*)(base + off + j * 1024); al[j] = *(const h8*)(base + off + j * 1024 + 512); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (LDS) { wh = *(const h8*)(base + off + i * 512 + 2048); wl = *(const h8*)(base + off + i * 512 + 2560); }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[j], acc[i][j], 0, 0, 0);
            }
        }
        if (BAR) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <bool LDS, bool BAR>
void run(const _Float16* init, float* out) {
    const int iters = 2000, grid = 256;
    (void)hipFuncSetAttribute((const void*)k<LDS, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<LDS, BAR>), dim3(grid), dim3(512), 72 * 1024, 0, out, 10, init);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<LDS, BAR>), dim3(grid), dim3(512), 72 * 1024, 0, out, iters, init);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 8 * iters * 120;
    printf("lds_reads=%d barrier=%d: %.0f TFLOP/s issued (%.1f%% of 2516.8), %.1f ns per k block\n", (int)LDS, (int)BAR,
           mfma * 16384 / ms / 1e9, mfma * 16384 / ms / 1e9 / 25.168, ms * 1e6 / iters);
}
int main() {
    _Float16 h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (_Float16)((float)((i * 2654435761u) >> 20 & 1023) / 512.f - 1.f);
    _Float16* init; float* out;
    (void)hipMalloc(&init, sizeof(h)); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemcpy(init, h, sizeof(h), hipMemcpyHostToDevice);
    run<false, false>(init, out); run<true, false>(init, out); run<true, true>(init, out); run<false, false>(init, out);
    return 0;
}
