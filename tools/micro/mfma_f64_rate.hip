// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (developer tool): NACC independent accumulators per wave, WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu) {
    double* out;
    hipMalloc(&out, 256 * 1024 * 8 * 8);
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * NACC;
    printf("NACC=%d waves/SIMD=%d: %.1f TFLOP/s f64, %.1f ns per MFMA per SIMD\n", NACC, wgs_per_cu, mfma * 2048 / ms / 1e9,
           ms * 1e6 / (mfma / 1024.0));
    hipFree(out);
}
int main() { run<1>(1); run<2>(1); run<4>(1); run<8>(1); run<4>(2); run<8>(2); run<4>(4); return 0; }
