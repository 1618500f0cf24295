// v_fma_f64 issue rate on gfx950 (developer tool): NACC independent chains per lane, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, int iters, double b, double c) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], b, c);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu) {
    double* out;
    (void)hipMalloc(&out, 256 * 1024 * 8 * 8);
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 0.999, 1e-3);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999, 1e-3);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fma_n = (double)grid * 256 * iters * NACC;
    printf("NACC=%d waves/SIMD=%d: %.1f TFLOP/s f64 (v_fma_f64)\n", NACC, wgs_per_cu, fma_n * 2 / ms / 1e9);
    (void)hipFree(out);
}
int main() { run<4>(1); run<8>(1); run<8>(2); run<8>(4); return 0; }
