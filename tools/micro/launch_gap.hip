// Dependent tiny kernels: stream launches vs hipGraph replay (developer micro-benchmark).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
int main() {
    const int n = 1 << 20, K = 50, REP = 200;      // 4 MB per kernel, 4096 workgroups
    float* p; CK(hipMalloc(&p, n * sizeof(float))); CK(hipMemset(p, 0, n * sizeof(float)));
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int sz : {1 << 12, 1 << 16, 1 << 20}) {
        const dim3 g((sz + 255) / 256);
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny, g, dim3(256), 0, s, p, sz);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REP; ++r) for (int i = 0; i < K; ++i) hipLaunchKernelGGL(tiny, g, dim3(256), 0, s, p, sz);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("elements %8d: stream launches %.2f us per kernel\n", sz, ms * 1e3 / (REP * K));
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(tiny, g, dim3(256), 0, s, p, sz);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(exec, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(exec, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("elements %8d: graph replay    %.2f us per node (chain of %d)\n", sz, ms * 1e3 / (REP * K), K);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
