// Counter-based RNG (Philox4x32-10) for in-kernel noise.  Every draw is a pure function of
// (seed, stream, step, global element index), so sampled tables do not depend on how rows are
// split over GPUs, launches or workgroups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clava_rng {

enum : uint32_t { STREAM_GAUSS = 0, STREAM_CAT = 1, STREAM_INIT_GAUSS = 2, STREAM_INIT_CAT = 3, STREAM_TRAIN = 4, STREAM_TRAIN_T = 5,
                  STREAM_TRAIN_NOISE = 6, STREAM_TRAIN_CAT = 7 };

struct u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(uint64_t seed, uint64_t elem, uint32_t step, uint32_t stream) {
    uint32_t c0 = (uint32_t)elem, c1 = (uint32_t)(elem >> 32), c2 = step, c3 = stream;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;   // one 32x32->64 multiply each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// uniform in [0, 1) with 24 random bits (the resolution of torch.rand for float32)
__device__ __forceinline__ float to_uniform(uint32_t v) { return (float)(v >> 8) * 5.9604644775390625e-8f; }

// standard normal by Box-Muller from two 32-bit words (u1 in (0, 1])
__device__ __forceinline__ float to_normal(uint32_t a, uint32_t b) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-8f;
    const float u2 = to_uniform(b);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

}  // namespace clava_rng
