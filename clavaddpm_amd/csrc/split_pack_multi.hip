// One launch that packs a list of float32 matrices into split rows (split.h) in BOTH orientations.
//
// The split-f16 training step re-packs every hidden weight each step (AdamW just changed it): W as the forward operand
// and W^T as the operand of the input-gradient product dy . W (scripts/train.py:37).  One launch per matrix and
// orientation was 8-10 graph nodes of a few microseconds each; this kernel does all of them: blockIdx.y = matrix,
// blockIdx.x = 64 x 64 tile.  A tile is read once (coalesced float4 rows), written as split rows directly from registers,
// and written transposed through a [64][65] LDS image (8-byte stores of 4 consecutive source rows per plane).
#include "split.h"
#include "clava_hip.h"

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

__global__ void __launch_bounds__(256) split_pack_multi_kernel(const clava_pack_desc* __restrict__ descs, int* overflow) {
    __shared__ float tile[64][65];
    const clava_pack_desc d = descs[blockIdx.y];
    const int tiles_c = (d.cols + 63) >> 6, tiles_r = (d.rows + 63) >> 6;
    if ((int)blockIdx.x >= tiles_c * tiles_r) return;
    const int r0 = ((int)blockIdx.x / tiles_c) << 6, c0 = ((int)blockIdx.x % tiles_c) << 6;
    const int colsp = (d.cols + 31) & ~31, rowsp = (d.rows + 31) & ~31;
    const float scale = d.scale;
    bool ovf = false;
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) << 2;      // 16 rows x 16 groups of 4 columns per pass
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = r0 + 16 * pass + tr, c = c0 + tc;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (r < d.rows && c + i < d.cols) ? d.src[(long long)r * d.ld + c + i] * scale : 0.f;
        h4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ovf |= !(fabsf(v[i]) <= 65504.f);
            hi[i] = (_Float16)v[i];
            lo[i] = (_Float16)(v[i] - (float)hi[i]);
            tile[16 * pass + tr][tc + i] = v[i];
        }
        if (d.dst && r < d.rows && c < colsp) {
            char* op = (char*)d.dst + (long long)r * d.pitch + (long long)(c >> 5) * 128 + ((c & 31) << 1);
            *reinterpret_cast<h4*>(op) = hi;
            *reinterpret_cast<h4*>(op + 64) = lo;
        }
    }
    if (d.dst_t) {
        __syncthreads();
        // transposed rows: thread -> (source column c0 + 16*pass + tr, 4 consecutive source rows r0 + tc ..)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int c = c0 + 16 * pass + tr, r = r0 + tc;
            if (c >= d.cols || r >= rowsp) continue;
            h4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = tile[tc + i][16 * pass + tr];               // zero beyond the source's rows
                hi[i] = (_Float16)v;
                lo[i] = (_Float16)(v - (float)hi[i]);
            }
            char* op = (char*)d.dst_t + (long long)c * d.pitch_t + (long long)(r >> 5) * 128 + ((r & 31) << 1);
            *reinterpret_cast<h4*>(op) = hi;
            *reinterpret_cast<h4*>(op + 64) = lo;
        }
    }
    if (ovf && overflow) *overflow = 1;
}

}  // namespace

int clava_split_pack_multi_f32(const clava_pack_desc* descs_dev, int n, int max_tiles, int* overflow, void* stream) {
    CLAVA_CHECK_ARG(descs_dev && n > 0 && max_tiles > 0, "split_pack_multi: bad arguments");
    ProfScope prof(PROF_MISC, 0.0, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(split_pack_multi_kernel, dim3(max_tiles, n), dim3(256), 0, (hipStream_t)stream, descs_dev, overflow);
    CLAVA_CHECK_LAUNCH("clava_split_pack_multi_f32");
    return CLAVA_OK;
}
