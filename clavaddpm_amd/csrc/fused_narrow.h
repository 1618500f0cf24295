// Argument blocks of the fused narrow-stack kernels (fused_narrow.hip), filled by the chain driver (chain.hip).
#pragma once
#include "common.h"
#include "step.h"

// Epilogue data of one Linear layer of a fused kernel: bias = 32 * n-tiles floats (zero padded, 16-byte aligned); bias2 = the
// broadcast time-embedding row of layer 0, or null.  The weights of all layers of a kernel come as ONE stream of MFMA
// fragments in consumption order (`stream` of the argument blocks; clavaddpm_amd/split_nets.py::fused_streams).
struct FusedLayer {
    const float* bias = nullptr;
    const float* bias2 = nullptr;
    float scale_acc = 1.f;      // 2^-(shift of the input planes + shift of the weight planes)
    float scale_out = 1.f;      // 2^(shift of the output planes)
    int act = 0;
};

// cls_front: l[0] proj (+ emb row), l[1] 128 -> 128, l[2] 128 -> 256, out = activation of l[2] as split rows.
// den_front: l[0] proj (+ emb row), l[1] 128 -> 32 * nt_out (nt_out a multiple of 4), out = activation of l[1].
struct FusedFrontArgs {
    int B = 0;
    const char* stream = nullptr; int n_chunks = 0;     // weight stream (n_chunks is filled in by the launcher)
    const char* x = nullptr; long long pitch_x = 0;     // chain input as split rows (<= 32 columns)
    FusedLayer l[3];
    int nt_out = 0;
    char* out = nullptr; long long pitch_out = 0;
    int* overflow = nullptr;
    unsigned* mask1 = nullptr;                          // cls_front: [B][2 lane halves][2] words, act'(l[1]) > 0 as bits in the
                                                        // lanes' register order - read back by cls_back (private layout)
};

struct FusedTailArgs {
    int B = 0, C = 0;
    const char* stream = nullptr; int n_chunks = 0;
    const char* a6 = nullptr; long long pitch_a6 = 0;   // activation of the 512 -> 256 layer (split rows)
    const int* y = nullptr;                             // [B] guidance labels
    FusedLayer l7, l8;                                  // 256 -> 128 (leaky), 128 -> C (padded to 64)
    float scale_dlast = 1.f;                            // 2^shift_dlast: planes of onehot - softmax
    float scale_acc_g8 = 1.f, scale_out_g8 = 1.f, slope7 = 0.f;   // d/d(pre-activation of l7): true scale, plane scale, act' off a7
    float scale_acc_g7 = 1.f, scale_out_g7 = 1.f, slope6 = 0.f;   // d/d(pre-activation of the 512 -> 256 layer)
    char* out = nullptr; long long pitch_out = 0;       // [B, 256] split rows
    int* overflow = nullptr;
};

struct FusedBackArgs {
    int B = 0, n_num = 0;
    const char* stream = nullptr; int n_chunks = 0;
    const char* g3 = nullptr; long long pitch_g3 = 0;   // d/d(pre-activation of l[2]) [B, 256] split rows
    const unsigned* mask1 = nullptr;                    // act'(l[1]) bits as cls_front stored them
    float scale_acc_g2 = 1.f, scale_out_g2 = 1.f, slope1 = 0.f;
    float scale_acc_g1 = 1.f, scale_out_g1 = 1.f;
    float scale_acc_g0 = 1.f;                           // includes classifier_scale
    float* gx = nullptr; int ld_gx = 0;
    int* overflow = nullptr;
    int gauss = 0;                                      // 1: the row's Gaussian reverse step follows in the same thread
    GaussStepArgs g;                                    // (g.grad = gx, g.eps = the denoiser's output, complete before the launch)
};

struct FusedHeadArgs {
    const char* stream = nullptr; int n_chunks = 0;
    const char* h = nullptr; long long pitch_h = 0;     // last hidden activation (split rows), 16 * ks columns
    int ks = 0;
    FusedLayer l;                                       // head: 16 * ks -> n_num (one 32-row n-tile)
    GaussStepArgs g;                                    // eps / ld_eps unused: the head's output feeds the step in registers
};

int clava_launch_cls_front(FusedFrontArgs a, int n_in, hipStream_t st);
int clava_launch_den_front(FusedFrontArgs a, int n_in, hipStream_t st);
int clava_launch_cls_tail(FusedTailArgs a, hipStream_t st);
int clava_launch_cls_back(FusedBackArgs a, hipStream_t st);
int clava_launch_den_head_gauss(FusedHeadArgs a, hipStream_t st);
bool clava_den_front_supports(int hidden);     // first hidden width the fused denoiser front is built for
bool clava_den_head_supports(int hidden);      // last hidden width the fused head is built for
long long clava_fused_stream_bytes(int pairs);
