// Majority vote of the child rows' cluster labels per parent group, and the per-cluster histogram of group sizes.
//
// Replaces the Python loop of pipeline_modules.py:300-321 (per group: np.bincount of the group's labels, np.argmax =
// the lowest label among the most frequent ones, agree rate = its count / group length, np.repeat of the winner over
// the group's rows) and the (cluster, group length) frequency table of :332-340 that freq_to_prob (pipeline_utils.py:
// 1125-1129) turns into the group-size distributions used when sampling child tables.
//
// Integer work, HBM-bound: 4 bytes in + 4 bytes out per child row, 8 + 8 per group.  One wavefront owns one group: its
// lanes count the group's labels into a private LDS histogram (LDS integer atomics), the histogram is scanned for the
// largest count with the lowest label winning ties, the winner is written back over the group's rows.  The size table
// is a global integer-atomic histogram (exact, order-independent).
#include "common.h"
#include "wave.h"

namespace {

constexpr int VOTE_MAX_LABELS = 1024;
constexpr int VOTE_WAVES = 4;

__global__ void __launch_bounds__(64 * VOTE_WAVES) group_vote_kernel(const int* __restrict__ labels, const long long* __restrict__ offsets,
                                                                     long long n_groups, int n_labels, int* __restrict__ group_label,
                                                                     int* __restrict__ top_count, int* __restrict__ assignment,
                                                                     long long* __restrict__ size_hist, int max_len, int* __restrict__ bad) {
    __shared__ int hist[VOTE_WAVES][VOTE_MAX_LABELS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* h = hist[wave];
    const long long wave0 = (long long)blockIdx.x * VOTE_WAVES + wave;
    const long long n_waves = (long long)gridDim.x * VOTE_WAVES;
    for (long long g = wave0; g < n_groups; g += n_waves) {
        const long long b = offsets[g], e = offsets[g + 1];
        for (int c = lane; c < n_labels; c += 64) h[c] = 0;
        __builtin_amdgcn_wave_barrier();
        bool out_of_range = false;
        for (long long i = b + lane; i < e; i += 64) {
            const int l = labels[i];
            if (l >= 0 && l < n_labels) atomicAdd(&h[l], 1);
            else out_of_range = true;
        }
        __builtin_amdgcn_wave_barrier();
        // largest count, lowest label on ties
        int best_cnt = -1, best_lab = 0x7fffffff;
        for (int c = lane; c < n_labels; c += 64) {
            const int v = h[c];
            if (v > best_cnt) { best_cnt = v; best_lab = c; }           // ascending c per lane: the first maximum stays
        }
        int m = best_cnt;
        m = max(m, dpp_i<DPP_XOR1>(m));
        m = max(m, dpp_i<DPP_XOR2>(m));
        m = max(m, dpp_i<DPP_HALF_MIRROR>(m));
        m = max(m, dpp_i<DPP_ROW_MIRROR>(m));
        m = max(max(__builtin_amdgcn_readlane(m, 0), __builtin_amdgcn_readlane(m, 16)),
                max(__builtin_amdgcn_readlane(m, 32), __builtin_amdgcn_readlane(m, 48)));
        const int winner = group_min<64>(best_cnt == m ? best_lab : 0x7fffffff);
        if (out_of_range && bad) *bad = 1;
        if (lane == 0) {
            group_label[g] = winner;
            top_count[g] = m;
            const long long len = e - b;
            if (size_hist && winner < n_labels && len <= max_len)
                atomicAdd(reinterpret_cast<unsigned long long*>(size_hist + (long long)winner * (max_len + 1) + len), 1ULL);
        }
        if (assignment)
            for (long long i = b + lane; i < e; i += 64) assignment[i] = winner;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int clava_group_vote_i32(const int* labels, const long long* group_offsets, long long n_groups, int n_labels,
                                    int* group_label, int* top_count, int* assignment, long long* size_hist, int max_len,
                                    int* bad_label_flag, void* stream) {
    CLAVA_CHECK_ARG(labels && group_offsets && group_label && top_count && n_groups > 0, "group_vote: null argument or no groups");
    CLAVA_CHECK_ARG(n_labels > 0 && n_labels <= VOTE_MAX_LABELS, "group_vote: n_labels=%d outside [1, %d]", n_labels, VOTE_MAX_LABELS);
    CLAVA_CHECK_ARG(!size_hist || max_len >= 0, "group_vote: negative max_len");
    const long long blocks = (n_groups + VOTE_WAVES - 1) / VOTE_WAVES;
    hipLaunchKernelGGL(group_vote_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(64 * VOTE_WAVES), 0, (hipStream_t)stream,
                       labels, group_offsets, n_groups, n_labels, group_label, top_count, assignment, size_hist, max_len, bad_label_flag);
    CLAVA_CHECK_LAUNCH("clava_group_vote_i32");
    return CLAVA_OK;
}
