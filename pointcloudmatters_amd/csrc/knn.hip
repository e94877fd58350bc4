// knn.hip -- k-nearest-neighbour query for gfx950 (MI355X).
//
// Replaces knn_query_cuda_kernel / _launcher
//   (/root/reference/libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:15-112).
//
// The reference runs one thread per query with a 128-entry max-heap in scratch memory.  Here:
//
//  fast kernel   one wave64 per query.  Lane l looks at point (chunk*64 + l): coalesced xyz reads,
//                one distance per lane, then `ballot(d2 < tau)` picks the few lanes that can enter
//                the result.  The running result is a sorted list of nsample+1 (d2, idx) pairs held
//                ACROSS LANES (lane s = s-th smallest); an insertion is a ballot/popcount for the
//                position plus one wave_shr DPP shift -- no LDS, no scratch, no divergence.
//                Candidates are consumed in ascending point index with a strict '<' against the
//                current (nsample+1)-th distance, so the list is exactly the lexicographic
//                (d2, idx) top-(nsample+1).
//
//  exactness     When the nsample+1 smallest d2 are pairwise distinct, the reference's output
//                (heap-select, then heap-sort ascending) is uniquely determined and equals the
//                first nsample list entries.  If two of them tie exactly, the reference's result
//                depends on its heap's history (which of two equal maxima sits at the root when
//                one must be evicted, and the unstable heap-sort order).  The fast kernel marks
//                such queries (dist2[q][0] = -1) and the exact kernel below re-runs ONLY those with
//                the reference's literal algorithm.  => bit-exact always, fast when ties are rare.
//
//  exact kernel  one thread per marked query, literal reheap/heap_sort (:15-42, :86-103).
//                Also serves nsample in 64..128, which does not fit the cross-lane list.
#include "pcm_common.hpp"

namespace {

constexpr int kFastMaxNsample = 63;  // list capacity nsample+1 <= 64 lanes

__global__ __launch_bounds__(256) void pcm_knn_fast_kernel(int b, int m, int nsample,
                                                           const float *__restrict__ xyz,
                                                           const float *__restrict__ new_xyz,
                                                           const int *__restrict__ offset,
                                                           const int *__restrict__ new_offset,
                                                           int *__restrict__ idx,
                                                           float *__restrict__ dist2)
{
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int K1 = nsample + 1;
    const uint32_t PAD = __float_as_uint(1e10f);

    for (int q = blockIdx.x * waves_per_block + wave_in_block; q < m; q += gridDim.x * waves_per_block) {
        const int bt = pcm_cloud_of(q, new_offset, b);
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float qx = new_xyz[(size_t)q * 3 + 0];
        const float qy = new_xyz[(size_t)q * 3 + 1];
        const float qz = new_xyz[(size_t)q * 3 + 2];

        // sorted list across lanes; lanes >= K1 hold a sentinel that never moves
        uint32_t ld = lane < K1 ? PAD : 0xFFFFFFFFu;
        int li = -1;
        uint32_t tau = PAD;  // d2 bits of list entry K1-1 (wave-uniform)

        for (int base = start; base < end; base += 64) {
            const int p = base + lane;
            uint32_t db = 0xFFFFFFFFu;
            if (p < end) {
                const float d = pcm_sqdist(qx, qy, qz, xyz[(size_t)p * 3 + 0], xyz[(size_t)p * 3 + 1], xyz[(size_t)p * 3 + 2]);
                db = __float_as_uint(d);  // d >= +0: unsigned order == float order
            }
            unsigned long long cand = __ballot(db < tau);
            while (cand) {
                const int l = __builtin_ctzll(cand);  // ascending lane == ascending point index
                cand &= cand - 1;
                const uint32_t d = __builtin_amdgcn_readlane(db, l);
                if (d < tau) {  // tau may have dropped since the ballot
                    const int pos = __builtin_popcountll(__ballot(ld <= d));  // sentinel lanes never count
                    const uint32_t up_d = pcm_dpp<0x138>(ld);                 // wave_shr:1
                    const uint32_t up_i = pcm_dpp<0x138>((uint32_t)li);
                    const bool shift = lane > pos && lane < K1;
                    ld = shift ? up_d : (lane == pos ? d : ld);
                    li = shift ? (int)up_i : (lane == pos ? base + l : li);
                    tau = __builtin_amdgcn_readlane(ld, K1 - 1);
                }
            }
        }
        // exact-tie detection over the nsample+1 smallest (pads, li == -1, are not ties)
        const uint32_t nd = pcm_dpp<0x130>(ld);                // wave_shl:1 -> lane l sees l+1
        const int ni = (int)pcm_dpp<0x130>((uint32_t)li);
        const bool tie = lane < K1 - 1 && ld == nd && li >= 0 && ni >= 0;
        const bool any_tie = __ballot(tie) != 0ull;
        if (lane < nsample) {
            idx[(size_t)q * nsample + lane] = li;
            dist2[(size_t)q * nsample + lane] = (lane == 0 && any_tie) ? -1.f : __uint_as_float(ld);
        }
    }
}

// Literal reference algorithm for the queries the fast kernel marked (or for all, if all_queries).
__global__ __launch_bounds__(64) void pcm_knn_exact_kernel(int b, int m, int nsample, int all_queries,
                                                          const float *__restrict__ xyz,
                                                          const float *__restrict__ new_xyz,
                                                          const int *__restrict__ offset,
                                                          const int *__restrict__ new_offset,
                                                          int *__restrict__ idx,
                                                          float *__restrict__ dist2)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < m; q += gridDim.x * blockDim.x) {
        if (!all_queries && !(dist2[(size_t)q * nsample] < 0.f)) continue;
        const int bt = pcm_cloud_of(q, new_offset, b);
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float qx = new_xyz[(size_t)q * 3 + 0];
        const float qy = new_xyz[(size_t)q * 3 + 1];
        const float qz = new_xyz[(size_t)q * 3 + 2];
        float bd[PCM_KNN_MAX_NSAMPLE];
        int bi[PCM_KNN_MAX_NSAMPLE];
        for (int i = 0; i < nsample; ++i) {
            bd[i] = 1e10f;
            bi[i] = -1;
        }
        auto reheap = [&](int k) {
            int root = 0, child = 1;
            while (child < k) {
                if (child + 1 < k && bd[child + 1] > bd[child]) child++;
                if (bd[root] > bd[child]) return;
                const float td = bd[root];
                const int ti = bi[root];
                bd[root] = bd[child];
                bi[root] = bi[child];
                bd[child] = td;
                bi[child] = ti;
                root = child;
                child = root * 2 + 1;
            }
        };
        for (int i = start; i < end; ++i) {
            const float d2 = pcm_sqdist(qx, qy, qz, xyz[(size_t)i * 3 + 0], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2]);
            if (d2 < bd[0]) {
                bd[0] = d2;
                bi[0] = i;
                reheap(nsample);
            }
        }
        for (int i = nsample - 1; i > 0; --i) {
            const float td = bd[0];
            const int ti = bi[0];
            bd[0] = bd[i];
            bi[0] = bi[i];
            bd[i] = td;
            bi[i] = ti;
            reheap(i);
        }
        for (int i = 0; i < nsample; ++i) {
            idx[(size_t)q * nsample + i] = bi[i];
            dist2[(size_t)q * nsample + i] = bd[i];
        }
    }
}

}  // namespace

extern "C" int pcm_knn_query_b_hip(int b, int m, int nsample, const float *xyz, const float *new_xyz,
                                   const int *offset, const int *new_offset, int *idx, float *dist2,
                                   void *stream)
{
    if (m < 0 || nsample < 1 || nsample > PCM_KNN_MAX_NSAMPLE) return PCM_ERR_BAD_ARG;
    if (m == 0) return PCM_OK;
    hipStream_t st = (hipStream_t)stream;
    const int exact_blocks = (m + 63) / 64 < 4096 ? (m + 63) / 64 : 4096;
    if (nsample <= kFastMaxNsample) {
        const int waves_per_block = 4;
        int blocks = (m + waves_per_block - 1) / waves_per_block;
        if (blocks > 256 * 8) blocks = 256 * 8;  // 8 workgroups per CU, grid-stride beyond
        hipLaunchKernelGGL(pcm_knn_fast_kernel, dim3(blocks), dim3(64 * waves_per_block), 0, st, b, m, nsample, xyz, new_xyz,
                           offset, new_offset, idx, dist2);
        int rc = PCM_LAUNCH_STATUS();
        if (rc) return rc;
        hipLaunchKernelGGL(pcm_knn_exact_kernel, dim3(exact_blocks), dim3(64), 0, st, b, m, nsample, 0, xyz, new_xyz, offset,
                           new_offset, idx, dist2);
        return PCM_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(pcm_knn_exact_kernel, dim3(exact_blocks), dim3(64), 0, st, b, m, nsample, 1, xyz, new_xyz, offset,
                       new_offset, idx, dist2);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_knn_query_hip(int m, int nsample, const float *xyz, const float *new_xyz,
                                 const int *offset, const int *new_offset, int *idx, float *dist2,
                                 void *stream)
{
    return pcm_knn_query_b_hip(0, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, stream);
}
