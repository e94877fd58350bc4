// fps.hip -- farthest point sampling for gfx950 (MI355X).
//
// Replaces farthest_point_sampling_cuda_kernel / _launcher
//   (/root/reference/libs/pointops/src/sampling/sampling_cuda_kernel.cu:15-171).
//
// Design (not a translation of the reference's 11-barrier shared-memory ladder):
//  * one workgroup per cloud; every point and its running min-distance live in VGPRs for the
//    whole kernel (xyz is read from HBM exactly once: 12 B/point), a float4 copy of the cloud
//    sits in LDS only so that the winner's coordinates can be broadcast with one ds_read_b128;
//  * per pick: registers-only distance update, a wave64 DPP reduction (no LDS, no barrier), one
//    8-byte LDS slot per wave, ONE s_barrier, then every thread reduces the <=16 wave slots itself;
//  * the serial chain per pick is ~12 VALU/point + ~10 DPP/readlane + 1 barrier + 2 LDS reads.
//
// Bit-exactness.  The reference's winner is: max distance; among equal maxima the CUDA thread
// whose id is smallest in BIT-REVERSED order (its shared-memory tree keeps the lower slot at
// every level, so bit 0 of tid has the highest priority), and inside one CUDA thread the smallest
// point index (strict '>').  With BS = opt_n_threads(n_max) the reference thread of local point j
// is t = j & (BS-1) and its visiting rank is r = j >> log2(BS), hence
//     prio(j) = brev32(t) | r            (smaller wins; brev32(t) occupies the top log2(BS) bits)
// and the pick is argmax over the 64-bit key (float_bits(d2) << 32) | ~prio(j).  d2 >= +0 always,
// so unsigned order on the bits equals float order.  Any thread<->point mapping gives the same
// answer; ours walks each thread's points in ascending prio so a strict '>' suffices in-thread.
#include "pcm_common.hpp"

#include <math.h>

namespace {

constexpr int kMaxRegPoints = 1024 * 16;

// slot -> i mapping.  Thread u holds local points j = u + i*T.  With q = BS/T = 2^LOGQ reference
// threads folded into one of ours, ascending prio inside the thread means ordering by
// (bitrev_LOGQ(i mod q), i / q); slot s = c' * (PPT/q) + r  <->  i = r*q + bitrev_LOGQ(c').
template <int PPT, int LOGQ>
__host__ __device__ constexpr int slot_to_i(int s)
{
    constexpr int Q = 1 << LOGQ;
    constexpr int RP = PPT / Q;
    const int cp = s / RP, r = s % RP;
    const int c = LOGQ == 2 ? (((cp & 1) << 1) | (cp >> 1)) : cp;  // 2-bit reversal; 0/1 bits: identity
    return r * Q + c;
}

__device__ __forceinline__ uint32_t fps_prio(uint32_t j, uint32_t bs_mask, int L)
{
    return __brev(j & bs_mask) | (j >> L);
}

__device__ __forceinline__ uint32_t fps_unprio(uint32_t prio, int L)
{
    const uint32_t topmask = L ? (~0u << (32 - L)) : 0u;
    const uint32_t t = __brev(prio & topmask);
    const uint32_t r = prio & ~topmask;
    return (r << L) | t;
}

// Workgroup arg-max over (bits, prio): wave DPP reduce -> one LDS slot per wave -> one barrier.
// Returns the winning local point index j (identical in every thread).
template <int W>
__device__ __forceinline__ uint32_t fps_block_argmax(uint32_t bits, uint32_t prio, bool valid,
                                                    unsigned long long *slots, int it, int wave,
                                                    int lane, int L)
{
    const uint32_t m1 = pcm_wave_max_u32(bits);
    const uint32_t cand = (valid && bits == m1) ? prio : 0xFFFFFFFFu;
    const uint32_t p1 = pcm_wave_min_u32(cand);
    unsigned long long *buf = slots + (it & 1) * W;  // double-buffered: one barrier per pick
    if (lane == 0) buf[wave] = ((unsigned long long)m1 << 32) | (unsigned long long)(uint32_t)(~p1);
    __syncthreads();
    unsigned long long key = buf[0];
#pragma unroll
    for (int w = 1; w < W; ++w) {
        const unsigned long long k2 = buf[w];
        key = k2 > key ? k2 : key;
    }
    return fps_unprio(~(uint32_t)key, L);
}

// ---------------------------------------------------------------------------------------------
// Register-resident kernel: N_i <= T*PPT for every cloud.
// ---------------------------------------------------------------------------------------------
template <int T, int PPT, int LOGQ, bool LDS_XYZ>
__global__ __launch_bounds__(T) void pcm_fps_reg_kernel(const float *__restrict__ xyz,
                                                        const int *__restrict__ offset,
                                                        const int *__restrict__ new_offset,
                                                        int *__restrict__ idx, int L)
{
    constexpr int W = T / PCM_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);  // 2*W entries
    float4 *lxyz = reinterpret_cast<float4 *>(smem + ((2 * W * 8 + 15) / 16) * 16);

    const int bid = blockIdx.x;
    const int u = threadIdx.x, lane = u & 63, wave = u >> 6;
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    const int end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1];
    const int end_m = new_offset[bid];
    const int N = end_n - start_n, M = end_m - start_m;
    if (M <= 0) return;
    if (N <= 0) {  // reference: every thread contributes (-1, start_n)
        for (int j = u; j < M; j += T) idx[start_m + j] = start_n;
        return;
    }
    if (u == 0) idx[start_m] = start_n;

    float px[PPT], py[PPT], pz[PPT], md[PPT];
    const float *cloud = xyz + (size_t)start_n * 3;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int j = u + slot_to_i<PPT, LOGQ>(s) * T;
        if (j < N) {
            px[s] = cloud[j * 3 + 0];
            py[s] = cloud[j * 3 + 1];
            pz[s] = cloud[j * 3 + 2];
            md[s] = 1e10f;  // functions/sampling.py:18 pre-fill
            if (LDS_XYZ) lxyz[j] = make_float4(px[s], py[s], pz[s], 0.f);
        } else {
            px[s] = py[s] = pz[s] = 0.f;
            md[s] = -1.f;  // min(d, -1) = -1 is never > best
        }
    }
    float ox = cloud[0], oy = cloud[1], oz = cloud[2];
    if (LDS_XYZ) __syncthreads();

    const uint32_t bs_mask = (1u << L) - 1u;
    for (int it = 1; it < M; ++it) {
        float best = -1.f;
        int bs = 0;
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const float d = pcm_sqdist(px[s], py[s], pz[s], ox, oy, oz);
            const float d2 = d < md[s] ? d : md[s];
            md[s] = d2;
            const bool g = d2 > best;
            best = g ? d2 : best;
            bs = g ? s : bs;
        }
        const bool valid = best >= 0.f;
        // bs -> i: same closed form as slot_to_i, on a runtime slot
        int i;
        if (LOGQ == 0) {
            i = bs;
        } else {
            constexpr int Q = 1 << LOGQ;
            constexpr int RP = PPT / Q;
            const int cp = bs / RP, r = bs % RP;
            const int c = LOGQ == 2 ? (((cp & 1) << 1) | (cp >> 1)) : cp;
            i = r * Q + c;
        }
        const uint32_t j = (uint32_t)(u + i * T);
        const uint32_t bits = valid ? __float_as_uint(best) : 0u;
        const uint32_t jw = fps_block_argmax<W>(bits, fps_prio(j, bs_mask, L), valid, slots, it, wave, lane, L);
        if (LDS_XYZ) {
            const float4 o = lxyz[jw];
            ox = o.x, oy = o.y, oz = o.z;
        } else {
            ox = cloud[(size_t)jw * 3 + 0], oy = cloud[(size_t)jw * 3 + 1], oz = cloud[(size_t)jw * 3 + 2];
        }
        if (u == 0) idx[start_m + it] = start_n + (int)jw;
    }
}

// ---------------------------------------------------------------------------------------------
// Any-size kernel (N_i > 16384): running min-distance in the caller's `tmp` (pre-filled 1e10f,
// like the reference), xyz re-read from L2 each pick.  T = 1024 = BS, so thread u walks its points
// j = u, u+1024, ... in ascending prio.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void pcm_fps_big_kernel(const float *__restrict__ xyz,
                                                           const int *__restrict__ offset,
                                                           const int *__restrict__ new_offset,
                                                           float *__restrict__ tmp,
                                                           int *__restrict__ idx, int L)
{
    constexpr int T = 1024, W = T / PCM_WAVE;
    __shared__ unsigned long long slots[2 * W];
    const int bid = blockIdx.x;
    const int u = threadIdx.x, lane = u & 63, wave = u >> 6;
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    const int end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1];
    const int end_m = new_offset[bid];
    const int N = end_n - start_n, M = end_m - start_m;
    if (M <= 0) return;
    if (N <= 0) {
        for (int j = u; j < M; j += T) idx[start_m + j] = start_n;
        return;
    }
    if (u == 0) idx[start_m] = start_n;
    const float *cloud = xyz + (size_t)start_n * 3;
    float *mind = tmp + start_n;
    float ox = cloud[0], oy = cloud[1], oz = cloud[2];
    const uint32_t bs_mask = (1u << L) - 1u;
    for (int it = 1; it < M; ++it) {
        float best = -1.f;
        int bj = 0;
        for (int j = u; j < N; j += T) {
            const float d = pcm_sqdist(cloud[(size_t)j * 3 + 0], cloud[(size_t)j * 3 + 1], cloud[(size_t)j * 3 + 2], ox, oy, oz);
            const float t = mind[j];
            const float d2 = d < t ? d : t;
            mind[j] = d2;
            const bool g = d2 > best;
            best = g ? d2 : best;
            bj = g ? j : bj;
        }
        const bool valid = best >= 0.f;
        const uint32_t bits = valid ? __float_as_uint(best) : 0u;
        const uint32_t jw = fps_block_argmax<W>(bits, fps_prio((uint32_t)bj, bs_mask, L), valid, slots, it, wave, lane, L);
        ox = cloud[(size_t)jw * 3 + 0], oy = cloud[(size_t)jw * 3 + 1], oz = cloud[(size_t)jw * 3 + 2];
        if (u == 0) idx[start_m + it] = start_n + (int)jw;
    }
}

template <int T, int PPT, int LOGQ, bool LDS_XYZ>
int launch_reg(int b, const float *xyz, const int *offset, const int *new_offset, int *idx, int L, hipStream_t st)
{
    constexpr int W = T / PCM_WAVE;
    const size_t lds = ((2 * W * 8 + 15) / 16) * 16 + (LDS_XYZ ? (size_t)T * PPT * 16 : 0);
    auto k = pcm_fps_reg_kernel<T, PPT, LOGQ, LDS_XYZ>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pcm_status(e);
    }
    hipLaunchKernelGGL(k, dim3(b), dim3(T), lds, st, xyz, offset, new_offset, idx, L);
    return PCM_LAUNCH_STATUS();
}

}  // namespace

extern "C" int pcm_opt_n_threads(int work_size)
{
    // cuda_utils.h:11-14, same double-precision log ratio and truncation.
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

extern "C" int pcm_farthest_point_sampling_hip(int b, int n, const float *xyz, const int *offset,
                                               const int *new_offset, float *tmp, int *idx, void *stream)
{
    if (b < 0 || n < 1) return b == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (b == 0) return PCM_OK;
    hipStream_t st = (hipStream_t)stream;
    const int BS = pcm_opt_n_threads(n);
    int L = 0;
    while ((1 << L) < BS) ++L;

    if (n <= 256 * 16) {
        const int need = (n + 255) / 256;
        if (BS <= 256) {  // n < 512: BS <= T, plain ascending-i slots
            if (need <= 1) return launch_reg<256, 1, 0, true>(b, xyz, offset, new_offset, idx, L, st);
            return launch_reg<256, 2, 0, true>(b, xyz, offset, new_offset, idx, L, st);
        }
        if (BS == 512) {  // 512 <= n < 1024: two reference threads per thread
            if (need <= 2) return launch_reg<256, 2, 1, true>(b, xyz, offset, new_offset, idx, L, st);
            return launch_reg<256, 4, 1, true>(b, xyz, offset, new_offset, idx, L, st);
        }
        // BS == 1024: four reference threads per thread
        if (need <= 4) return launch_reg<256, 4, 2, true>(b, xyz, offset, new_offset, idx, L, st);
        if (need <= 8) return launch_reg<256, 8, 2, true>(b, xyz, offset, new_offset, idx, L, st);
        return launch_reg<256, 16, 2, true>(b, xyz, offset, new_offset, idx, L, st);
    }
    if (n <= 1024 * 8) return launch_reg<1024, 8, 0, true>(b, xyz, offset, new_offset, idx, L, st);
    if (n <= kMaxRegPoints) return launch_reg<1024, 16, 0, false>(b, xyz, offset, new_offset, idx, L, st);
    if (tmp == nullptr) return PCM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pcm_fps_big_kernel, dim3(b), dim3(1024), 0, st, xyz, offset, new_offset, tmp, idx, L);
    return PCM_LAUNCH_STATUS();
}
