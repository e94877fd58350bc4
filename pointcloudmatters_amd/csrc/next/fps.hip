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
#include <stdlib.h>

namespace {

constexpr int kMaxRegPoints = 1024 * 16;

// slot -> i mapping.  Thread u holds local points j = u + i*T.  With q = BS/T = 2^LOGQ reference
// threads folded into one of ours, ascending prio inside the thread means ordering by
// (bitrev_LOGQ(i mod q), i / q); slot s = c' * (PPT/q) + r  <->  i = r*q + bitrev_LOGQ(c').
template <int NB>
__host__ __device__ constexpr int bitrev_n(int v)
{
    int r = 0;
    for (int b = 0; b < NB; ++b) r |= ((v >> b) & 1) << (NB - 1 - b);
    return r;
}

template <int PPT, int LOGQ>
__host__ __device__ constexpr int slot_to_i(int s)
{
    constexpr int Q = 1 << LOGQ;
    constexpr int RP = PPT / Q;
    const int cp = s / RP, r = s % RP;
    return r * Q + bitrev_n<LOGQ>(cp);
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
// Returns the winning local point index j (identical in every thread).  (any-size kernel)
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

typedef float fps_f2 __attribute__((ext_vector_type(2)));  // a point pair: packed fp32 in this file (plain forms only, see the pick loop); csrc/fps.hip (frozen, NO_PK): two scalar chains

// Wave-wide unsigned max in 6 DPP-fused instructions + 1 readlane (the builtin form costs a v_mov + s_nop + v_mov_dpp +
// v_max per step and 4 readlanes): quad swaps, row_half_mirror, row_mirror leave every row's maximum in all of its lanes,
// row_bcast:15 / row_bcast:31 (GFX9 DPP, present on gfx950) fold the four rows into lane 63.
__device__ __forceinline__ uint32_t fps_wave_max_fast(uint32_t v)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---------------------------------------------------------------------------------------------
// Register-resident kernel: N_i <= T*PPT for every cloud.  The pick loop is bound by instruction issue of ONE wave per
// SIMD (~6 clocks per dependent instruction), so it is written to be short:
//  (a) distance update on point PAIRS, written on two-element vectors (sub, mul, add, mul, add, one rounding at a time; the build
//      compiles them to scalar v_sub / v_mul / v_add -- packed fp32 instructions are disabled, Makefile NO_PK -- so the pair form only
//      keeps two independent chains in flight), v_min for the running distance, one compare + two selects for the thread's candidate;
//  (b) wave maximum of the candidates' distance bits: 6 fused DPP instructions + 1 readlane;
//  (c) a ballot finds the owning lane; only an exact tie inside the wave (wave-uniform branch, rare) computes the
//      reference's tie order (prio) and reduces it;
//  (d) the OWNING LANE publishes {distance key} and {local index, x, y, z} in the wave's LDS slot -- ONE s_barrier --
//  (e) every thread takes the maximum of the W keys; a unique maximum (usual) selects the slot directly, an exact tie
//      across waves compares the reference's prio of the tied candidates; one more broadcast read fetches index + xyz.
// The cloud itself stays in registers (PPT <= 4: the candidate's xyz comes out of registers in the owning lane; larger
// PPT: the owning lane reads it from the LDS copy), so nothing is ever re-read from HBM.
// ---------------------------------------------------------------------------------------------
template <int T, int PPT, int LOGQ, bool LDS_XYZ>
__global__ __launch_bounds__(T) void pcm_fps_reg_kernel(const float *__restrict__ xyz,
                                                        const int *__restrict__ offset,
                                                        const int *__restrict__ new_offset,
                                                        int *__restrict__ idx, int L)
{
    constexpr int W = T / PCM_WAVE;
    constexpr bool TRACK = PPT <= 4;               // candidate xyz taken from registers
    constexpr bool STAGE = LDS_XYZ && !TRACK;      // float4 copy of the cloud in LDS for the owning lane
    constexpr int NP = (PPT + 1) / 2;              // point pairs per thread
    constexpr int WP = (W + 3) / 4 * 4;            // keys padded to whole 16-byte reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem);                 // [2][WP]   0 = wave without candidate, else bits + 1
    float4 *cand = reinterpret_cast<float4 *>(smem + 2 * WP * 4);        // [2][W]    (local index, x, y, z)
    float4 *lxyz = cand + 2 * W;                                         // [N] when STAGE

    asm volatile("" ::"s"(xyz), "s"(offset), "s"(new_offset), "s"(idx), "s"(L));  // "Kernel heads", pcm_common.hpp
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
    if (u < 2 * WP) keys[u] = 0u;  // padding slots never win

    fps_f2 px[NP], py[NP], pz[NP];
    float md[2 * NP];
    const float *cloud = xyz + (size_t)start_n * 3;
    // Two passes: ALL of the thread's points are requested first -- branch-free: a slot past the cloud reads point N - 1 and is zeroed by a
    // select --, the LDS copy is written from the registers afterwards.  (One pass with the loads inside `if (j < N)` -- load, stage, next
    // point -- compiled to a full `s_waitcnt vmcnt(0)` per point: PPT round trips in series before the first pick.)
#pragma unroll
    for (int s = 0; s < 2 * NP; ++s) {
        const int j = s < PPT ? u + slot_to_i<PPT, LOGQ>(s < PPT ? s : 0) * T : N;
        const bool in = j < N;
        const int jc = in ? j : N - 1;
        const float x = cloud[jc * 3 + 0], y = cloud[jc * 3 + 1], z = cloud[jc * 3 + 2];
        md[s] = in ? 1e10f : -1.f;  // functions/sampling.py:18 pre-fill; min(d, -1) = -1 is never > best
        px[s / 2][s % 2] = in ? x : 0.f, py[s / 2][s % 2] = in ? y : 0.f, pz[s / 2][s % 2] = in ? z : 0.f;
    }
    float ox = cloud[0], oy = cloud[1], oz = cloud[2];
    if (STAGE) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int j = u + slot_to_i<PPT, LOGQ>(s) * T;
            if (j < N) lxyz[j] = make_float4(px[s / 2][s % 2], py[s / 2][s % 2], pz[s / 2][s % 2], 0.f);
        }
    }
    __syncthreads();

    const uint32_t bs_mask = (1u << L) - 1u;
    int *out = idx + start_m;
    for (int it = 1; it < M; ++it) {
        float best = -1.f;
        int bs = 0;
        // PACKED fp32 again, in its safe forms only (round 6; this file alone is built without the Makefile's NO_PK).  The hazard of this
        // stack (tools/dbg/pk_hazard: measured on hardware in round 4) is specific to v_pk_*_f32 with OP_SEL set -- what
        // the compiler emits for `pair - splat(scalar)`; plain forms, op_sel_hi and neg modifiers were exact in every run beside a GEMM graph.
        // The pick's centre is therefore made a REAL pair in two registers and hidden from the compiler (the empty asm): it can no longer
        // fold the splat into an operand-half select, and every packed instruction of this file is plain or neg-modified
        // (tests/test_build_flags.py checks the assembly).  Same roundings as the scalar chain: a packed add / mul rounds each half on its own.
        // 8 packed instead of 16 scalar operations per point pair: the per-pick issue budget of DESIGN.md section 10 item 6.
        fps_f2 o2x = (fps_f2)(ox), o2y = (fps_f2)(oy), o2z = (fps_f2)(oz);
        asm volatile("" : "+v"(o2x), "+v"(o2y), "+v"(o2z));
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            // pcm_sqdist on two points at once: (a-b)*(a-b) for x, y, z summed left to right, every op rounded on its own
            const fps_f2 dx = px[k] - o2x, dy = py[k] - o2y, dz = pz[k] - o2z;
            fps_f2 d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = 2 * k + h;
                if (s >= PPT) break;
                const float d2 = __builtin_fminf(d[h], md[s]);  // no NaNs here: same value as d < md ? d : md
                md[s] = d2;
                const bool g = d2 > best;
                best = g ? d2 : best;
                bs = g ? s : bs;
            }
        }
        const bool valid = best >= 0.f;
        const uint32_t key = valid ? __float_as_uint(best) + 1u : 0u;  // d2 >= +0: unsigned order == float order; 0 = no candidate
        // bs -> local index j: same closed form as slot_to_i, on a runtime slot
        int i;
        if (LOGQ == 0) {
            i = bs;
        } else {
            constexpr int Q = 1 << LOGQ;
            constexpr int RP = PPT / Q;
            const int cp = bs / RP, r = bs % RP;
            i = r * Q + bitrev_n<LOGQ>(cp);
        }
        const uint32_t j = (uint32_t)(u + i * T);
        // ---- wave level: maximum, then its owner
        const uint32_t m1 = fps_wave_max_fast(key);
        unsigned long long own = __ballot(key == m1);
        if (m1 != 0u && __popcll(own) > 1) {  // exact distance tie inside the wave: the reference's order decides
            const uint32_t prio = (key == m1) ? fps_prio(j, bs_mask, L) : 0xFFFFFFFFu;
            const uint32_t p1 = pcm_wave_min_u32(prio);
            own = __ballot(prio == p1);
        }
        const int buf = it & 1;  // double-buffered slots: one barrier per pick
        if (lane == (int)__builtin_ctzll(own)) {
            float bx = 0.f, by = 0.f, bz = 0.f;
            if (TRACK) {
#pragma unroll
                for (int s = 0; s < PPT; ++s)
                    if (s == bs) bx = px[s / 2][s % 2], by = py[s / 2][s % 2], bz = pz[s / 2][s % 2];
            } else if (STAGE) {
                const float4 o = lxyz[j];
                bx = o.x, by = o.y, bz = o.z;
            } else {
                bx = cloud[(size_t)j * 3 + 0], by = cloud[(size_t)j * 3 + 1], bz = cloud[(size_t)j * 3 + 2];
            }
            keys[buf * WP + wave] = m1;
            cand[buf * W + wave] = make_float4(__uint_as_float(j), bx, by, bz);
        }
        __syncthreads();
        // ---- workgroup level: maximum of the W keys (broadcast reads), then the winning slot
        uint32_t kw[WP];
#pragma unroll
        for (int w4 = 0; w4 < WP; w4 += 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(keys + buf * WP + w4);
            kw[w4] = q.x, kw[w4 + 1] = q.y, kw[w4 + 2] = q.z, kw[w4 + 3] = q.w;
        }
        uint32_t gmax = kw[0];
#pragma unroll
        for (int w = 1; w < W; ++w) gmax = kw[w] > gmax ? kw[w] : gmax;
        int wsel = 0, nmatch = 0;
#pragma unroll
        for (int w = W - 1; w >= 0; --w) {
            const bool e = kw[w] == gmax;
            wsel = e ? w : wsel;
            nmatch += e ? 1 : 0;
        }
        // (fetching all W candidates together with the keys and selecting in registers was measured slower: 1381 vs 1234
        //  clocks per pick at N = 1024 -- the extra broadcast reads cost more than the dependent one saves)
        float4 o = cand[buf * W + wsel];
        if (nmatch > 1) {  // exact tie across waves (wave-uniform: every lane read the same keys): smallest prio wins
            uint32_t bestp = fps_prio(__float_as_uint(o.x), bs_mask, L);
            for (int w = wsel + 1; w < W; ++w) {
                if (kw[w] != gmax) continue;
                const float4 c = cand[buf * W + w];
                const uint32_t pw = fps_prio(__float_as_uint(c.x), bs_mask, L);
                if (pw < bestp) bestp = pw, o = c;
            }
        }
        ox = o.y, oy = o.z, oz = o.w;
        if (u == 0) out[it] = start_n + (int)__float_as_uint(o.x);
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
    const size_t lds = (size_t)2 * ((W + 3) / 4 * 4) * 4 + (size_t)2 * W * 16 + ((LDS_XYZ && PPT > 4) ? (size_t)T * PPT * 16 : 0);
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
        // n <= 1024: 128 threads x 8 points (two waves to synchronise per pick instead of four) measured 476 vs 498 ns / pick
        // at 8 x 1024 and 482 vs 562 at 128 x 1024; ONE wave x 16 points is slower again (599: the in-thread chain dominates)
        static const int small_t = pcm_mb_switch("PCM_FPS_SMALL_T", 128);  // A/B switch for tools/mb
        if (need <= 4 && small_t == 64) return launch_reg<64, 16, 4, true>(b, xyz, offset, new_offset, idx, L, st);
        if (need <= 4 && small_t == 128) return launch_reg<128, 8, 3, true>(b, xyz, offset, new_offset, idx, L, st);
        if (need <= 4) return launch_reg<256, 4, 2, true>(b, xyz, offset, new_offset, idx, L, st);
        static const int big_t = pcm_mb_switch("PCM_FPS_BIG_T", 256);  // A/B switch for tools/mb (2048 < n <= 4096)
        if (need <= 8) {
            if (big_t == 512) return launch_reg<512, 4, 1, true>(b, xyz, offset, new_offset, idx, L, st);
            if (big_t == 128) return launch_reg<128, 16, 3, true>(b, xyz, offset, new_offset, idx, L, st);
            return launch_reg<256, 8, 2, true>(b, xyz, offset, new_offset, idx, L, st);
        }
        if (big_t == 512) return launch_reg<512, 8, 1, true>(b, xyz, offset, new_offset, idx, L, st);
        if (big_t == 1024) return launch_reg<1024, 4, 0, true>(b, xyz, offset, new_offset, idx, L, st);
        if (big_t == 128) return launch_reg<128, 32, 3, true>(b, xyz, offset, new_offset, idx, L, st);
        return launch_reg<256, 16, 2, true>(b, xyz, offset, new_offset, idx, L, st);
    }
    if (n <= 1024 * 8) {
        // 4096 < n_max <= 8192 (BS = 1024).  The 1024-thread kernel pays 16 waves' worth of slots per pick (1.24 us / pick on
        // a ragged 8 x ~4096 batch); fewer, fatter threads keep the pick latency of the 4096-point variant:
        // 256 threads x 24 / 32 points (four reference threads folded into one) or 512 x 12 / 16 (two).
        static const int t512 = pcm_mb_switch("PCM_FPS_T512", 0);  // A/B switch for tools/mb
        if (t512 == 2) return launch_reg<1024, 8, 0, true>(b, xyz, offset, new_offset, idx, L, st);
        if (n <= 256 * 24) {
            if (t512) return launch_reg<512, 12, 1, true>(b, xyz, offset, new_offset, idx, L, st);
            return launch_reg<256, 24, 2, true>(b, xyz, offset, new_offset, idx, L, st);
        }
        if (t512) return launch_reg<512, 16, 1, true>(b, xyz, offset, new_offset, idx, L, st);
        return launch_reg<256, 32, 2, true>(b, xyz, offset, new_offset, idx, L, st);
    }
    if (n <= kMaxRegPoints) return launch_reg<1024, 16, 0, false>(b, xyz, offset, new_offset, idx, L, st);
    if (tmp == nullptr) return PCM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pcm_fps_big_kernel, dim3(b), dim3(1024), 0, st, xyz, offset, new_offset, tmp, idx, L);
    return PCM_LAUNCH_STATUS();
}
