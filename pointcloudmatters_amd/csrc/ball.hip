// ball.hip -- ball query (K3) and random ball query (K4) for gfx950 (MI355X).
//
// Replaces ball_query_cuda_kernel (/root/reference/libs/pointops/src/ball_query/
// ball_query_cuda_kernel.cu:58-190) and random_ball_query_cuda_kernel
// (random_ball_query/random_ball_query_cuda_kernel.cu:58-123).  API-only ops (no call site in the
// reference's src/) -- built for drop-in completeness and bit-exactness, not speed.
//
// One wave64 per query: lane l tests point chunk*64+l (coalesced xyz reads) and in-range points
// are compacted with ballot + prefix popcount, which preserves the reference's scan order.
//  * ball query: the reference then calls heap_sort on the candidate array WITHOUT heapifying it
//    (:103) -- a deterministic but unsorted permutation that later feeds the strided subsample, so
//    it must be replayed literally: candidates are staged in LDS (2048 x 8 B per query, the
//    reference's stack-array bound) and lane 0 replays the exact swap/reheap sequence.
//  * random ball query: first nsample hits in `order`; no LDS at all.
#include "pcm_common.hpp"

#include <stdlib.h>

namespace {

__device__ __forceinline__ bool in_ball(float d2, float min_r2, float max_r2)
{
    return (double)d2 <= 1e-5 || (d2 >= min_r2 && d2 < max_r2);  // :91, the 1e-5 test is in double
}

__global__ __launch_bounds__(64) void pcm_ball_query_kernel(int m, int nsample, float min_radius, float max_radius,
                                                           const float *__restrict__ xyz,
                                                           const float *__restrict__ new_xyz,
                                                           const int *__restrict__ offset,
                                                           const int *__restrict__ new_offset, int b,
                                                           int *__restrict__ idx, float *__restrict__ dist2, int only_flagged)
{
    __shared__ float cd[PCM_BALL_MAX_CAND];
    __shared__ int ci[PCM_BALL_MAX_CAND];
    const int lane = threadIdx.x;
    const float max_r2 = max_radius * max_radius;
    const float min_r2 = min_radius * min_radius;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    for (int q = blockIdx.x; q < m; q += gridDim.x) {
        if (only_flagged && idx[(size_t)q * nsample] != -2) continue;  // block-uniform: answered by the lane-per-query kernel
        const int bt = pcm_cloud_of(q, new_offset, b);
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float qx = new_xyz[(size_t)q * 3 + 0];
        const float qy = new_xyz[(size_t)q * 3 + 1];
        const float qz = new_xyz[(size_t)q * 3 + 2];
        int cnt = 0;
        // The scan is a chain of dependent (load -> ballot -> LDS append) rounds on ONE wave per workgroup: latency-
        // bound.  Eight chunks of coordinates are loaded before the first ballot, so eight rounds share one memory wait.
        constexpr int UN = 8;
        for (int base = start; base < end; base += 64 * UN) {
            float d2[UN];
            bool in[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int p = base + u * 64 + lane;
                in[u] = false;
                d2[u] = 0.f;
                if (p < end) {
                    d2[u] = pcm_sqdist(qx, qy, qz, xyz[(size_t)p * 3 + 0], xyz[(size_t)p * 3 + 1], xyz[(size_t)p * 3 + 2]);
                    in[u] = in_ball(d2[u], min_r2, max_r2);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const unsigned long long mask = __ballot(in[u]);
                const int pos = cnt + __builtin_popcountll(mask & lt_mask);
                if (in[u] && pos < PCM_BALL_MAX_CAND) {
                    cd[pos] = d2[u];
                    ci[pos] = base + u * 64 + lane;
                }
                cnt += __builtin_popcountll(mask);
            }
        }
        __syncthreads();
        const bool overflow = cnt > PCM_BALL_MAX_CAND;  // UB in the reference; we emit an empty row
        if (!overflow && cnt > 1 && cnt <= 64) {
            // heap_sort(:33-42) on the un-heapified array, replayed literally with the array held one element per
            // lane: every index below is wave-uniform, so each access is a v_readlane or a v_cndmask (a few cycles)
            // instead of a dependent LDS round trip (the serial LDS replay below was ~90 % of this kernel's time at
            // 10-20 candidates per query).
            float d = lane < cnt ? cd[lane] : 0.f;
            int ix = lane < cnt ? ci[lane] : 0;
#define PCM_RL_F(v, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l))
#define PCM_WL(v, val, l) ((lane == (l)) ? (val) : (v))  // v_cmp + v_cndmask against a uniform lane id
            for (int i = cnt - 1; i > 0; --i) {
                const float d0 = PCM_RL_F(d, 0), di = PCM_RL_F(d, i);
                const int x0 = __builtin_amdgcn_readlane(ix, 0), xi = __builtin_amdgcn_readlane(ix, i);
                d = PCM_WL(d, di, 0);
                d = PCM_WL(d, d0, i);
                ix = PCM_WL(ix, xi, 0);
                ix = PCM_WL(ix, x0, i);
                int root = 0, child = 1;
                float dr = di;  // value now at the root
                int xr = xi;
                while (child < i) {
                    float dc = PCM_RL_F(d, child);
                    if (child + 1 < i) {
                        const float dc1 = PCM_RL_F(d, child + 1);
                        if (dc1 > dc) {
                            child++;
                            dc = dc1;
                        }
                    }
                    if (dr > dc) break;
                    const int xc = __builtin_amdgcn_readlane(ix, child);
                    d = PCM_WL(d, dc, root);
                    d = PCM_WL(d, dr, child);
                    ix = PCM_WL(ix, xc, root);
                    ix = PCM_WL(ix, xr, child);
                    root = child;  // dr / xr travel down with the root
                    child = root * 2 + 1;
                }
            }
#undef PCM_RL_F
#undef PCM_WL
            if (lane < cnt) {
                cd[lane] = d;
                ci[lane] = ix;
            }
        } else if (!overflow && lane == 0) {
            // more than 64 candidates: the same replay, serially in LDS
            for (int i = cnt - 1; i > 0; --i) {
                float td = cd[0];
                int ti = ci[0];
                cd[0] = cd[i];
                ci[0] = ci[i];
                cd[i] = td;
                ci[i] = ti;
                int root = 0, child = 1;
                while (child < i) {
                    if (child + 1 < i && cd[child + 1] > cd[child]) child++;
                    if (cd[root] > cd[child]) break;
                    td = cd[root];
                    ti = ci[root];
                    cd[root] = cd[child];
                    ci[root] = ci[child];
                    cd[child] = td;
                    ci[child] = ti;
                    root = child;
                    child = root * 2 + 1;
                }
            }
        }
        __syncthreads();
        int *oi = idx + (size_t)q * nsample;
        float *od = dist2 + (size_t)q * nsample;
        if (overflow || cnt <= nsample) {
            const int have = overflow ? 0 : cnt;
            for (int i = lane; i < nsample; i += 64) {
                oi[i] = i < have ? ci[i] : -1;
                od[i] = i < have ? cd[i] : 1e10f;
            }
        } else {
            const float sep = (float)cnt / nsample;  // :115
            for (int i = lane; i < nsample; i += 64) {
                const int index = (int)(sep * i);  // :118
                oi[i] = ci[index];
                od[i] = (float)ci[index];  // :120 (sic): the reference stores the index as dist2
            }
        }
        __syncthreads();  // cd/ci are reused by the next query
    }
}

// ---- lane-per-query variant ------------------------------------------------------------------------------------
// The wave-per-query kernel above spends its time in the literal heap_sort replay: O(cnt log cnt) DEPENDENT steps for one
// query at a time (1.45 ms for 65 536 queries with ~67 candidates each).  The replay cannot be parallelised inside a
// query, so it is run for 64 queries at once instead, the way the reference's one-thread-per-query kernel does
// (ball_query_cuda_kernel.cu:58-123), but out of LDS rather than scratch memory:
//   * one wave = 64 consecutive queries, lane = query; the candidates of lane l live in the LDS columns cd[k][l], ci[k][l]
//     (bank = lane: conflict-free whatever k each lane touches);
//   * the cloud is staged through LDS in chunks of 512 points and every lane walks the chunk in scan order (same address
//     in all lanes = LDS broadcast), appending to its own column -- the scan order of the reference by construction;
//   * the heap_sort replay and the output rows are plain per-lane code.
// A block of queries that straddles clouds walks each cloud it touches; a query with more than kLaneCap candidates is
// flagged (idx[q][0] = -2) and redone by the wave-per-query kernel, which handles up to PCM_BALL_MAX_CAND.
constexpr int kLaneCap = 96;    // candidates per query held in LDS: 96 x 64 x (4 + 2) B = 36 KiB per wave (indices cloud-local, 16 bit)
constexpr int kBallChunk = 512;

// (double)d2 <= 1e-5 for a float d2  <=>  d2 <= 1e-5f: 1e-5f = 9.99999974737875e-06 lies below 1e-5 and the next float above it
// lies above 1e-5, so the comparison can stay in single precision (same truth value for every float)
__device__ __forceinline__ bool in_ball_f(float d2, float min_r2, float max_r2)
{
    return d2 <= 1e-5f || (d2 >= min_r2 && d2 < max_r2);
}

__global__ __launch_bounds__(64) void pcm_ball_query_lanes_kernel(int m, int nsample, float min_radius, float max_radius,
                                                                   const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                   const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                   int b, int *__restrict__ idx, float *__restrict__ dist2)
{
    __shared__ float cd[kLaneCap][64];
    __shared__ unsigned short ci[kLaneCap][64];
    __shared__ float4 pts[kBallChunk];
    const int lane = threadIdx.x;
    const float max_r2 = max_radius * max_radius;
    const float min_r2 = min_radius * min_radius;
    for (int qb = blockIdx.x * 64; qb < m; qb += gridDim.x * 64) {
        const int q = qb + lane;
        const bool live = q < m;
        const int qq = live ? q : m - 1;
        const float qx = new_xyz[(size_t)qq * 3 + 0], qy = new_xyz[(size_t)qq * 3 + 1], qz = new_xyz[(size_t)qq * 3 + 2];
        const int c_first = pcm_cloud_of(qb, new_offset, b), c_last = pcm_cloud_of(min(qb + 64, m) - 1, new_offset, b);
        int cnt = 0, my_start = 0;
        for (int c = c_first; c <= c_last; ++c) {
            const int start = c == 0 ? 0 : offset[c - 1], end = offset[c];
            const int qs = c == 0 ? 0 : new_offset[c - 1], qe = new_offset[c];
            const bool mine = live && q >= qs && q < qe;
            if (mine) {
                my_start = start;
                if (end - start > 65535) cnt = kLaneCap + 1;  // cloud-local indices would not fit 16 bits: leave it to the other kernel
            }
            for (int base = start; base < end; base += kBallChunk) {
                const int nch = min(kBallChunk, end - base);
                __syncthreads();  // the previous chunk has been read by every lane
                for (int t = lane; t < kBallChunk; t += 64) {
                    const float *p = xyz + (size_t)(base + (t < nch ? t : 0)) * 3;
                    pts[t] = t < nch ? make_float4(p[0], p[1], p[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                __syncthreads();
                if (mine) {
                    constexpr int UN = 8;  // distances of 8 points in flight before the first (divergent) append
                    for (int t0 = 0; t0 < nch; t0 += UN) {
                        float d2[UN];
#pragma unroll
                        for (int u = 0; u < UN; ++u) {
                            const float4 p = pts[t0 + u];  // same address in every lane: broadcast
                            d2[u] = pcm_sqdist(qx, qy, qz, p.x, p.y, p.z);
                        }
#pragma unroll
                        for (int u = 0; u < UN; ++u) {
                            if (t0 + u < nch && in_ball_f(d2[u], min_r2, max_r2)) {
                                if (cnt < kLaneCap) cd[cnt][lane] = d2[u], ci[cnt][lane] = (unsigned short)(base + t0 + u - start);
                                ++cnt;
                            }
                        }
                    }
                }
            }
        }
        if (live) {
            int *oi = idx + (size_t)q * nsample;
            float *od = dist2 + (size_t)q * nsample;
            if (cnt > kLaneCap) {
                oi[0] = -2;  // redone by the wave-per-query kernel
            } else {
                // heap_sort (:33-42) on the un-heapified candidate array, literally
                for (int i = cnt - 1; i > 0; --i) {
                    float td = cd[0][lane];
                    unsigned short ti = ci[0][lane];
                    cd[0][lane] = cd[i][lane], ci[0][lane] = ci[i][lane];
                    cd[i][lane] = td, ci[i][lane] = ti;
                    int root = 0, child = 1;
                    float dr = cd[0][lane];
                    while (child < i) {
                        float dc = cd[child][lane];
                        if (child + 1 < i) {
                            const float dc1 = cd[child + 1][lane];
                            if (dc1 > dc) child++, dc = dc1;
                        }
                        if (dr > dc) break;
                        const unsigned short xr = ci[root][lane], xc = ci[child][lane];
                        cd[root][lane] = dc, ci[root][lane] = xc;
                        cd[child][lane] = dr, ci[child][lane] = xr;
                        root = child;  // dr travels down with the root
                        child = root * 2 + 1;
                    }
                }
                if (cnt <= nsample) {
                    for (int i = 0; i < nsample; ++i) {
                        oi[i] = i < cnt ? my_start + (int)ci[i][lane] : -1;
                        od[i] = i < cnt ? cd[i][lane] : 1e10f;
                    }
                } else {
                    const float sep = (float)cnt / nsample;  // :115
                    for (int i = 0; i < nsample; ++i) {
                        const int index = (int)(sep * i);  // :118
                        oi[i] = my_start + (int)ci[index][lane];
                        od[i] = (float)(my_start + (int)ci[index][lane]);  // :120 (sic): the reference stores the index as dist2
                    }
                }
            }
        }
    }
}

// ---- split variant (round 4): candidate COLLECTION at full occupancy, heap replay out of LDS in a second, short kernel ----------
// The lane-per-query kernel above keeps 36 KiB of candidate columns in LDS for its whole life, so three waves fit a CU and the
// 4096-point scan -- a chain of dependent instructions -- runs with ONE wave per SIMD: ~290 clocks per point, 1.0 ms for 65 536
// queries.  The scan needs no LDS at all: every lane of a wave (= 64 consecutive queries) tests the SAME point at the same
// time, so the point's coordinates are wave-uniform and come through the scalar cache (s_load), and the rare appends (1-2 %
// of the points) go straight to a workspace in global memory, lane-major so that equal slots of neighbouring queries share
// lines.  With no LDS and ~40 VGPRs a CU holds 32 waves and the scan runs at the VALU rate (~11 vector instructions per point).
// When there are few queries, S = 2 / 4 / 8 waves split each query's cloud into S scan-order segments with their own slot ranges;
// reading the segments back in order restores the reference's scan order.  The replay kernel then loads each query's list into
// LDS columns (lane = query, as above), replays heap_sort-without-heapify literally and writes the rows; queries with more than
// kLaneCap candidates are flagged and redone by the wave-per-query kernel, exactly as before.
// workspace: cnt (S, m) int32 | dist (S * cap_S, m) f32 | index (S * cap_S, m) u16 (cloud-local); cap_S = ball_seg_cap(S) slots per
// segment (a segment that overflows flags its query like a query with more than kLaneCap candidates)
__host__ __device__ inline int ball_seg_cap(int S) { return S == 1 ? kLaneCap : (S == 2 ? 80 : (S == 4 ? 64 : 48)); }
__global__ __launch_bounds__(256) void pcm_ball_collect_kernel(int m, int S, float min_radius, float max_radius,
                                                               const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                               const int *__restrict__ offset, const int *__restrict__ new_offset, int b,
                                                               int *__restrict__ wcnt, float *__restrict__ wd, unsigned short *__restrict__ wi)
{
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));  // global wave: (query block, segment)
    const int nqb = (m + 63) >> 6;
    if (gw >= nqb * S) return;
    const int qblock = gw / S, seg = gw - qblock * S;
    const int qb = qblock * 64;
    const int q = qb + lane;
    const bool live = q < m;
    const int qq = live ? q : m - 1;
    const float qx = new_xyz[(size_t)qq * 3 + 0], qy = new_xyz[(size_t)qq * 3 + 1], qz = new_xyz[(size_t)qq * 3 + 2];
    const float max_r2 = max_radius * max_radius, min_r2 = min_radius * min_radius;
    const int last = qb + 63 < m ? qb + 63 : m - 1;
    const int c_first = pcm_cloud_of(qb, new_offset, b), c_last = pcm_cloud_of(last, new_offset, b);
    int cnt = 0;
    const int cap = ball_seg_cap(S);
    float *myd = wd + (size_t)seg * cap * m + qq;
    unsigned short *myi = wi + (size_t)seg * cap * m + qq;
    for (int c = c_first; c <= c_last; ++c) {  // wave-uniform: a block of 64 queries rarely straddles clouds
        const int start = c == 0 ? 0 : offset[c - 1], end = offset[c];
        const int qs = c == 0 ? 0 : new_offset[c - 1], qe = new_offset[c];
        const bool mine = live && q >= qs && q < qe;
        const long len = end - start;
        if (len > 65535) {  // cloud-local indices would not fit 16 bits: leave these queries to the wave-per-query kernel
            if (mine) cnt = cap + 1;
            continue;
        }
        const int p0 = start + (int)(len * seg / S), p1 = start + (int)(len * (seg + 1) / S);
        if (!mine) continue;  // (lanes of other clouds sit this cloud out: exec mask, the loop bounds stay uniform)
        // p is wave-uniform: the coordinates come through the scalar cache, eight points (three s_load_dwordx8) in flight before the
        // first use; per point 3 v_sub + 5 v_mul / v_add against SGPR operands and three compares
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float c8[24];
            const float *src = xyz + (size_t)p * 3;
#pragma unroll
            for (int i = 0; i < 24; ++i) c8[i] = src[i];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d2 = pcm_sqdist(qx, qy, qz, c8[3 * u], c8[3 * u + 1], c8[3 * u + 2]);
                if (in_ball_f(d2, min_r2, max_r2)) {
                    if (cnt < cap) {
                        myd[(size_t)cnt * m] = d2;
                        myi[(size_t)cnt * m] = (unsigned short)(p + u - start);
                    }
                    ++cnt;
                }
            }
        }
        for (; p < p1; ++p) {
            const float d2 = pcm_sqdist(qx, qy, qz, xyz[(size_t)p * 3 + 0], xyz[(size_t)p * 3 + 1], xyz[(size_t)p * 3 + 2]);
            if (in_ball_f(d2, min_r2, max_r2)) {
                if (cnt < cap) {
                    myd[(size_t)cnt * m] = d2;
                    myi[(size_t)cnt * m] = (unsigned short)(p - start);
                }
                ++cnt;
            }
        }
    }
    if (live) wcnt[(size_t)seg * m + q] = cnt;
}

__global__ __launch_bounds__(64) void pcm_ball_replay_kernel(int m, int S, int nsample, const int *__restrict__ offset,
                                                             const int *__restrict__ new_offset, int b, const int *__restrict__ wcnt,
                                                             const float *__restrict__ wd, const unsigned short *__restrict__ wi,
                                                             int *__restrict__ idx, float *__restrict__ dist2)
{
    __shared__ float cd[kLaneCap][64];
    __shared__ unsigned short ci[kLaneCap][64];
    const int lane = threadIdx.x;
    for (int qb = blockIdx.x * 64; qb < m; qb += gridDim.x * 64) {
        const int q = qb + lane;
        const bool live = q < m;
        if (!live) continue;  // no barrier below: every lane owns its LDS column
        const int c = pcm_cloud_of(q, new_offset, b);
        const int my_start = c == 0 ? 0 : offset[c - 1];
        int cnt = 0;
        bool over = false;
        const int cap = ball_seg_cap(S);
        for (int sg = 0; sg < S; ++sg) {  // segments in scan order
            const int n_s = wcnt[(size_t)sg * m + q];
            if (n_s > cap) over = true;
            const int take = n_s < cap ? n_s : cap;
            const float *sd = wd + (size_t)sg * cap * m + q;
            const unsigned short *si = wi + (size_t)sg * cap * m + q;
            for (int k = 0; k < take; ++k) {
                if (cnt < kLaneCap) cd[cnt][lane] = sd[(size_t)k * m], ci[cnt][lane] = si[(size_t)k * m];
                ++cnt;
            }
        }
        int *oi = idx + (size_t)q * nsample;
        float *od = dist2 + (size_t)q * nsample;
        if (over || cnt > kLaneCap) {
            oi[0] = -2;  // redone by the wave-per-query kernel
            continue;
        }
        // heap_sort (:33-42) on the un-heapified candidate array, literally -- with the element that travels down the heap held in
        // registers: swap(0, i) + reheap(i) moves a[i] to the root and sifts it down, every child that beats it moving up one level.
        // The reference swaps at every level; holding the traveller and writing it once where it stops leaves the same array, with
        // one LDS round trip per level (the two children) instead of three.
        for (int i = cnt - 1; i > 0; --i) {
            const float dr = cd[i][lane];
            const unsigned short xr = ci[i][lane];
            cd[i][lane] = cd[0][lane], ci[i][lane] = ci[0][lane];
            int root = 0, child = 1;
            while (child < i) {
                // both children in one LDS round trip (child + 1 <= i is always a valid slot; it only counts below i)
                float dc = cd[child][lane];
                const float dc1 = cd[child + 1][lane];
                if (child + 1 < i && dc1 > dc) child++, dc = dc1;
                if (dr > dc) break;
                cd[root][lane] = dc, ci[root][lane] = ci[child][lane];
                root = child;
                child = root * 2 + 1;
            }
            cd[root][lane] = dr, ci[root][lane] = xr;
        }
        if (cnt <= nsample) {
            for (int i = 0; i < nsample; ++i) {
                oi[i] = i < cnt ? my_start + (int)ci[i][lane] : -1;
                od[i] = i < cnt ? cd[i][lane] : 1e10f;
            }
        } else {
            const float sep = (float)cnt / nsample;  // :115
            for (int i = 0; i < nsample; ++i) {
                const int index = (int)(sep * i);  // :118
                oi[i] = my_start + (int)ci[index][lane];
                od[i] = (float)(my_start + (int)ci[index][lane]);  // :120 (sic): the reference stores the index as dist2
            }
        }
    }
}

inline int ball_split_segments(int m)
{
    static const int forced = pcm_mb_switch("PCM_BALL_S", 0);  // A/B switch for tools/mb/mb_ball.py
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    int S = 1;  // >= 4 waves per SIMD (1024 SIMDs x 64 lanes): the scan hides its scalar-load latency (C5: S = 1 / 2 / 4 / 8 -> 611 / 420 / 345 / 327 us)
    while (S < 8 && (long)m * S < 262144) S *= 2;
    return S;
}

__global__ __launch_bounds__(256) void pcm_random_ball_query_kernel(int m, int nsample, float min_radius, float max_radius,
                                                                    const int *__restrict__ order,
                                                                    const float *__restrict__ xyz,
                                                                    const float *__restrict__ new_xyz,
                                                                    const int *__restrict__ offset,
                                                                    const int *__restrict__ new_offset, int b,
                                                                    int *__restrict__ idx, float *__restrict__ dist2)
{
    const int lane = threadIdx.x & 63;
    const int waves_per_block = blockDim.x >> 6;
    const float max_r2 = max_radius * max_radius;
    const float min_r2 = min_radius * min_radius;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    for (int q = blockIdx.x * waves_per_block + (threadIdx.x >> 6); q < m; q += gridDim.x * waves_per_block) {
        const int bt = pcm_cloud_of(q, new_offset, b);
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float qx = new_xyz[(size_t)q * 3 + 0];
        const float qy = new_xyz[(size_t)q * 3 + 1];
        const float qz = new_xyz[(size_t)q * 3 + 2];
        int *oi = idx + (size_t)q * nsample;
        float *od = dist2 + (size_t)q * nsample;
        int cnt = 0;
        for (int base = start; base < end && cnt < nsample; base += 64) {
            const int p = base + lane;
            bool in = false;
            float d2 = 0.f;
            int o = 0;
            if (p < end) {
                o = order[p];
                d2 = pcm_sqdist(qx, qy, qz, xyz[(size_t)o * 3 + 0], xyz[(size_t)o * 3 + 1], xyz[(size_t)o * 3 + 2]);
                in = in_ball(d2, min_r2, max_r2);
            }
            const unsigned long long mask = __ballot(in);
            const int pos = cnt + __builtin_popcountll(mask & lt_mask);
            if (in && pos < nsample) {
                od[pos] = d2;
                oi[pos] = o;
            }
            cnt += __builtin_popcountll(mask);
        }
        if (cnt > nsample) cnt = nsample;
        for (int i = cnt + lane; i < nsample; i += 64) {
            oi[i] = -1;
            od[i] = 1e10f;
        }
    }
}

}  // namespace

// b > 0: number of clouds (the reference ABI does not carry it, so its entry points scan new_offset linearly -- up to
// b dependent loads per query, the dominant cost at b = 128; with b the owning cloud is found by bisection).
extern "C" int pcm_ball_query_b_hip(int b, int m, int nsample, float min_radius, float max_radius, const float *xyz,
                                    const float *new_xyz, const int *offset, const int *new_offset, int *idx,
                                    float *dist2, void *stream)
{
    if (m < 0 || nsample < 1 || b < 0) return PCM_ERR_BAD_ARG;
    if (m == 0) return PCM_OK;
    int blocks = m < 256 * 16 ? m : 256 * 16;
    static const int force_wave = pcm_mb_switch("PCM_BALL_WAVE", 0);  // A/B switch for tools/mb
    if (b > 0 && m >= 32768 && !force_wave) {
        // enough queries to fill the chip with 64-query waves (48 KiB of LDS each: three per CU): lane-per-query kernel, then
        // the flagged leftovers.  Measured, radius 0.1, nsample 16 (wave-per-query -> lanes): 128 x 1024 points / 65 536 queries
        // 0.32 -> 0.25 ms; 32 x 4096 / 65 536: 1.45 -> 1.00 ms; 8 x ~4096 / 16 384: 0.40 -> 0.60 ms (too few waves), hence the bound
        int lblocks = (m + 63) / 64;
        if (lblocks > 256 * 8) lblocks = 256 * 8;
        hipLaunchKernelGGL(pcm_ball_query_lanes_kernel, dim3(lblocks), dim3(64), 0, (hipStream_t)stream, m, nsample, min_radius,
                           max_radius, xyz, new_xyz, offset, new_offset, b, idx, dist2);
        hipLaunchKernelGGL(pcm_ball_query_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, m, nsample, min_radius,
                           max_radius, xyz, new_xyz, offset, new_offset, b, idx, dist2, 1);
        return PCM_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(pcm_ball_query_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, m, nsample, min_radius,
                       max_radius, xyz, new_xyz, offset, new_offset, b, idx, dist2, 0);
    return PCM_LAUNCH_STATUS();
}

// Split path with a caller-provided workspace (the reference ABI has no room for one, so pcm_ball_query(_b)_hip keep the single-kernel
// paths).  pcm_ball_query_ws_bytes(m) = 0 means "too few queries to pay for two launches": call pcm_ball_query_b_hip.
extern "C" size_t pcm_ball_query_ws_bytes(int m)
{
    if (m < 8192) return 0;  // measured: 4096 queries x 1024 points 26 us in one kernel, 54 us split
    const size_t S = (size_t)ball_split_segments(m);
    return S * (size_t)m * 4 + S * ball_seg_cap((int)S) * (size_t)m * 4 + S * ball_seg_cap((int)S) * (size_t)m * 2 + 64;
}

extern "C" int pcm_ball_query_ws_hip(int b, int m, int nsample, float min_radius, float max_radius, const float *xyz,
                                     const float *new_xyz, const int *offset, const int *new_offset, int *idx, float *dist2,
                                     void *ws, size_t ws_bytes, void *stream)
{
    if (m < 0 || nsample < 1 || b <= 0) return PCM_ERR_BAD_ARG;
    if (m == 0) return PCM_OK;
    const size_t need = pcm_ball_query_ws_bytes(m);
    if (need == 0 || ws == nullptr || ws_bytes < need)
        return pcm_ball_query_b_hip(b, m, nsample, min_radius, max_radius, xyz, new_xyz, offset, new_offset, idx, dist2, stream);
    hipStream_t st = (hipStream_t)stream;
    const int S = ball_split_segments(m);
    int *wcnt = static_cast<int *>(ws);
    float *wd = reinterpret_cast<float *>(wcnt + (size_t)S * m);
    unsigned short *wi = reinterpret_cast<unsigned short *>(wd + (size_t)S * ball_seg_cap(S) * m);
    const long waves = (long)((m + 63) / 64) * S;
    hipLaunchKernelGGL(pcm_ball_collect_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, m, S, min_radius, max_radius, xyz,
                       new_xyz, offset, new_offset, b, wcnt, wd, wi);
    int rblocks = (m + 63) / 64;
    if (rblocks > 256 * 8) rblocks = 256 * 8;
    hipLaunchKernelGGL(pcm_ball_replay_kernel, dim3(rblocks), dim3(64), 0, st, m, S, nsample, offset, new_offset, b, wcnt, wd, wi, idx,
                       dist2);
    int blocks = m < 256 * 16 ? m : 256 * 16;
    hipLaunchKernelGGL(pcm_ball_query_kernel, dim3(blocks), dim3(64), 0, st, m, nsample, min_radius, max_radius, xyz, new_xyz, offset,
                       new_offset, b, idx, dist2, 1);  // the flagged leftovers (> kLaneCap candidates, clouds above 65 535 points)
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_ball_query_hip(int m, int nsample, float min_radius, float max_radius, const float *xyz,
                                  const float *new_xyz, const int *offset, const int *new_offset, int *idx,
                                  float *dist2, void *stream)
{
    return pcm_ball_query_b_hip(0, m, nsample, min_radius, max_radius, xyz, new_xyz, offset, new_offset, idx, dist2, stream);
}

extern "C" int pcm_random_ball_query_b_hip(int b, int m, int nsample, float min_radius, float max_radius, const int *order,
                                           const float *xyz, const float *new_xyz, const int *offset,
                                           const int *new_offset, int *idx, float *dist2, void *stream)
{
    if (m < 0 || nsample < 1 || b < 0) return PCM_ERR_BAD_ARG;
    if (m == 0) return PCM_OK;
    int blocks = (m + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(pcm_random_ball_query_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, m, nsample,
                       min_radius, max_radius, order, xyz, new_xyz, offset, new_offset, b, idx, dist2);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_random_ball_query_hip(int m, int nsample, float min_radius, float max_radius, const int *order,
                                         const float *xyz, const float *new_xyz, const int *offset,
                                         const int *new_offset, int *idx, float *dist2, void *stream)
{
    return pcm_random_ball_query_b_hip(0, m, nsample, min_radius, max_radius, order, xyz, new_xyz, offset, new_offset, idx, dist2,
                                       stream);
}
