// sa_scatter.hip -- reproducible (atomic-free) index statistics and delta scatter of the fused set-abstraction layer.
//
// csrc/sa_fused.hip accumulates cnt / S / RM and the m*H backward deltas with float atomics (LDS or global): fast, but
// the order of the additions -- and with it the last bits of every gradient -- changes from run to run, exactly like the
// reference's index_put_(accumulate=True) / atomicAdd backward does
//   (/root/reference/libs/pointops/functions/grouping.py:35-59 autograd, libs/pointops/src/grouping/grouping_cuda_kernel.cu:18-25).
// SURVEY.md section 5 asks for a deterministic mode built on a sorted segmented reduction; this file is it, and it is the
// default path of policy/sa_fused.py (the atomic kernels stay selectable, PCM_SA_SCATTER=atomic).
//
// The neighbour lists idx (m,K) are inverted ONCE per batch, next to the kNN query on the sampling stream, into a CSR
// whose segments are sorted ascending (pcm_scatter_plan_sorted_hip): point j -> the rows r = i*K + s that name it.
//
//   index statistics   cnt[j] = segment length, S[j] = sum of the relative coordinates of its rows in list order,
//                      RM = 12 global moments through per-block partial rows reduced in a fixed order.
//   delta scatter      D[j,c] = sum over rows r = (i,s) of point j of  delta[i,c] * [asel[i,c] == s].
//     A straight gather would touch every dz / sel / asel row K times.  Instead
//       pack    (one wave per query) computes delta = dz * [a*sel+b > 0], drops the zeros, and buckets the query's
//               channels by their arg-extremum slot: run (i,s) = { (c, delta[i,c]) : asel[i,c] == s } stored contiguously
//               (cperm u16 / dperm f32 at i*H + goff[i][s] ...), plus the five per-channel sums the backward needs;
//       gather  (a group of lanes per point j) walks the point's rows in list order and adds each run into an LDS copy
//               of the D row; within a run all channels differ, runs are taken strictly one after the other, so every
//               D[j,c] is summed in ascending (i,s) order whatever the hardware does.  D is written once, coalesced.
//     Traffic: 9 B read + <= 6 B written per (query, channel) in pack, <= 6 B read per (query, channel) + 4 B written per
//     (point, channel) in gather -- no m*K*H term and no read-modify-write of D in HBM.
#include "pcm_common.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// index statistics from the sorted CSR: one thread per point
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pcm_sa_index_csr_kernel(int n, const int *__restrict__ start, const int *__restrict__ list,
                                                                  const float4 *__restrict__ ent, float *__restrict__ cnt,
                                                                  float *__restrict__ S, float *__restrict__ rm_partial)
{
    __shared__ float red[kWaves][12];
    float acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = 0.f;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j < n) {
        const int t0 = start[j], t1 = start[j + 1];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int t = t0; t < t1; ++t) {
            const float4 e = ent[list[t]];
            sx += e.y, sy += e.z, sz += e.w;
            const float rel[3] = {e.y, e.z, e.w};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] += rel[c];
#pragma unroll
                for (int d = 0; d < 3; ++d) acc[3 + c * 3 + d] += rel[c] * rel[d];
            }
        }
        cnt[j] = (float)(t1 - t0);
        S[(size_t)j * 3 + 0] = sx, S[(size_t)j * 3 + 1] = sy, S[(size_t)j * 3 + 2] = sz;
    }
    // fixed-order block sum: butterfly inside the wave, then the waves in order
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        float v = acc[t];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += red[w][threadIdx.x];
        rm_partial[(size_t)blockIdx.x * 12 + threadIdx.x] = v;
    }
}

// RM[t] = sum over blocks of rm_partial[block][t], t < 12: one wave per output (grid 12), lanes stride over the blocks, fp64,
// fixed-order tree -- the single-thread-per-output loop this replaces took 86 us at n = 131072 (512 dependent loads)
__global__ __launch_bounds__(64) void pcm_sa_rm_reduce_kernel(int nblocks, const float *__restrict__ rm_partial, float *__restrict__ RM)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    double v = 0.0;
    for (int b = lane; b < nblocks; b += 64) v += (double)rm_partial[(size_t)b * 12 + t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) RM[t] = (float)v;
}

// ---------------------------------------------------------------------------------------------
// pack: one wave per query.  partial[slot][5][H] = { dbeta, dgamma, E0, E1, E2 } sums of this workgroup's queries
// goff (m, K+1) u16: run (i,s) = positions [goff[i][s], goff[i][s+1]) of row i of cperm / dperm
// ---------------------------------------------------------------------------------------------
template <int TMAX>
__global__ __launch_bounds__(kBlock, 3) void pcm_sa_bwd1_pack_kernel(int m, int K, int H, const float *__restrict__ dz,
                                                                  const float *__restrict__ sel, const uint8_t *__restrict__ asel,
                                                                  const float *__restrict__ stat, const float4 *__restrict__ ent,
                                                                  uint16_t *__restrict__ goff, uint16_t *__restrict__ cperm,
                                                                  float *__restrict__ dperm, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per wave: count / offset table (64 ints), staged run data (H floats + H u16)
    const size_t per_wave = 64 * sizeof(int) + (size_t)H * 4 + (((size_t)H * 2 + 15) & ~(size_t)15);
    unsigned char *base = smem_raw + wave * per_wave;
    int *tab = reinterpret_cast<int *>(base);
    float *outd = reinterpret_cast<float *>(base + 64 * sizeof(int));
    uint16_t *outc = reinterpret_cast<uint16_t *>(base + 64 * sizeof(int) + (size_t)H * 4);
    float mean[TMAX], invstd[TMAX], a[TMAX], bb[TMAX], acc[5][TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int c = lane + 64 * t;
        const bool on = c < H;
        mean[t] = on ? stat[c] : 0.f, invstd[t] = on ? stat[H + c] : 0.f, a[t] = on ? stat[2 * H + c] : 0.f, bb[t] = on ? stat[3 * H + c] : 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k][t] = 0.f;
    }
    const int KS = K + 1;
    for (int i = blockIdx.x * kWaves + wave; i < m; i += gridDim.x * kWaves) {
        tab[lane] = 0;
        wave_lds_sync();
        // three phases so that the memory round trips overlap: every row load first, then every record gather, then the
        // sums and the run positions (a per-channel chain of load -> gather -> atomic cost ~12 us per query)
        float delta[TMAX], ssv[TMAX];
        int slot[TMAX], pos[TMAX];
        const size_t row = (size_t)i * H;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const int c = lane + 64 * t;
            const bool on = c < H;
            ssv[t] = on ? sel[row + c] : 0.f;
            const float dd = on ? dz[row + c] : 0.f;
            slot[t] = on ? (int)asel[row + c] : 0;
            delta[t] = (on && (a[t] * ssv[t] + bb[t]) > 0.f) ? dd : 0.f;
        }
        float4 rec[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) rec[t] = ent[(size_t)i * K + slot[t]];  // the K records of a query are 256 contiguous bytes: cache hits
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            pos[t] = -1;
            const float d = delta[t];
            if (d != 0.f) {
                acc[0][t] += d;
                acc[1][t] += d * ((ssv[t] - mean[t]) * invstd[t]);
                if (__float_as_int(rec[t].x) >= 0) {
                    acc[2][t] += d * rec[t].y, acc[3][t] += d * rec[t].z, acc[4][t] += d * rec[t].w;
                    pos[t] = atomicAdd(&tab[slot[t]], 1);  // integer LDS atomic: position inside the run (any order: channels of a run are distinct)
                }
            }
        }
        wave_lds_sync();
        // exclusive scan of the K run lengths (K <= 64: one lane per run)
        const int len = lane < K ? tab[lane] : 0;
        int inc = len;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        const int excl = inc - len;
        const int total = __shfl(inc, 63);
        wave_lds_sync();
        tab[lane] = excl;
        if (lane <= K) goff[(size_t)i * KS + lane] = (uint16_t)(lane < K ? excl : total);
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (pos[t] >= 0) {
                const int p = tab[slot[t]] + pos[t];
                outd[p] = delta[t];
                outc[p] = (uint16_t)(lane + 64 * t);
            }
        }
        wave_lds_sync();
        for (int p = lane; p < total; p += 64) {
            dperm[row + p] = outd[p];
            cperm[row + p] = outc[p];
        }
        wave_lds_sync();
    }
    // per-channel sums of the workgroup: the waves in order
    __syncthreads();
    float *scr = reinterpret_cast<float *>(smem_raw);  // [kWaves][5][H] <= kWaves * per_wave (host guarantees)
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int c = lane + 64 * t;
        if (c < H) {
#pragma unroll
            for (int k = 0; k < 5; ++k) scr[((size_t)wave * 5 + k) * H + c] = acc[k][t];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 5 * H; e += kBlock) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += scr[(size_t)w * 5 * H + e];
        partial[(size_t)blockIdx.x * 5 * H + e] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// gather: G lanes per point.  row tile in LDS, runs added strictly in list order.
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kBlock) void pcm_sa_bwd1_gather_kernel(int n, int K, int H, const int *__restrict__ start,
                                                                    const int *__restrict__ list, const uint16_t *__restrict__ goff,
                                                                    const uint16_t *__restrict__ cperm, const float *__restrict__ dperm,
                                                                    float *__restrict__ D)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [kBlock / G][H]
    constexpr int GPB = kBlock / G;
    constexpr int UN = 4;  // runs whose first chunk is in flight together
    const int grp = threadIdx.x / G, sub = threadIdx.x % G;
    float *row = tile + (size_t)grp * H;
    const int KS = K + 1;
    // XCD x scatters into the x-th eighth of the points = whole clouds: all runs of a query's (channel, delta) records are then read
    // through ONE L2 (round 3: consecutive workgroups took consecutive points, a record line was fetched by ~4 XCDs: PMC 3.57x at H = 96)
    const PcmXcdSplit sp = pcm_xcd_split(n, GPB);
    const long span = sp.hi > sp.lo ? sp.hi - sp.lo : 0;  // uniform inside a workgroup: every group of a wave makes the same trips
    const int npass = (int)((span + sp.step - 1) / sp.step);
    for (int pass = 0; pass < npass; ++pass) {
        const long jl = sp.first + (long)pass * sp.step + grp;
        const int j = (int)jl;
        const bool on = jl < sp.hi;
        for (int c = sub; c < H; c += G) row[c] = 0.f;
        wave_lds_sync();
        const int t0 = on ? start[j] : 0, t1 = on ? start[j + 1] : 0;
        for (int tb = t0; tb < t1; tb += G) {
            // one row id per lane: where its run starts and how long it is
            int my_base = 0, my_len = 0;
            if (tb + sub < t1) {
                const int r = list[tb + sub];
                const int i = r / K, s = r - i * K;
                const int g0 = goff[(size_t)i * KS + s], g1 = goff[(size_t)i * KS + s + 1];
                my_base = i * H + g0, my_len = g1 - g0;
            }
            const int nb = min(G, t1 - tb);
            for (int u0 = 0; u0 < nb; u0 += UN) {
                int rb[UN], rl[UN], c0[UN];
                float d0[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    rb[u] = __shfl(my_base, u0 + u, G), rl[u] = u0 + u < nb ? __shfl(my_len, u0 + u, G) : 0;
                    c0[u] = 0, d0[u] = 0.f;
                    if (sub < rl[u]) c0[u] = cperm[(size_t)rb[u] + sub], d0[u] = dperm[(size_t)rb[u] + sub];
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (sub < rl[u]) atomicAdd(&row[c0[u]], d0[u]);  // ds_add_f32: LDS operations of one wave retire in issue order
                    for (int p = sub + G; p < rl[u]; p += G) atomicAdd(&row[cperm[(size_t)rb[u] + p]], dperm[(size_t)rb[u] + p]);
                }
            }
        }
        wave_lds_sync();
        if (on) {
            if ((H & 3) == 0) {
                for (int c = sub * 4; c < H; c += G * 4)
                    *reinterpret_cast<float4 *>(D + (size_t)j * H + c) = *reinterpret_cast<const float4 *>(row + c);
            } else {
                for (int c = sub; c < H; c += G) D[(size_t)j * H + c] = row[c];
            }
        }
        wave_lds_sync();
    }
}

inline size_t pack_wave_bytes(int H) { return 64 * sizeof(int) + (size_t)H * 4 + (((size_t)H * 2 + 15) & ~(size_t)15); }

inline int pack_grid(int m)
{
    long g = ((long)m + kWaves - 1) / kWaves;
    if (g > 1024) g = 1024;  // <= 1024 partial rows; every wave takes >= 1 query
    return (int)(g < 1 ? 1 : g);
}

inline int gather_lanes(int K, int H)
{
    const int avg = (H + K - 1) / K;  // mean run length before the ReLU zeros are dropped
    return avg > 32 ? 64 : (avg > 16 ? 32 : 16);
}

}  // namespace

extern "C" int pcm_sa_det_supported(int K, int H) { return K >= 1 && K <= 63 && H >= 1 && H <= 1024; }

// ---- index pass ---------------------------------------------------------------------------------------
// csr: (n + 1) + m*K ints = start | list of the sorted plan; scratch ints: pcm_sa_index_det_scratch_ints(n)
// (plan scratch followed by the RM partial rows).  Writes ent, csr, cnt, S, RM -- nothing needs zeroing.
extern "C" long pcm_sa_index_det_scratch_ints(int n)
{
    const long blocks = ((long)n + kBlock - 1) / kBlock;
    return pcm_scatter_plan_sorted_scratch_ints(n) + blocks * 12;
}

extern "C" int pcm_sa_index_entries_hip(int m, int K, const float *p, const float *q, const int *idx, void *ent, void *stream);
extern "C" int pcm_sa_reduce_rows_hip(int nslots, int VH, const float *partial, float *scratch, float *out, void *stream);

extern "C" int pcm_sa_index_det_hip(int m, int K, int n, const float *p, const float *q, const int *idx, void *ent, int *csr,
                                    int *scratch, float *cnt, float *S, float *RM, void *stream)
{
    if (m <= 0 || n <= 0 || K <= 0 || K > 63) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc = pcm_sa_index_entries_hip(m, K, p, q, idx, ent, stream);
    if (rc) return rc;
    const long rows = (long)m * K;
    int *start = csr, *list = csr + n + 1;
    rc = pcm_scatter_plan_sorted_hip(rows, n, idx, scratch, start, list, stream);
    if (rc) return rc;
    float *rm_partial = reinterpret_cast<float *>(scratch + pcm_scatter_plan_sorted_scratch_ints(n));
    const int blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(pcm_sa_index_csr_kernel, dim3(blocks), dim3(kBlock), 0, st, n, start, list, (const float4 *)ent, cnt, S, rm_partial);
    hipLaunchKernelGGL(pcm_sa_rm_reduce_kernel, dim3(12), dim3(64), 0, st, blocks, rm_partial, RM);
    return PCM_LAUNCH_STATUS();
}

// ---- backward pass 1 ----------------------------------------------------------------------------------
extern "C" int pcm_sa_bwd1_det_slots(int m) { return pack_grid(m); }

extern "C" long pcm_sa_bwd1_det_ws_bytes(int m, int K, int H)
{
    // goff (m, K+1) u16 | cperm (m, H) u16 | dperm (m, H) f32, each 16-byte aligned
    const long a = (((long)m * (K + 1) * 2 + 15) / 16) * 16, b = (((long)m * H * 2 + 15) / 16) * 16;
    return a + b + (long)m * H * 4;
}

// D (n,H) written entirely; partial: pcm_sa_bwd1_det_slots(m) * 5 * H floats; red1 (5,H) = the reduced sums.
// stage_mask: 1 pack, 2 gather, 4 reduce (<= 0: all) -- bench.py times the kernels one at a time.
extern "C" int pcm_sa_bwd1_det_hip(int m, int n, int K, int H, const float *dz, const float *sel, const unsigned char *asel,
                                   const float *stat, const void *ent, const int *csr, void *ws, float *D, float *partial,
                                   float *red1, int stage_mask, void *stream)
{
    if (m <= 0 || n <= 0 || !pcm_sa_det_supported(K, H)) return PCM_ERR_BAD_ARG;
    if ((long)m * H >= 2147483647L) return PCM_ERR_UNSUPPORTED;  // run positions are 32-bit
    if (stage_mask <= 0) stage_mask = 7;
    hipStream_t st = (hipStream_t)stream;
    unsigned char *w = static_cast<unsigned char *>(ws);
    const long a = (((long)m * (K + 1) * 2 + 15) / 16) * 16, b = (((long)m * H * 2 + 15) / 16) * 16;
    uint16_t *goff = reinterpret_cast<uint16_t *>(w), *cperm = reinterpret_cast<uint16_t *>(w + a);
    float *dperm = reinterpret_cast<float *>(w + a + b);
    const int grid = pack_grid(m);
    if (stage_mask & 1) {
        size_t lds = kWaves * pack_wave_bytes(H);
        const size_t red = (size_t)kWaves * 5 * H * sizeof(float);
        if (red > lds) lds = red;
        const int T = (H + 63) / 64;
#define PCM_PACK(TM)                                                                                                                   \
    do {                                                                                                                               \
        auto kfn = pcm_sa_bwd1_pack_kernel<TM>;                                                                                        \
        if (lds > 64 * 1024) {                                                                                                         \
            const int rc_ = pcm_status(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            if (rc_) return rc_;                                                                                                       \
        }                                                                                                                              \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), lds, st, m, K, H, dz, sel, asel, stat, (const float4 *)ent, goff, cperm,    \
                           dperm, partial);                                                                                            \
    } while (0)
        if (T <= 2) PCM_PACK(2); else if (T <= 8) PCM_PACK(8); else PCM_PACK(16);
#undef PCM_PACK
    }
    if (stage_mask & 2) {
        const int *start = csr, *list = csr + n + 1;
        const int G = gather_lanes(K, H);
        const int gpb = kBlock / G;
        long blocks = ((long)n + gpb - 1) / gpb;
        if (blocks > 256L * 32) blocks = 256L * 32;
        blocks = pcm_xcd_grid(blocks);
        const size_t lds = (size_t)gpb * H * sizeof(float);
#define PCM_GATHER(GG) hipLaunchKernelGGL(pcm_sa_bwd1_gather_kernel<GG>, dim3((int)blocks), dim3(kBlock), lds, st, n, K, H, start, list, goff, cperm, dperm, D)
        if (G == 64) PCM_GATHER(64); else if (G == 32) PCM_GATHER(32); else PCM_GATHER(16);
#undef PCM_GATHER
    }
    if (stage_mask & 4) return pcm_sa_reduce_rows_hip(grid, 5 * H, partial, partial + (size_t)grid * 5 * H, red1, stream);
    return PCM_LAUNCH_STATUS();
}
