// sa_fused.hip -- fused set-abstraction layer for gfx950 (MI355X).  HBM / L2 bound.
//
// Replaces the 7 framework ops of the reference's SA layer after FPS+kNN
//   (/root/reference/src/models/components/act/act.py:446-460 and the pure-PyTorch grouping(),
//    /root/reference/libs/pointops/functions/grouping.py:35-59):
//       group [rel xyz | feat] (m,K,3+C)  ->  Linear(3+C -> H, no bias)  ->  BatchNorm1d(H) over m*K rows
//       ->  ReLU  ->  max over K                                                    => tokens (m,H)
//
// Algebra:  Linear([p_j - q_i, f_j]) = Gf[j] + Wp (p_j - q_i)  with  Gf = f Wf^T  (n,H):
// one (n x C)(C x H) GEMM on the n points (hipBLASLt, done by the caller) instead of one on the m*K
// gathered rows (K*m/n = 8x fewer FLOPs at M = N/2), and the xyz term -- 3 multiply-adds -- is
// evaluated in fp32 inside the gather, so relative coordinates never get rounded to bf16.
// BN + ReLU + max commute with a per-channel choice:  max_s relu(a y_s + b) = relu(a * (a>=0 ? max_s y_s
// : min_s y_s) + b),  so ONE gather pass produces the batch statistics AND per-(query,channel)
// max/min/argmax/argmin; the grouped (m,K,3+C) tensor (540 MB at the shipped config) and the
// (m,H,K) BN tensors are never materialised.
//
// Backward needs no m*K*H pass either.  With delta = dz * [z > 0] living at slot s* (argmax/argmin):
//   dbeta = sum_i delta, dgamma = sum_i delta * yhat_sel,
//   dGf[j] = a * ( D[j] - cnt_j dbeta/N - (dgamma/N) r (cnt_j (Gf[j]-mu) + Wp S_j) ),   N = m*K,
//   D[j] = scatter of the m*H deltas (K times fewer atomics than a grouped-tensor backward),
//   cnt_j / S_j = occurrence count / summed relative coordinates of point j (index-only pass),
//   dWp from the same small sums (E, R, T, M below).
//
// Layout: Gf (n,H) fp32 or bf16 row-major; idx (m,K) int32 with -1 placeholders (an all-zero row,
// exactly like the reference's appended zero row + mask); ymax/ymin (m,H) fp32, amax/amin (m,H) u8.
// Thread mapping: a lane owns VEC consecutive channels (16-byte loads), a wave owns a fixed chunk of
// 64*VEC channels and strides over queries, so per-channel sums stay in registers for the whole
// kernel and leave as one partial row per wave (reduced in fp64 by pcm_sa_reduce_kernel).
#include "pcm_common.hpp"

#include <hip/hip_bf16.h>

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxK = 64;

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <>
struct Elem<__hip_bfloat16> {
    static __device__ __forceinline__ float ld(const __hip_bfloat16 *p)
    {
        return __uint_as_float((uint32_t)(*reinterpret_cast<const uint16_t *>(p)) << 16);
    }
    static __device__ __forceinline__ void st(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *src, float (&out)[VEC])
{
    if constexpr (VEC == 4 && sizeof(T) == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(src);
        out[0] = v.x, out[1] = v.y, out[2] = v.z, out[3] = v.w;
    } else if constexpr (VEC == 4 && sizeof(T) == 2) {
        const uint2 v = *reinterpret_cast<const uint2 *>(src);
        out[0] = __uint_as_float(v.x << 16), out[1] = __uint_as_float(v.x & 0xFFFF0000u);
        out[2] = __uint_as_float(v.y << 16), out[3] = __uint_as_float(v.y & 0xFFFF0000u);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) out[v] = Elem<T>::ld(src + v);
    }
}


// Sum the per-wave register partials of a block's kWaves waves (all own the SAME channel chunk) in
// LDS and write ONE partial row per block: partial[slot][t][c0+v].
template <int NV, int VEC>
__device__ __forceinline__ void block_combine_store(float (&vals)[NV][VEC], float *__restrict__ partial, int slot, int H,
                                                    int c0, bool act, float *lds /* kWaves*64*NV*VEC floats */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < NV; ++t)
#pragma unroll
        for (int v = 0; v < VEC; ++v) lds[((wave * NV + t) * VEC + v) * 64 + lane] = vals[t][v];
    __syncthreads();
    if (wave == 0 && act) {
#pragma unroll
        for (int t = 0; t < NV; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) acc += lds[((w * NV + t) * VEC + v) * 64 + lane];
                partial[((size_t)slot * NV + t) * H + c0 + v] = acc;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// forward: gather + statistics + per-(query,channel) max / min / argmax / argmin
// partial layout: [slot][2][H]   (slot = wave index within its chunk class)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void pcm_sa_fwd_kernel(int m, int K, int H, int nchunk, const T *__restrict__ Gf,
                                                            const float *__restrict__ p, const float *__restrict__ q,
                                                            const int *__restrict__ idx, const float *__restrict__ Wp,
                                                            float *__restrict__ ymax, float *__restrict__ ymin,
                                                            uint8_t *__restrict__ amax, uint8_t *__restrict__ amin,
                                                            float *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware block -> work mapping.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (observed: block b
    // runs on XCD b % 8), each with a private 4 MiB L2.  Queries are therefore split into 8 CONTIGUOUS ranges, one per
    // XCD: neighbours live in the query's own cloud, so an XCD's L2 only ever holds the Gf rows of "its" clouds instead
    // of every XCD streaming the whole (n, H) matrix.  gridDim.x is a multiple of 8 * nchunk (launcher guarantees).
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int chunk = local % nchunk;
    const int slot_local = local / nchunk, slots_per_xcd = per_xcd / nchunk;
    const int slot = xcd * slots_per_xcd + slot_local;  // row of `partial` written by this block
    const int q_begin = (int)((long)m * xcd / 8), q_end = (int)((long)m * (xcd + 1) / 8);
    const int c0 = chunk * 64 * VEC + lane * VEC;
    const bool act = c0 < H;  // H % VEC == 0
    __shared__ float lds[kWaves * 64 * 2 * VEC];
    float wx[VEC], wy[VEC], wz[VEC], sum[VEC], sq[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        wx[v] = act ? Wp[(c0 + v) * 3 + 0] : 0.f;
        wy[v] = act ? Wp[(c0 + v) * 3 + 1] : 0.f;
        wz[v] = act ? Wp[(c0 + v) * 3 + 2] : 0.f;
        sum[v] = 0.f, sq[v] = 0.f;
    }
    // BatchNorm statistics are accumulated around a per-channel constant close to the mean (the Gf row of the very first
    // neighbour): sum (y - sh), sum (y - sh)^2 do not cancel when |mean| >> std.  pcm_sa_stats_kernel adds sh back.
    float sh[VEC];
    {
        const int j0 = idx[0];
#pragma unroll
        for (int v = 0; v < VEC; ++v) sh[v] = 0.f;
        if (j0 >= 0 && act) load_vec<T, VEC>(Gf + (size_t)j0 * H + c0, sh);
    }
    for (int i = q_begin + slot_local * kWaves + wave; i < q_end; i += slots_per_xcd * kWaves) {
        // lane s < K fetches neighbour s of query i and its relative coordinates
        int j = -1;
        float rx = 0.f, ry = 0.f, rz = 0.f;
        if (lane < K) {
            j = idx[(size_t)i * K + lane];
            if (j >= 0) {
                rx = p[(size_t)j * 3 + 0] - q[(size_t)i * 3 + 0];
                ry = p[(size_t)j * 3 + 1] - q[(size_t)i * 3 + 1];
                rz = p[(size_t)j * 3 + 2] - q[(size_t)i * 3 + 2];
            }
        }
        float mx[VEC], mn[VEC];
        int ax[VEC], an[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) mx[v] = -INFINITY, mn[v] = INFINITY, ax[v] = 0, an[v] = 0;
        for (int s = 0; s < K; ++s) {
            const int js = __builtin_amdgcn_readlane(j, s);
            float y[VEC];
            if (js >= 0 && act) {
                const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rx), s));
                const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ry), s));
                const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rz), s));
                load_vec<T, VEC>(Gf + (size_t)js * H + c0, y);
#pragma unroll
                for (int v = 0; v < VEC; ++v) y[v] = y[v] + (wx[v] * sx + wy[v] * sy + wz[v] * sz);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) y[v] = 0.f;  // idx == -1: the reference's all-zero row
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float d = y[v] - sh[v];
                sum[v] += d;
                sq[v] += d * d;
                if (y[v] > mx[v]) mx[v] = y[v], ax[v] = s;  // strict: first maximum, like MaxPool1d
                if (y[v] < mn[v]) mn[v] = y[v], an[v] = s;
            }
        }
        if (act) {
            const size_t o = (size_t)i * H + c0;
            if constexpr (VEC == 4) {
                *reinterpret_cast<float4 *>(ymax + o) = make_float4(mx[0], mx[1], mx[2], mx[3]);
                *reinterpret_cast<float4 *>(ymin + o) = make_float4(mn[0], mn[1], mn[2], mn[3]);
                *reinterpret_cast<uint32_t *>(amax + o) = (uint32_t)ax[0] | ((uint32_t)ax[1] << 8) | ((uint32_t)ax[2] << 16) | ((uint32_t)ax[3] << 24);
                *reinterpret_cast<uint32_t *>(amin + o) = (uint32_t)an[0] | ((uint32_t)an[1] << 8) | ((uint32_t)an[2] << 16) | ((uint32_t)an[3] << 24);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    ymax[o + v] = mx[v], ymin[o + v] = mn[v];
                    amax[o + v] = (uint8_t)ax[v], amin[o + v] = (uint8_t)an[v];
                }
            }
        }
    }
    float vals[2][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) vals[0][v] = sum[v], vals[1][v] = sq[v];
    block_combine_store<2, VEC>(vals, partial, slot, H, c0, act, lds);
}

// out[e] = sum over slots of partial[slot][e], e in [0, V*H), accumulated in fp64.
// One block per 64 consecutive elements; its 8 waves stride over the slots with coalesced 256-byte
// reads and meet in LDS.
constexpr int kRedWaves = 8;
__global__ __launch_bounds__(64 * kRedWaves) void pcm_sa_reduce_kernel(int nslots, int VH, const float *__restrict__ partial,
                                                                         float *__restrict__ out)
{
    __shared__ double red[kRedWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    double acc = 0.0;
    if (e < VH) {
        for (int s = wave; s < nslots; s += kRedWaves) acc += (double)partial[(size_t)s * VH + e];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && e < VH) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kRedWaves; ++w) t += red[w][lane];
        out[e] = (float)t;
    }
}

// stats: sums[2][H] -> stat[4][H] = { mean, invstd, a = gamma*invstd, b = beta - a*mean }, running stats update
template <typename T>
__global__ __launch_bounds__(kBlock) void pcm_sa_stats_kernel(int H, double count, float eps, float momentum, const T *__restrict__ Gf,
                                                              const int *__restrict__ idx, const float *__restrict__ sums,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *__restrict__ stat,
                                                              float *__restrict__ running_mean, float *__restrict__ running_var)
{
    const int h = blockIdx.x * kBlock + threadIdx.x;
    if (h >= H) return;
    const int j0 = idx[0];
    const double shift = j0 >= 0 ? (double)Elem<T>::ld(Gf + (size_t)j0 * H + h) : 0.0;  // see pcm_sa_fwd_kernel
    const double dm = (double)sums[h] / count;
    const double mean = shift + dm;
    double var = (double)sums[H + h] / count - dm * dm;  // biased, like BatchNorm's normalisation
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float a = gamma[h] * invstd;
    stat[h] = (float)mean;
    stat[H + h] = invstd;
    stat[2 * H + h] = a;
    stat[3 * H + h] = beta[h] - a * (float)mean;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[h] = (1.f - momentum) * running_mean[h] + momentum * (float)mean;
        running_var[h] = (1.f - momentum) * running_var[h] + momentum * (float)unbiased;
    }
}

// z[i,h] = relu(a*sel + b), sel = a >= 0 ? ymax : ymin
__global__ __launch_bounds__(kBlock) void pcm_sa_apply_kernel(long total, int H, const float *__restrict__ ymax,
                                                              const float *__restrict__ ymin, const float *__restrict__ stat,
                                                              float *__restrict__ z)
{
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const int h = (int)(e % H);
        const float a = stat[2 * H + h], b = stat[3 * H + h];
        const float v = a * (a >= 0.f ? ymax[e] : ymin[e]) + b;
        z[e] = v > 0.f ? v : 0.f;
    }
}

// index-only pass: cnt[j], S[j][3], RM[12] = { R[3], M[9] }   (atomics; m*K threads)
__global__ __launch_bounds__(kBlock) void pcm_sa_index_kernel(long rows, int K, const float *__restrict__ p,
                                                              const float *__restrict__ q, const int *__restrict__ idx,
                                                              float *__restrict__ cnt, float *__restrict__ S, float *__restrict__ RM)
{
    float acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = 0.f;
    for (long r = (long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long)gridDim.x * kBlock) {
        const int j = idx[r];
        if (j < 0) continue;
        const long i = r / K;
        const float rel[3] = {p[(size_t)j * 3 + 0] - q[i * 3 + 0], p[(size_t)j * 3 + 1] - q[i * 3 + 1],
                              p[(size_t)j * 3 + 2] - q[i * 3 + 2]};
        unsafeAtomicAdd(cnt + j, 1.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            unsafeAtomicAdd(S + (size_t)j * 3 + c, rel[c]);
            acc[c] += rel[c];
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[3 + c * 3 + d] += rel[c] * rel[d];
        }
    }
    __shared__ float red[kWaves][12];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        float v = acc[t];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.f;
        for (int w = 0; w < kWaves; ++w) v += red[w][threadIdx.x];
        unsafeAtomicAdd(RM + threadIdx.x, v);  // one atomic per block and sum
    }
}

// backward pass 1 over (m,H): delta, partial[slot][5][H] = { dbeta, dgamma, E0, E1, E2 }, D[j*,h] += delta
template <int VEC>
__global__ __launch_bounds__(kBlock) void pcm_sa_bwd1_kernel(int m, int K, int H, int nchunk, const float *__restrict__ dz,
                                                             const float *__restrict__ z, const float *__restrict__ ymax,
                                                             const float *__restrict__ ymin, const uint8_t *__restrict__ amax,
                                                             const uint8_t *__restrict__ amin, const float *__restrict__ stat,
                                                             const float *__restrict__ p, const float *__restrict__ q,
                                                             const int *__restrict__ idx, float *__restrict__ D,
                                                             float *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x % nchunk;  // gridDim.x is a multiple of nchunk (launcher guarantees)
    const int slot = blockIdx.x / nchunk, nslots = gridDim.x / nchunk;
    const int c0 = chunk * 64 * VEC + lane * VEC;
    const bool act = c0 < H;
    __shared__ float lds[kWaves * 64 * 5 * VEC];
    float mean[VEC], invstd[VEC], a[VEC], acc[VEC][5];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        mean[v] = act ? stat[c0 + v] : 0.f;
        invstd[v] = act ? stat[H + c0 + v] : 0.f;
        a[v] = act ? stat[2 * H + c0 + v] : 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[v][t] = 0.f;
    }
    if (act) {
        for (int i = slot * kWaves + wave; i < m; i += nslots * kWaves) {
            const float qx = q[(size_t)i * 3 + 0], qy = q[(size_t)i * 3 + 1], qz = q[(size_t)i * 3 + 2];
            const size_t o0 = (size_t)i * H + c0;
            float zz[VEC], dd[VEC], ymx[VEC], ymn[VEC];
            int sx[VEC], sn[VEC];
            load_vec<float, VEC>(z + o0, zz);
            load_vec<float, VEC>(dz + o0, dd);
            load_vec<float, VEC>(ymax + o0, ymx);
            load_vec<float, VEC>(ymin + o0, ymn);
            if constexpr (VEC == 4) {
                const uint32_t ua = *reinterpret_cast<const uint32_t *>(amax + o0), ub = *reinterpret_cast<const uint32_t *>(amin + o0);
#pragma unroll
                for (int v = 0; v < 4; ++v) sx[v] = (ua >> (8 * v)) & 0xFF, sn[v] = (ub >> (8 * v)) & 0xFF;
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) sx[v] = amax[o0 + v], sn[v] = amin[o0 + v];
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float delta = zz[v] > 0.f ? dd[v] : 0.f;
                if (delta == 0.f) continue;
                const bool pos = a[v] >= 0.f;
                const float sel = pos ? ymx[v] : ymn[v];
                const int s = pos ? sx[v] : sn[v];
                const int j = idx[(size_t)i * K + s];
                acc[v][0] += delta;
                acc[v][1] += delta * ((sel - mean[v]) * invstd[v]);
                if (j >= 0) {
                    acc[v][2] += delta * (p[(size_t)j * 3 + 0] - qx);
                    acc[v][3] += delta * (p[(size_t)j * 3 + 1] - qy);
                    acc[v][4] += delta * (p[(size_t)j * 3 + 2] - qz);
                    unsafeAtomicAdd(D + (size_t)j * H + c0 + v, delta);
                }
            }
        }
    }
    float vals[5][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int t = 0; t < 5; ++t) vals[t][v] = acc[v][t];
    block_combine_store<5, VEC>(vals, partial, slot, H, c0, act, lds);
}


// backward pass 1, LDS-staged variant: one workgroup per (cloud, chunk of CH channels).  The D rows of
// one cloud restricted to CH channels (N_c x CH floats <= 96 KiB) live in LDS, the m*H deltas are added
// with ds_add_f32 (no global atomics, no memset of D, coalesced write-out), and the per-channel sums
// { dbeta, dgamma, E0..2 } leave as ONE partial row per cloud.  partial layout [cloud][5][H].
template <int CH>
__global__ __launch_bounds__(kBlock) void pcm_sa_bwd1_lds_kernel(int K, int H, const float *__restrict__ dz,
                                                                 const float *__restrict__ z, const float *__restrict__ ymax,
                                                                 const float *__restrict__ ymin, const uint8_t *__restrict__ amax,
                                                                 const uint8_t *__restrict__ amin, const float *__restrict__ stat,
                                                                 const float *__restrict__ p, const float *__restrict__ q,
                                                                 const int *__restrict__ idx, const int *__restrict__ offset,
                                                                 const int *__restrict__ new_offset, float *__restrict__ D,
                                                                 float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [N_c][CH] then [kBlock/LPQ][5][CH] scratch
    constexpr int LPQ = CH / 4;        // lanes per query (4 channels per lane)
    constexpr int QPB = kBlock / LPQ;  // queries per block pass
    const int nchunks = H / CH;
    const int cloud = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    const int start_n = cloud == 0 ? 0 : offset[cloud - 1], end_n = offset[cloud];
    const int start_m = cloud == 0 ? 0 : new_offset[cloud - 1], end_m = new_offset[cloud];
    const int N = end_n - start_n;
    const int c0 = chunk * CH + (threadIdx.x % LPQ) * 4;  // this lane's 4 channels
    const int lc = (threadIdx.x % LPQ) * 4;               // ... inside the chunk
    for (int e = threadIdx.x; e < N * CH; e += kBlock) tile[e] = 0.f;
    float mean[4], invstd[4], a[4], acc[4][5];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        mean[v] = stat[c0 + v], invstd[v] = stat[H + c0 + v], a[v] = stat[2 * H + c0 + v];
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[v][t] = 0.f;
    }
    __syncthreads();
    for (int i = start_m + threadIdx.x / LPQ; i < end_m; i += QPB) {
        const size_t o0 = (size_t)i * H + c0;
        const float4 zz = *reinterpret_cast<const float4 *>(z + o0), dd = *reinterpret_cast<const float4 *>(dz + o0);
        const float4 mx = *reinterpret_cast<const float4 *>(ymax + o0), mn = *reinterpret_cast<const float4 *>(ymin + o0);
        const uint32_t ua = *reinterpret_cast<const uint32_t *>(amax + o0), ub = *reinterpret_cast<const uint32_t *>(amin + o0);
        const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, dv[4] = {dd.x, dd.y, dd.z, dd.w};
        const float xv[4] = {mx.x, mx.y, mx.z, mx.w}, nv[4] = {mn.x, mn.y, mn.z, mn.w};
        const float qx = q[(size_t)i * 3 + 0], qy = q[(size_t)i * 3 + 1], qz = q[(size_t)i * 3 + 2];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float delta = zv[v] > 0.f ? dv[v] : 0.f;
            if (delta == 0.f) continue;
            const bool pos = a[v] >= 0.f;
            const float sel = pos ? xv[v] : nv[v];
            const int s = (int)(((pos ? ua : ub) >> (8 * v)) & 0xFF);
            const int j = idx[(size_t)i * K + s];
            acc[v][0] += delta;
            acc[v][1] += delta * ((sel - mean[v]) * invstd[v]);
            if (j >= 0) {
                acc[v][2] += delta * (p[(size_t)j * 3 + 0] - qx);
                acc[v][3] += delta * (p[(size_t)j * 3 + 1] - qy);
                acc[v][4] += delta * (p[(size_t)j * 3 + 2] - qz);
                atomicAdd(&tile[(j - start_n) * CH + lc + v], delta);  // ds_add_f32
            }
        }
    }
    __syncthreads();
    // coalesced write-out of the cloud's D rows for this chunk (every element written: no memset needed)
    for (int e = threadIdx.x; e < N * (CH / 4); e += kBlock) {
        const int row = e / (CH / 4), c4 = (e % (CH / 4)) * 4;
        *reinterpret_cast<float4 *>(D + (size_t)(start_n + row) * H + chunk * CH + c4) =
            *reinterpret_cast<const float4 *>(tile + row * CH + c4);
    }
    __syncthreads();
    // per-channel sums: QPB threads share each channel group -> LDS tree over the tile memory (N*CH >= ... not
    // guaranteed), so use a dedicated scratch region after the tile
    float *scr = tile + (size_t)N * CH;  // [QPB][LPQ*4*5] floats, sized by the launcher
    const int qslot = threadIdx.x / LPQ;
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int t = 0; t < 5; ++t) scr[(qslot * 5 + t) * CH + lc + v] = acc[v][t];
    __syncthreads();
    for (int e = threadIdx.x; e < 5 * CH; e += kBlock) {
        const int t = e / CH, ch = e % CH;
        float sum = 0.f;
        for (int w = 0; w < QPB; ++w) sum += scr[(w * 5 + t) * CH + ch];
        partial[((size_t)cloud * 5 + t) * H + chunk * CH + ch] = sum;
    }
}

// backward pass 2 over (n,H): dGf, partial[slot][3][H] = T[c][h] = sum_j S_j[c] (Gf[j,h] - mean_h)
// red1[5][H] = reduced { dbeta, dgamma, E0..2 }
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void pcm_sa_bwd2_kernel(int n, int H, int nchunk, double count, const T *__restrict__ Gf,
                                                             const float *__restrict__ D, const float *__restrict__ cnt,
                                                             const float *__restrict__ S, const float *__restrict__ Wp,
                                                             const float *__restrict__ stat, const float *__restrict__ red1,
                                                             T *__restrict__ dGf, float *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x % nchunk;  // gridDim.x is a multiple of nchunk (launcher guarantees)
    const int slot = blockIdx.x / nchunk, nslots = gridDim.x / nchunk;
    const int c0 = chunk * 64 * VEC + lane * VEC;
    const bool act = c0 < H;
    __shared__ float lds[kWaves * 64 * 3 * VEC];
    const float invN = (float)(1.0 / count);
    float mean[VEC], r[VEC], a[VEC], db[VEC], dg[VEC], wx[VEC], wy[VEC], wz[VEC], t0[VEC], t1[VEC], t2[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const int c = act ? c0 + v : 0;
        mean[v] = stat[c], r[v] = stat[H + c], a[v] = stat[2 * H + c];
        db[v] = red1[c] * invN, dg[v] = red1[H + c] * invN;
        wx[v] = Wp[c * 3 + 0], wy[v] = Wp[c * 3 + 1], wz[v] = Wp[c * 3 + 2];
        t0[v] = t1[v] = t2[v] = 0.f;
    }
    for (int j = slot * kWaves + wave; act && j < n; j += nslots * kWaves) {
        const float cj = cnt[j];
        const float sx = S[(size_t)j * 3 + 0], sy = S[(size_t)j * 3 + 1], sz = S[(size_t)j * 3 + 2];
        float g[VEC];
        load_vec<T, VEC>(Gf + (size_t)j * H + c0, g);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float gc = g[v] - mean[v];
            const float sum_yhat = r[v] * (cj * gc + (wx[v] * sx + wy[v] * sy + wz[v] * sz));
            const float out = a[v] * (D[(size_t)j * H + c0 + v] - cj * db[v] - dg[v] * sum_yhat);
            Elem<T>::st(dGf + (size_t)j * H + c0 + v, out);
            t0[v] += sx * gc, t1[v] += sy * gc, t2[v] += sz * gc;
        }
    }
    float vals[3][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) vals[0][v] = t0[v], vals[1][v] = t1[v], vals[2][v] = t2[v];
    block_combine_store<3, VEC>(vals, partial, slot, H, c0, act, lds);
}

// dWp[h][c], dgamma[h], dbeta[h] from the reduced sums
__global__ __launch_bounds__(kBlock) void pcm_sa_bwd3_kernel(int H, double count, const float *__restrict__ stat,
                                                             const float *__restrict__ red1, const float *__restrict__ red2,
                                                             const float *__restrict__ RM, const float *__restrict__ Wp,
                                                             float *__restrict__ dWp, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta)
{
    const int h = blockIdx.x * kBlock + threadIdx.x;
    if (h >= H) return;
    const float invN = (float)(1.0 / count);
    const float r = stat[H + h], a = stat[2 * H + h];
    const float db = red1[h], dg = red1[H + h];
    dbeta[h] = db;
    dgamma[h] = dg;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float E = red1[(2 + c) * H + h];
        const float T = red2[c * H + h];
        // sum over valid rows of (y - mean) * rel_c = T + sum_c' Wp[h][c'] * M[c'][c]
        const float wm = Wp[h * 3 + 0] * RM[3 + 0 * 3 + c] + Wp[h * 3 + 1] * RM[3 + 1 * 3 + c] + Wp[h * 3 + 2] * RM[3 + 2 * 3 + c];
        dWp[h * 3 + c] = a * (E - db * invN * RM[c] - dg * invN * r * (T + wm));
    }
}

inline int waves_grid(long units, int nchunk)
{
    long blocks = (units + kWaves - 1) / kWaves * nchunk;  // one unit per wave, one chunk per block
    const long cap = 256L * 4;                             // 4 workgroups per CU; waves walk several units
    if (blocks > cap) blocks = cap;
    const long unit = 8L * nchunk;  // whole chunk sets on each of the 8 XCDs (pcm_sa_fwd_kernel's mapping)
    blocks = (blocks + unit - 1) / unit * unit;
    return (int)blocks;
}

}  // namespace

// number of partial-row slots the forward/backward kernels will write for `units` work items
extern "C" int pcm_sa_fused_slots(int units, int H, int vec)
{
    const int nchunk = (H + 64 * vec - 1) / (64 * vec);
    return waves_grid(units, nchunk) / nchunk;
}

// channels per workgroup of the LDS-staged bwd1 (0: does not fit / not applicable -> global-atomic kernel)
extern "C" int pcm_sa_fused_bwd1_lds_channels(int H, int n_max)
{
    if (H % 4 != 0 || n_max <= 0) return 0;
    for (int c : {32, 16, 8, 4})
        if (H % c == 0 && (size_t)n_max * c * 4 + (size_t)(kBlock / (c / 4)) * 5 * c * 4 <= 128 * 1024) return c;
    return 0;
}

#define PCM_SA_ST ((hipStream_t)stream)

extern "C" int pcm_sa_fused_forward_hip(int m, int K, int H, int gf_is_bf16, const void *Gf, const float *p, const float *q,
                                        const int *idx, const float *Wp, const float *gamma, const float *beta, float eps,
                                        float momentum, float *running_mean, float *running_var, float *ymax, float *ymin,
                                        unsigned char *amax, unsigned char *amin, float *partial, float *sums, float *stat,
                                        float *z, int stage_mask, void *stream)
{
    // stage_mask: bit0 gather+stats, bit1 reduce, bit2 stats->affine, bit3 apply; <= 0 means all (used by
    // bench.py to time each kernel alone with HIP events)
    if (stage_mask <= 0) stage_mask = 0xF;
    if (m <= 0 || K <= 0 || K > kMaxK || H <= 0) return PCM_ERR_BAD_ARG;
    const int vec = (H % 4 == 0) ? 4 : 1;
    const int nchunk = (H + 64 * vec - 1) / (64 * vec);
    const int grid = waves_grid(m, nchunk);
    const int nslots = grid / nchunk;
#define PCM_FWD(T, V)                                                                                                        \
    do {                                                                                                                     \
        if (stage_mask & 1)                                                                                                  \
            hipLaunchKernelGGL((pcm_sa_fwd_kernel<T, V>), dim3(grid), dim3(kBlock), 0, PCM_SA_ST, m, K, H, nchunk,          \
                               (const T *)Gf, p, q, idx, Wp, ymax, ymin, amax, amin, partial);                              \
    } while (0)
    if (gf_is_bf16) {
        if (vec == 4) PCM_FWD(__hip_bfloat16, 4); else PCM_FWD(__hip_bfloat16, 1);
    } else {
        if (vec == 4) PCM_FWD(float, 4); else PCM_FWD(float, 1);
    }
#undef PCM_FWD
    int rc = PCM_LAUNCH_STATUS();
    if (rc) return rc;
    if (stage_mask & 2)
        hipLaunchKernelGGL(pcm_sa_reduce_kernel, dim3((2 * H + 63) / 64), dim3(64 * kRedWaves), 0, PCM_SA_ST, nslots, 2 * H, partial, sums);
    if (stage_mask & 4) {
        if (gf_is_bf16)
            hipLaunchKernelGGL(pcm_sa_stats_kernel<__hip_bfloat16>, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, PCM_SA_ST, H,
                               (double)m * K, eps, momentum, (const __hip_bfloat16 *)Gf, idx, sums, gamma, beta, stat, running_mean,
                               running_var);
        else
            hipLaunchKernelGGL(pcm_sa_stats_kernel<float>, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, PCM_SA_ST, H, (double)m * K,
                               eps, momentum, (const float *)Gf, idx, sums, gamma, beta, stat, running_mean, running_var);
    }
    const long total = (long)m * H;
    long blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256L * 16) blocks = 256L * 16;
    if (stage_mask & 8) hipLaunchKernelGGL(pcm_sa_apply_kernel, dim3((int)blocks), dim3(kBlock), 0, PCM_SA_ST, total, H, ymax, ymin, stat, z);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_sa_fused_backward_hip(int m, int n, int K, int H, int gf_is_bf16, const void *Gf, const float *p,
                                         const float *q, const int *idx, const float *Wp, const float *stat, const float *dz,
                                         const float *z, const float *ymax, const float *ymin, const unsigned char *amax,
                                         const unsigned char *amin, float *D, float *cnt, float *S, float *RM, float *partial,
                                         float *red1, float *red2, void *dGf, float *dWp, float *dgamma, float *dbeta,
                                         const int *offset, const int *new_offset, int b, int n_max, int stage_mask,
                                         void *stream)
{
    // stage_mask: bit0 index pass, bit1 bwd1, bit2 reduce1, bit3 bwd2, bit4 reduce2, bit5 bwd3; <= 0 means all
    if (stage_mask <= 0) stage_mask = 0x3F;
    if (m <= 0 || n <= 0 || K <= 0 || K > kMaxK || H <= 0) return PCM_ERR_BAD_ARG;
    const int vec = (H % 4 == 0) ? 4 : 1;
    const int nchunk = (H + 64 * vec - 1) / (64 * vec);
    const double count = (double)m * K;
    // cnt, S, RM must be zero on entry; D too unless the LDS-staged bwd1 is taken (it writes every element)
    const long rows = (long)m * K;
    long iblocks = (rows + kBlock - 1) / kBlock;
    if (iblocks > 256) iblocks = 256;
    if (stage_mask & 1) hipLaunchKernelGGL(pcm_sa_index_kernel, dim3((int)iblocks), dim3(kBlock), 0, PCM_SA_ST, rows, K, p, q, idx, cnt, S, RM);
    {
        // LDS-staged variant when the cloud layout is known and a cloud's D rows for >= 4 channels fit in LDS
        const int CH = (b > 0 && offset && new_offset) ? pcm_sa_fused_bwd1_lds_channels(H, n_max) : 0;
        int nslots;
        if (CH) {
            nslots = b;
            const size_t lds = (size_t)n_max * CH * 4 + (size_t)(kBlock / (CH / 4)) * 5 * CH * 4;
            const int grid = b * (H / CH);
#define PCM_B1L(C)                                                                                                            \
    do {                                                                                                                     \
        auto kfn = pcm_sa_bwd1_lds_kernel<C>;                                                                                 \
        if (lds > 64 * 1024) {                                                                                               \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e_ != hipSuccess) return pcm_status(e_);                                                                     \
        }                                                                                                                    \
        if (stage_mask & 2)                                                                                                  \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), lds, PCM_SA_ST, K, H, dz, z, ymax, ymin, amax, amin, stat, p, q, \
                               idx, offset, new_offset, D, partial);                                                          \
    } while (0)
            if (CH == 32) PCM_B1L(32); else if (CH == 16) PCM_B1L(16); else if (CH == 8) PCM_B1L(8); else PCM_B1L(4);
#undef PCM_B1L
        } else {
            const int grid = waves_grid(m, nchunk);
            nslots = grid / nchunk;
            if (!(stage_mask & 2)) {
            } else if (vec == 4)
                hipLaunchKernelGGL((pcm_sa_bwd1_kernel<4>), dim3(grid), dim3(kBlock), 0, PCM_SA_ST, m, K, H, nchunk, dz, z, ymax, ymin,
                                   amax, amin, stat, p, q, idx, D, partial);
            else
                hipLaunchKernelGGL((pcm_sa_bwd1_kernel<1>), dim3(grid), dim3(kBlock), 0, PCM_SA_ST, m, K, H, nchunk, dz, z, ymax, ymin,
                                   amax, amin, stat, p, q, idx, D, partial);
        }
        if (stage_mask & 4) hipLaunchKernelGGL(pcm_sa_reduce_kernel, dim3((5 * H + 63) / 64), dim3(64 * kRedWaves), 0, PCM_SA_ST, nslots, 5 * H, partial, red1);
    }
    {
        const int grid = waves_grid(n, nchunk);
        const int nslots = grid / nchunk;
#define PCM_B2(T, V)                                                                                                         \
    do {                                                                                                                     \
        if (stage_mask & 8)                                                                                                  \
            hipLaunchKernelGGL((pcm_sa_bwd2_kernel<T, V>), dim3(grid), dim3(kBlock), 0, PCM_SA_ST, n, H, nchunk, count,     \
                               (const T *)Gf, D, cnt, S, Wp, stat, red1, (T *)dGf, partial);                                \
    } while (0)
        if (gf_is_bf16) {
            if (vec == 4) PCM_B2(__hip_bfloat16, 4); else PCM_B2(__hip_bfloat16, 1);
        } else {
            if (vec == 4) PCM_B2(float, 4); else PCM_B2(float, 1);
        }
#undef PCM_B2
        if (stage_mask & 16) hipLaunchKernelGGL(pcm_sa_reduce_kernel, dim3((3 * H + 63) / 64), dim3(64 * kRedWaves), 0, PCM_SA_ST, nslots, 3 * H, partial, red2);
    }
    if (stage_mask & 32) hipLaunchKernelGGL(pcm_sa_bwd3_kernel, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, PCM_SA_ST, H, count, stat, red1, red2, RM,
                       Wp, dWp, dgamma, dbeta);
    return PCM_LAUNCH_STATUS();
}
