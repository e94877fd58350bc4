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
// BN + ReLU + max commute with a per-channel choice:  max_s relu(a y_s + b) = relu(a * sel + b) with
// sel = (a >= 0 ? max_s y_s : min_s y_s), and sign(a) = sign(gamma) is known BEFORE the statistics are
// (a = gamma * invstd, invstd > 0).  So ONE gather pass produces the batch statistics and, per (query,
// channel), the one extremum that can matter plus its slot: 5 bytes per (query, channel) leave the kernel
// (sel f32 + asel u8); the grouped (m,K,3+C) tensor (540 MB at the shipped config) and the (m,H,K) BN
// tensors are never materialised.
//
// Backward needs no m*K*H pass either.  With delta = dz * [a sel + b > 0] living at slot asel:
//   dbeta = sum_i delta, dgamma = sum_i delta * yhat_sel,
//   dGf[j] = a * ( D[j] - cnt_j dbeta/N - (dgamma/N) r (cnt_j (Gf[j]-mu) + Wp S_j) ),   N = m*K,
//   D[j] = scatter of the m*H deltas (K times fewer atomics than a grouped-tensor backward),
//   cnt_j / S_j = occurrence count / summed relative coordinates of point j (index-only pass: a function of the
//   coordinates alone, so the caller may run it ahead of time next to the kNN query),
//   dWp from the same small sums (E, R, T, M below).
//
// Layout: Gf (n,H) fp32 or bf16 row-major; idx (m,K) int32 with -1 placeholders (an all-zero row,
// exactly like the reference's appended zero row + mask); sel (m,H) fp32, asel (m,H) u8.
// Thread mapping: a lane owns VEC consecutive channels (16-byte loads: VEC = 8 for bf16, 4 for fp32); LPQ =
// min(64, H / VEC) lanes cover one row, so a wave works on QPW = 64 / LPQ queries at once (H = 96, bf16: 12 lanes per
// row, 5 rows per wave) or, for wide layers, on one chunk of 64*VEC channels.  Per-channel sums stay in registers for
// the whole kernel and leave as one partial row per workgroup (reduced in fp64 by pcm_sa_reduce_kernel) -- in a
// fixed order, so the BatchNorm statistics are deterministic.
#include "pcm_common.hpp"

#include <hip/hip_bf16.h>

#include <algorithm>

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxK = 64;
constexpr int kStageCap = 256;  // staged (query, slot) entries per wave and pass

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
    } else if constexpr (VEC == 8 && sizeof(T) == 2) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src);
        out[0] = __uint_as_float(v.x << 16), out[1] = __uint_as_float(v.x & 0xFFFF0000u);
        out[2] = __uint_as_float(v.y << 16), out[3] = __uint_as_float(v.y & 0xFFFF0000u);
        out[4] = __uint_as_float(v.z << 16), out[5] = __uint_as_float(v.z & 0xFFFF0000u);
        out[6] = __uint_as_float(v.w << 16), out[7] = __uint_as_float(v.w & 0xFFFF0000u);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) out[v] = Elem<T>::ld(src + v);
    }
}

// A row fragment as it sits in memory (VEC elements of T, fetched with ONE load instruction): gathered rows wait in
// registers in this packed form -- 4 VGPRs for 8 bf16 channels -- so that a lane can keep all K rows of a query in flight.
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC >= 4 ? sizeof(T) * VEC : 4) RawRow {
    uint32_t w[(sizeof(T) * VEC + 3) / 4];
};

template <typename T, int VEC>
__device__ __forceinline__ RawRow<T, VEC> load_raw(const T *src)
{
    RawRow<T, VEC> r;
    if constexpr (sizeof(T) * VEC >= 4)
        __builtin_memcpy(&r, __builtin_assume_aligned(src, sizeof(T) * VEC), sizeof(T) * VEC);
    else
        r.w[0] = *reinterpret_cast<const uint16_t *>(src);
    return r;
}

// -> fp32, every word AND-ed with `keep` first (0 or ~0: the all-zero row of idx == -1)
template <typename T, int VEC>
__device__ __forceinline__ void unpack(const RawRow<T, VEC> &r, uint32_t keep, float (&out)[VEC])
{
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) out[v] = __uint_as_float(r.w[v] & keep);
    } else if constexpr (VEC % 2 == 0) {
#pragma unroll
        for (int v = 0; v < VEC; v += 2) {
            const uint32_t x = r.w[v / 2] & keep;
            out[v] = __uint_as_float(x << 16), out[v + 1] = __uint_as_float(x & 0xFFFF0000u);
        }
    } else {
        out[0] = __uint_as_float((r.w[0] & keep) << 16);
    }
}

template <int VEC>
__device__ __forceinline__ void load_f32(const float *src, float (&out)[VEC])
{
    if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int v = 0; v < VEC; v += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(src + v);
            out[v] = t.x, out[v + 1] = t.y, out[v + 2] = t.z, out[v + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) out[v] = src[v];
    }
}

template <int VEC>
__device__ __forceinline__ void store_f32(float *dst, const float (&in)[VEC])
{
    if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int v = 0; v < VEC; v += 4) *reinterpret_cast<float4 *>(dst + v) = make_float4(in[v], in[v + 1], in[v + 2], in[v + 3]);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) dst[v] = in[v];
    }
}

// Streaming (non-temporal) form for outputs that nobody in THIS kernel reads again.  Round 4's counters at the REF shape: pcm_sa_fwd
// fetched 92.9 MB for 36 MB of algorithmic reads while writing 46 MB -- one cloud's Gf rows are 4 MiB (bf16) / 8 MiB (fp32), an XCD's L2 is
// 4 MiB, every row is gathered by ~8 queries in farthest-point (= spatially scattered) order, and the 40 MB of sel / asel written through
// the same L2 in between push those rows out.  The `nt` hint keeps the written lines from displacing them (hypothesis, to be confirmed
// by the PMC pass of the next hardware run; the results are the same bits either way).
typedef float f4nt __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ void store_f32_nt(float *dst, const float (&in)[VEC])
{
    if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int v = 0; v < VEC; v += 4) {
            const f4nt q = {in[v], in[v + 1], in[v + 2], in[v + 3]};
            __builtin_nontemporal_store(q, reinterpret_cast<f4nt *>(dst + v));
        }
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) __builtin_nontemporal_store(in[v], dst + v);
    }
}

template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T *dst, const float (&in)[VEC])
{
    if constexpr (sizeof(T) == 4) {
        store_f32<VEC>(reinterpret_cast<float *>(dst), in);
    } else if constexpr (VEC % 2 == 0) {
        uint32_t w[VEC / 2];
#pragma unroll
        for (int v = 0; v < VEC; v += 2) {
            w[v / 2] = pcm_cvt_pk_bf16(in[v], in[v + 1]);
        }
        if constexpr (VEC == 8)
            *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
        else if constexpr (VEC == 4)
            *reinterpret_cast<uint2 *>(dst) = make_uint2(w[0], w[1]);
        else
            *reinterpret_cast<uint32_t *>(dst) = w[0];
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) Elem<T>::st(dst + v, in[v]);
    }
}

// Make LDS traffic of ONE wave visible to its own lanes in program order (wave-private staging areas: no s_barrier).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A workgroup's lanes hold NV*VEC running sums each; all lanes with the same `gl` (position inside a row) own the same
// channels.  Sum them over the workgroup's kWaves * qpw row slots in a FIXED order and write one partial row:
// partial[slot][t][c].  scratch: kWaves * qpw * NV * VEC * lpq floats (<= 64 * NV * VEC * kWaves).
template <int NV, int VEC>
__device__ __forceinline__ void group_reduce_store(const float (&vals)[NV][VEC], float *__restrict__ partial, int slot, int H,
                                                   int chunk, int lpq, int qpw, int qi, int gl, bool lane_on, float *scratch)
{
    const int wave = threadIdx.x >> 6;
    const int per = NV * VEC * lpq;  // floats per row slot
    __syncthreads();                 // scratch may alias a staging area that other waves are still reading
    if (qi < qpw) {
#pragma unroll
        for (int t = 0; t < NV; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) scratch[(wave * qpw + qi) * per + (t * VEC + v) * lpq + gl] = lane_on ? vals[t][v] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < per; e += kBlock) {
        const int tv = e / lpq, g = e - tv * lpq;
        const int t = tv / VEC, v = tv - t * VEC;
        const int c = (chunk * lpq + g) * VEC + v;
        if (c >= H) continue;
        float acc = 0.f;
        for (int w = 0; w < kWaves * qpw; ++w) acc += scratch[w * per + e];
        partial[((size_t)slot * NV + t) * H + c] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// forward: gather + statistics + per-(query,channel) selected extremum and its slot
// partial layout: [slot][2][H]
// ---------------------------------------------------------------------------------------------
// per-channel constants of a lane's VEC consecutive channels, loaded with 16-byte accesses where VEC allows
template <int VEC>
__device__ __forceinline__ void load_wp(const float *__restrict__ Wp, int c0, float (&wx)[VEC], float (&wy)[VEC], float (&wz)[VEC])
{
    float w[3 * VEC];
    load_f32<3 * VEC>(Wp + (size_t)c0 * 3, w);  // (H, 3) row-major: 3*VEC contiguous floats, 16-byte aligned when VEC % 4 == 0
#pragma unroll
    for (int v = 0; v < VEC; ++v) wx[v] = w[3 * v], wy[v] = w[3 * v + 1], wz[v] = w[3 * v + 2];
}

typedef float f2 __attribute__((ext_vector_type(2)));  // v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 operands

// The gather is VALU-bound (measured: ~125 scalar fp32 instructions per gathered 8-channel row fragment before this
// form), so its inner loop is written on channel PAIRS with contraction allowed: 3 packed FMAs for the xyz term, packed
// statistics, and the -1 ("all-zero row") case handled by AND-ing the still-packed bf16 words with a lane mask.
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void pcm_sa_fwd_kernel(int m, int K, int H, int lpq, int qpw, int nchunk,
                                                            const T *__restrict__ Gf, const float4 *__restrict__ ent_g,
                                                            const float *__restrict__ Wp, const float *__restrict__ gamma,
                                                            float *__restrict__ sel, uint8_t *__restrict__ asel,
                                                            float *__restrict__ partial)
{
#pragma clang fp contract(fast)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = (VEC + 1) / 2;  // channel pairs per lane (VEC == 1: the second half of the pair is a dummy)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware block -> work mapping.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (block b runs on
    // XCD b % 8), each with a private 4 MiB L2.  Queries are therefore split into 8 CONTIGUOUS ranges, one per XCD:
    // neighbours live in the query's own cloud, so an XCD's L2 only ever holds the Gf rows of "its" clouds instead of
    // every XCD streaming the whole (n, H) matrix.  gridDim.x is a multiple of 8 * nchunk (launcher guarantees).
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    // chunk-MAJOR within an XCD (was chunk = local % nchunk: all channel chunks of a row group in flight together).  Workgroups start in
    // index order, so an XCD now works through all its queries for chunk 0, then chunk 1, ...: the rows live in its L2 are one cloud x ONE
    // chunk of channels (REF, fp32 Gf: 4096 rows x 256 channels x 4 B = 4 MiB instead of 8).  A bijection of the same (slot, chunk) pairs:
    // every block computes what it computed before, `partial` rows included.
    const int slots_per_xcd = per_xcd / nchunk;
    const int chunk = local / slots_per_xcd, slot_local = local - chunk * slots_per_xcd;
    const int slot = xcd * slots_per_xcd + slot_local;  // row of `partial` written by this block
    const int q_begin = (int)((long)m * xcd / 8), q_end = (int)((long)m * (xcd + 1) / 8);
    const int qi = lane / lpq, gl = lane - qi * lpq;
    const int c0 = (chunk * lpq + gl) * VEC;
    const bool lane_on = qi < qpw && c0 < H;  // H % VEC == 0
    float4 *ent = reinterpret_cast<float4 *>(smem) + wave * (qpw * K);  // wave-private staging: (j, rel xyz) per (query, slot)
    f2 wx[P], wy[P], wz[P], sgn[P], sum[P], sq[P], sh[P];
#pragma unroll
    for (int k = 0; k < P; ++k) wx[k] = wy[k] = wz[k] = sum[k] = sq[k] = sh[k] = (f2)(0.f), sgn[k] = (f2)(1.f);
    if (lane_on) {
        float w0[VEC], w1[VEC], w2[VEC], gm[VEC], s0v[VEC];
        load_wp<VEC>(Wp, c0, w0, w1, w2);
        load_f32<VEC>(gamma + c0, gm);
#pragma unroll
        for (int v = 0; v < VEC; ++v) s0v[v] = 0.f;
        // BatchNorm statistics are accumulated around a per-channel constant close to the mean (the Gf row of the very
        // first neighbour): sum (y - sh), sum (y - sh)^2 do not cancel when |mean| >> std.  pcm_sa_stats_kernel adds sh back.
        const int j0 = __float_as_int(ent_g[0].x);
        if (j0 >= 0) load_vec<T, VEC>(Gf + (size_t)j0 * H + c0, s0v);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            wx[v / 2][v % 2] = w0[v], wy[v / 2][v % 2] = w1[v], wz[v / 2][v % 2] = w2[v];
            sgn[v / 2][v % 2] = gm[v] < 0.f ? -1.f : 1.f;  // a = gamma * invstd has gamma's sign
            sh[v / 2][v % 2] = s0v[v];
        }
    }
    const int span = q_end - q_begin;
    const int ngroups = (span + qpw - 1) / qpw;
    const int qs = qi < qpw ? qi : 0;
    constexpr int U = 8;  // rows per batch; two batches (= all K = 16 rows of a query) in flight per lane, still packed
#define PCM_NO_ENTRY make_float4(__int_as_float(-1), 0.f, 0.f, 0.f)
    for (int g = slot_local * kWaves + wave; g < ngroups; g += slots_per_xcd * kWaves) {
        const int i0 = q_begin + g * qpw;
        // stage the (neighbour, relative coordinates) entries of the qpw queries: one coalesced 16-byte load per entry
        const int nent = min(qpw, q_end - i0) * K;
        for (int e = lane; e < qpw * K; e += 64) ent[e] = e < nent ? ent_g[(size_t)i0 * K + e] : PCM_NO_ENTRY;
        wave_lds_sync();
        const int i = i0 + qi;
        if (lane_on && i < q_end) {
            f2 best[P];
            int arg[VEC];
#pragma unroll
            for (int k = 0; k < P; ++k) best[k] = (f2)(-INFINITY);
#pragma unroll
            for (int v = 0; v < VEC; ++v) arg[v] = 0;
            RawRow<T, VEC> raw[2][U];
            auto issue = [&](int buf, int s0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int js = (s0 + u < K) ? __float_as_int(ent[qs * K + s0 + u].x) : -1;
                    raw[buf][u] = load_raw<T, VEC>(Gf + (size_t)(js >= 0 ? js : 0) * H + c0);
                }
            };
            auto consume = [&](int buf, int s0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (s0 + u >= K) break;  // wave-uniform
                    const float4 en = ent[qs * K + s0 + u];  // LDS broadcast read (again: cheaper than 4 live VGPRs per row)
                    const uint32_t keep = __float_as_int(en.x) >= 0 ? 0xFFFFFFFFu : 0u;  // idx == -1: the reference's all-zero row
                    f2 r[P];
                    float rf[VEC];
                    unpack<T, VEC>(raw[buf][u], keep, rf);
#pragma unroll
                    for (int k = 0; k < P; ++k) r[k] = (f2)(0.f);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) r[v / 2][v % 2] = rf[v];
                    // rel = 0 for idx == -1, so the xyz term vanishes with the row
                    const f2 ex = (f2)(en.y), ey = (f2)(en.z), ez = (f2)(en.w);
#pragma unroll
                    for (int k = 0; k < P; ++k) {
                        const f2 y = wz[k] * ez + (wy[k] * ey + (wx[k] * ex + r[k]));
                        const f2 d = y - sh[k];
                        sum[k] += d;
                        sq[k] = d * d + sq[k];
                        const f2 t = y * sgn[k];  // max of y (gamma >= 0) or of -y, i.e. min of y (gamma < 0)
                        // strict: first extremum, like MaxPool1d
                        if (t.x > best[k].x) best[k].x = t.x, arg[2 * k] = s0 + u;
                        if constexpr (VEC > 1)
                            if (t.y > best[k].y) best[k].y = t.y, arg[2 * k + 1] = s0 + u;
                    }
                }
            };
            issue(0, 0);
            for (int s0 = 0; s0 < K; s0 += 2 * U) {
                if (s0 + U < K) issue(1, s0 + U);
                consume(0, s0);
                if (s0 + 2 * U < K) issue(0, s0 + 2 * U);
                if (s0 + U < K) consume(1, s0 + U);
            }
            const size_t o = (size_t)i * H + c0;
            float out[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) out[v] = best[v / 2][v % 2] * sgn[v / 2][v % 2];
            store_f32_nt<VEC>(sel + o, out);
            if constexpr (VEC % 4 == 0) {
#pragma unroll
                for (int v = 0; v < VEC; v += 4)
                {
                    const uint32_t packed = (uint32_t)arg[v] | ((uint32_t)arg[v + 1] << 8) | ((uint32_t)arg[v + 2] << 16) | ((uint32_t)arg[v + 3] << 24);
                    __builtin_nontemporal_store(packed, reinterpret_cast<uint32_t *>(asel + o + v));
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const uint8_t a8 = (uint8_t)arg[v];
                    __builtin_nontemporal_store(a8, asel + o + v);
                }
            }
        }
        wave_lds_sync();  // the next pass overwrites the staging area
    }
    float vals[2][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) vals[0][v] = sum[v / 2][v % 2], vals[1][v] = sq[v / 2][v % 2];
    group_reduce_store<2, VEC>(vals, partial, slot, H, chunk, lpq, qpw, qi, gl, lane_on, smem);
}

// out[e] = sum over slots of partial[slot][e], e in [0, V*H), accumulated in fp64, in a fixed order.
// One block per (64 consecutive elements, slot group): its 8 waves stride over the group's slots with coalesced 256-byte
// reads and meet in LDS.  With many slots a single level leaves the chip idle (V*H / 64 = 16 .. 40 workgroups streaming
// megabytes: 57 us at the shipped ACT shape), so the slots are first folded into kRedGroups partial rows by
// V*H / 64 x kRedGroups workgroups and those rows are summed by a second, tiny launch.
constexpr int kRedWaves = 8;
constexpr int kRedGroups = 16;
__global__ __launch_bounds__(64 * kRedWaves) void pcm_sa_reduce_kernel(int nslots, int VH, const float *__restrict__ partial,
                                                                         float *__restrict__ out)
{
    __shared__ double red[kRedWaves][64];
    asm volatile("" ::"s"(gridDim.y), "s"(nslots), "s"(VH), "s"(partial), "s"(out));  // "Kernel heads", pcm_common.hpp
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int g = blockIdx.y, ng = gridDim.y;  // slot group: slots g, g + ng, ...
    double acc = 0.0;
    if (e < VH) acc = pcm_slot_sum(partial, (size_t)VH, e, g + wave * ng, kRedWaves * ng, nslots);
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && e < VH) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kRedWaves; ++w) t += red[w][lane];
        out[(size_t)g * VH + e] = (float)t;
    }
}

// stats: sums[2][H] -> stat[4][H] = { mean, invstd, a = gamma*invstd, b = beta - a*mean }, running stats update
template <typename T>
__global__ __launch_bounds__(kBlock) void pcm_sa_stats_kernel(int H, double count, float eps, float momentum, const T *__restrict__ Gf,
                                                              const float4 *__restrict__ ent_g, const float *__restrict__ sums,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *__restrict__ stat,
                                                              float *__restrict__ running_mean, float *__restrict__ running_var)
{
    const int h = blockIdx.x * kBlock + threadIdx.x;
    if (h >= H) return;
    const int j0 = __float_as_int(ent_g[0].x);
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

// z[i,h] = relu(a*sel + b)      (backward recomputes a*sel + b with the same two roundings to find the ReLU mask)
template <int VEC>
__global__ __launch_bounds__(kBlock) void pcm_sa_apply_kernel(long total, int H, const float *__restrict__ sel,
                                                              const float *__restrict__ stat, float *__restrict__ z)
{
    for (long e = ((long)blockIdx.x * kBlock + threadIdx.x) * VEC; e < total; e += (long)gridDim.x * kBlock * VEC) {
        const int h = (int)(e % H);
        float s[VEC], o[VEC];
        load_f32<VEC>(sel + e, s);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float t = stat[2 * H + h + v] * s[v] + stat[3 * H + h + v];
            o[v] = t > 0.f ? t : 0.f;
        }
        store_f32<VEC>(z + e, o);
    }
}

// index-only passes (functions of the coordinates alone; the caller runs them next to the kNN query).
// entries: ent[i][s] = (j, p_j - q_i) as one 16-byte record per neighbour slot, j = -1 -> (-1, 0, 0, 0).  The gather and
// both backward passes read these instead of chasing idx -> p / q with four dependent 4-byte loads.
__global__ __launch_bounds__(kBlock) void pcm_sa_entries_kernel(long rows, int K, const float *__restrict__ p,
                                                                const float *__restrict__ q, const int *__restrict__ idx,
                                                                float4 *__restrict__ ent)
{
    for (long r = (long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long)gridDim.x * kBlock) {
        const int j = idx[r];
        float4 e = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f);
        if (j >= 0) {
            const long i = r / K;
            e = make_float4(__int_as_float(j), p[(size_t)j * 3 + 0] - q[i * 3 + 0], p[(size_t)j * 3 + 1] - q[i * 3 + 1],
                            p[(size_t)j * 3 + 2] - q[i * 3 + 2]);
        }
        ent[r] = e;
    }
}

__device__ __forceinline__ void moments_accumulate(float (&acc)[12], const float4 &e)
{
    const float rel[3] = {e.y, e.z, e.w};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        acc[c] += rel[c];
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[3 + c * 3 + d] += rel[c] * rel[d];
    }
}

// block-wide sum of the 12 moments -> one atomic per block and moment (scratch: [waves][12] floats)
__device__ __forceinline__ void moments_flush(float (&acc)[12], float *__restrict__ RM, float *red)
{
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        float v = acc[t];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 12 + t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += red[w * 12 + threadIdx.x];
        unsafeAtomicAdd(RM + threadIdx.x, v);
    }
}

// cnt[j], S[j][3], RM[12] = { R[3], M[9] } with one workgroup per cloud: the cloud's (cnt, S) rows live in LDS
// (ds_add_f32) and are written out once -- for batches of many clouds (one workgroup each fills the chip).
__global__ __launch_bounds__(1024) void pcm_sa_index_lds_kernel(int K, const float4 *__restrict__ ent,
                                                                const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                float *__restrict__ cnt, float *__restrict__ S, float *__restrict__ RM)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [N_c][4] = cnt, Sx, Sy, Sz; then [waves][12]
    const int cloud = blockIdx.x;
    const int start_n = cloud == 0 ? 0 : offset[cloud - 1], end_n = offset[cloud];
    const int start_m = cloud == 0 ? 0 : new_offset[cloud - 1], end_m = new_offset[cloud];
    const int N = end_n - start_n;
    for (int e = threadIdx.x; e < N * 4; e += blockDim.x) tile[e] = 0.f;
    __syncthreads();
    float acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = 0.f;
    const long r0 = (long)start_m * K, r1 = (long)end_m * K;
    for (long r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const float4 e = ent[r];
        const int j = __float_as_int(e.x);
        if (j < 0) continue;
        float *row = tile + (size_t)(j - start_n) * 4;
        atomicAdd(row, 1.f);
        atomicAdd(row + 1, e.y);
        atomicAdd(row + 2, e.z);
        atomicAdd(row + 3, e.w);
        moments_accumulate(acc, e);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N; e += blockDim.x) {
        const float4 v = *reinterpret_cast<const float4 *>(tile + (size_t)e * 4);
        cnt[start_n + e] = v.x;
        S[(size_t)(start_n + e) * 3 + 0] = v.y;
        S[(size_t)(start_n + e) * 3 + 1] = v.z;
        S[(size_t)(start_n + e) * 3 + 2] = v.w;
    }
    moments_flush(acc, RM, tile + (size_t)N * 4);
}

// ... and the layout-free variant (few clouds, or clouds too large for LDS): global atomics, one thread per slot
__global__ __launch_bounds__(kBlock) void pcm_sa_index_kernel(long rows, const float4 *__restrict__ ent, float *__restrict__ cnt,
                                                              float *__restrict__ S, float *__restrict__ RM)
{
    __shared__ float red[kWaves * 12];
    float acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = 0.f;
    for (long r = (long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long)gridDim.x * kBlock) {
        const float4 e = ent[r];
        const int j = __float_as_int(e.x);
        if (j < 0) continue;
        unsafeAtomicAdd(cnt + j, 1.f);
        unsafeAtomicAdd(S + (size_t)j * 3 + 0, e.y);
        unsafeAtomicAdd(S + (size_t)j * 3 + 1, e.z);
        unsafeAtomicAdd(S + (size_t)j * 3 + 2, e.w);
        moments_accumulate(acc, e);
    }
    moments_flush(acc, RM, red);
}

// backward pass 1 over (m,H), layout-free variant: delta, partial[slot][5][H] = { dbeta, dgamma, E0, E1, E2 },
// D[j*,h] += delta with global atomics (D zeroed by the caller).  Used when a cloud's D rows do not fit in LDS.
template <int VEC>
__global__ __launch_bounds__(kBlock) void pcm_sa_bwd1_kernel(int m, int K, int H, int lpq, int qpw, int nchunk,
                                                             const float *__restrict__ dz, const float *__restrict__ sel,
                                                             const uint8_t *__restrict__ asel, const float *__restrict__ stat,
                                                             const float4 *__restrict__ ent, float *__restrict__ D,
                                                             float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x % nchunk;  // gridDim.x is a multiple of nchunk (launcher guarantees)
    const int slot = blockIdx.x / nchunk, nslots = gridDim.x / nchunk;
    const int qi = lane / lpq, gl = lane - qi * lpq;
    const int c0 = (chunk * lpq + gl) * VEC;
    const bool lane_on = qi < qpw && c0 < H;
    float mean[VEC], invstd[VEC], a[VEC], bb[VEC], acc[5][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        mean[v] = invstd[v] = a[v] = bb[v] = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t][v] = 0.f;
    }
    if (lane_on) {
        load_f32<VEC>(stat + c0, mean);
        load_f32<VEC>(stat + H + c0, invstd);
        load_f32<VEC>(stat + 2 * H + c0, a);
        load_f32<VEC>(stat + 3 * H + c0, bb);
        const int ngroups = (m + qpw - 1) / qpw;
        for (int g = slot * kWaves + wave; g < ngroups; g += nslots * kWaves) {
            const int i = g * qpw + qi;
            if (i >= m) continue;
            const size_t o0 = (size_t)i * H + c0;
            float ss[VEC], dd[VEC];
            load_f32<VEC>(sel + o0, ss);
            load_f32<VEC>(dz + o0, dd);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float delta = (a[v] * ss[v] + bb[v]) > 0.f ? dd[v] : 0.f;
                if (delta == 0.f) continue;
                const float4 e = ent[(size_t)i * K + asel[o0 + v]];
                const int j = __float_as_int(e.x);
                acc[0][v] += delta;
                acc[1][v] += delta * ((ss[v] - mean[v]) * invstd[v]);
                if (j >= 0) {
                    acc[2][v] += delta * e.y;
                    acc[3][v] += delta * e.z;
                    acc[4][v] += delta * e.w;
                    unsafeAtomicAdd(D + (size_t)j * H + c0 + v, delta);
                }
            }
        }
    }
    group_reduce_store<5, VEC>(acc, partial, slot, H, chunk, lpq, qpw, qi, gl, lane_on, smem);
}

// backward pass 1, LDS-staged variant: one workgroup per (cloud, chunk of CH channels).  The D rows of one cloud
// restricted to CH channels (N_c x CH floats, up to ~150 KiB of the CU's 160 KiB) live in LDS, the m*H deltas are added
// with ds_add_f32 (no global atomics, no memset of D, coalesced write-out), and the per-channel sums { dbeta, dgamma,
// E0..2 } leave as ONE partial row per cloud.  partial layout [cloud][5][H].
// Workgroup -> (cloud, chunk): logical id L = (b % 8) * (grid / 8) + b / 8, so CONSECUTIVE logical ids run on the same
// XCD: the CH-wide column slices of sibling chunks share 128-byte lines of dz / sel, and now also an L2.
template <int CH>
__global__ __launch_bounds__(1024) void pcm_sa_bwd1_lds_kernel(int K, int H, int nwork, const float *__restrict__ dz,
                                                                 const float *__restrict__ sel, const uint8_t *__restrict__ asel,
                                                                 const float *__restrict__ stat, const float4 *__restrict__ ent,
                                                                 const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                 float *__restrict__ D, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [N_c][CH] then [waves][5][CH] scratch
    constexpr int LPQ = CH / 4;  // lanes per query (4 channels per lane)
    const int L = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (L >= nwork) return;  // grid padded to a multiple of 8
    const int nchunks = H / CH;
    const int cloud = L / nchunks, chunk = L % nchunks;
    const int start_n = cloud == 0 ? 0 : offset[cloud - 1], end_n = offset[cloud];
    const int start_m = cloud == 0 ? 0 : new_offset[cloud - 1], end_m = new_offset[cloud];
    const int N = end_n - start_n;
    const int QPB = blockDim.x / LPQ;  // queries per block pass
    const int lc = (threadIdx.x % LPQ) * 4;  // this lane's 4 channels inside the chunk
    const int c0 = chunk * CH + lc;
    for (int e = threadIdx.x; e < N * (CH / 4); e += blockDim.x) reinterpret_cast<float4 *>(tile)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    float mean[4], invstd[4], a[4], bb[4], acc[5][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        mean[v] = stat[c0 + v], invstd[v] = stat[H + c0 + v], a[v] = stat[2 * H + c0 + v], bb[v] = stat[3 * H + c0 + v];
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t][v] = 0.f;
    }
    __syncthreads();
    constexpr int UQ = 2;  // queries per thread and pass: their loads are issued together
    for (int ib = start_m + threadIdx.x / LPQ; ib < end_m; ib += QPB * UQ) {
        float4 ss[UQ], dd[UQ];
        uint32_t ua[UQ];
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int i = ib + u * QPB;
            const size_t o0 = (size_t)(i < end_m ? i : ib) * H + c0;
            ss[u] = *reinterpret_cast<const float4 *>(sel + o0), dd[u] = *reinterpret_cast<const float4 *>(dz + o0);
            ua[u] = *reinterpret_cast<const uint32_t *>(asel + o0);
        }
        float4 e[UQ][4];
        float delta[UQ][4];
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int i = ib + u * QPB;
            const float sv[4] = {ss[u].x, ss[u].y, ss[u].z, ss[u].w}, dv[4] = {dd[u].x, dd[u].y, dd[u].z, dd[u].w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                delta[u][v] = (i < end_m && (a[v] * sv[v] + bb[v]) > 0.f) ? dv[v] : 0.f;
                // the query's K records share 256 bytes; lanes without a gradient re-read slot 0 (same line)
                e[u][v] = ent[(size_t)(i < end_m ? i : ib) * K + (delta[u][v] != 0.f ? (int)((ua[u] >> (8 * v)) & 0xFF) : 0)];
            }
        }
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const float sv[4] = {ss[u].x, ss[u].y, ss[u].z, ss[u].w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float d = delta[u][v];
                if (d == 0.f) continue;
                const int j = __float_as_int(e[u][v].x);
                acc[0][v] += d;
                acc[1][v] += d * ((sv[v] - mean[v]) * invstd[v]);
                if (j >= 0) {
                    acc[2][v] += d * e[u][v].y;
                    acc[3][v] += d * e[u][v].z;
                    acc[4][v] += d * e[u][v].w;
                    atomicAdd(&tile[(size_t)(j - start_n) * CH + lc + v], d);  // ds_add_f32
                }
            }
        }
    }
    __syncthreads();
    // coalesced write-out of the cloud's D rows for this chunk (every element written: no memset needed)
    for (int e = threadIdx.x; e < N * (CH / 4); e += blockDim.x) {
        const int row = e / (CH / 4), c4 = (e % (CH / 4)) * 4;
        *reinterpret_cast<float4 *>(D + (size_t)(start_n + row) * H + chunk * CH + c4) = reinterpret_cast<const float4 *>(tile)[e];
    }
    // per-channel sums: lanes LPQ apart inside a wave share the channels -> fixed-order butterfly, then one row per wave
    float *scr = tile + (size_t)N * CH;  // [waves][5][CH]
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float x = acc[t][v];
            for (int off = 32; off >= LPQ; off >>= 1) x += __shfl_xor(x, off);
            acc[t][v] = x;
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane < LPQ) {
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) scr[(wave * 5 + t) * CH + lc + v] = acc[t][v];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 5 * CH; e += blockDim.x) {
        const int t = e / CH, ch = e % CH;
        float sum = 0.f;
        for (int w = 0; w < nw; ++w) sum += scr[(w * 5 + t) * CH + ch];
        partial[((size_t)cloud * 5 + t) * H + chunk * CH + ch] = sum;
    }
}

// backward pass 2 over (n,H): dGf, partial[slot][3][H] = T[c][h] = sum_j S_j[c] (Gf[j,h] - mean_h)
// red1[5][H] = reduced { dbeta, dgamma, E0..2 }
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void pcm_sa_bwd2_kernel(int n, int H, int lpq, int qpw, int nchunk, double count,
                                                             const T *__restrict__ Gf, const float *__restrict__ D,
                                                             const float *__restrict__ cnt, const float *__restrict__ S,
                                                             const float *__restrict__ Wp, const float *__restrict__ stat,
                                                             const float *__restrict__ red1, T *__restrict__ dGf,
                                                             float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x % nchunk;  // gridDim.x is a multiple of nchunk (launcher guarantees)
    const int slot = blockIdx.x / nchunk, nslots = gridDim.x / nchunk;
    const int qi = lane / lpq, gl = lane - qi * lpq;
    const int c0 = (chunk * lpq + gl) * VEC;
    const bool lane_on = qi < qpw && c0 < H;
    const float invN = (float)(1.0 / count);
    float mean[VEC], r[VEC], a[VEC], db[VEC], dg[VEC], wx[VEC], wy[VEC], wz[VEC], tt[3][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) mean[v] = r[v] = a[v] = db[v] = dg[v] = wx[v] = wy[v] = wz[v] = 0.f, tt[0][v] = tt[1][v] = tt[2][v] = 0.f;
    if (lane_on) {
        load_f32<VEC>(stat + c0, mean);
        load_f32<VEC>(stat + H + c0, r);
        load_f32<VEC>(stat + 2 * H + c0, a);
        load_f32<VEC>(red1 + c0, db);
        load_f32<VEC>(red1 + H + c0, dg);
        load_wp<VEC>(Wp, c0, wx, wy, wz);
#pragma unroll
        for (int v = 0; v < VEC; ++v) db[v] *= invN, dg[v] *= invN;
        const int ngroups = (n + qpw - 1) / qpw;
        for (int g = slot * kWaves + wave; g < ngroups; g += nslots * kWaves) {
            const int j = g * qpw + qi;
            if (j >= n) continue;
            const float cj = cnt[j];
            const float sx = S[(size_t)j * 3 + 0], sy = S[(size_t)j * 3 + 1], sz = S[(size_t)j * 3 + 2];
            float gv[VEC], dv[VEC], out[VEC];
            load_vec<T, VEC>(Gf + (size_t)j * H + c0, gv);
            load_f32<VEC>(D + (size_t)j * H + c0, dv);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float gc = gv[v] - mean[v];
                const float sum_yhat = r[v] * (cj * gc + (wx[v] * sx + wy[v] * sy + wz[v] * sz));
                out[v] = a[v] * (dv[v] - cj * db[v] - dg[v] * sum_yhat);
                tt[0][v] += sx * gc, tt[1][v] += sy * gc, tt[2][v] += sz * gc;
            }
            store_vec<T, VEC>(dGf + (size_t)j * H + c0, out);
        }
    }
    group_reduce_store<3, VEC>(tt, partial, slot, H, chunk, lpq, qpw, qi, gl, lane_on, smem);
}

// dWp[h][c], dgamma[h], dbeta[h] from the reduced sums
__global__ __launch_bounds__(kBlock) void pcm_sa_bwd3_kernel(int H, double count, const float *__restrict__ stat,
                                                             const float *__restrict__ red1, const float *__restrict__ redn,
                                                             const float *__restrict__ red2,
                                                             const float *__restrict__ RM, const float *__restrict__ Wp,
                                                             float *__restrict__ dWp, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta)
{
    const int h = blockIdx.x * kBlock + threadIdx.x;
    if (h >= H) return;
    const float invN = (float)(1.0 / count);
    const float r = stat[H + h], a = stat[2 * H + h];
    dbeta[h] = red1[h];      // parameter gradients: this rank's sums (the gradient exchange averages them)
    dgamma[h] = red1[H + h];
    const float db = redn[h], dg = redn[H + h];  // normalisation terms: sums over the whole (global) batch
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float E = red1[(2 + c) * H + h];
        const float T = red2[c * H + h];
        // sum over valid rows of (y - mean) * rel_c = T + sum_c' Wp[h][c'] * M[c'][c]
        const float wm = Wp[h * 3 + 0] * RM[3 + 0 * 3 + c] + Wp[h * 3 + 1] * RM[3 + 1 * 3 + c] + Wp[h * 3 + 2] * RM[3 + 2 * 3 + c];
        dWp[h * 3 + c] = a * (E - db * invN * RM[c] - dg * invN * r * (T + wm));
    }
}

// lane -> (row slot, channel group) mapping shared by the row-streaming kernels
struct RowMap {
    int vec, lpq, qpw, nchunk;
};

inline RowMap row_map(int H, int bf16, int K)
{
    RowMap mp;
    mp.vec = bf16 ? (H % 8 == 0 ? 8 : (H % 4 == 0 ? 4 : 1)) : (H % 4 == 0 ? 4 : 1);
    const int lanes = (H + mp.vec - 1) / mp.vec;  // lanes that cover one row
    mp.lpq = lanes < 64 ? lanes : 64;
    mp.qpw = 64 / mp.lpq;
    const int k = K > 0 ? K : 1;
    const int cap = kStageCap / k > 0 ? kStageCap / k : 1;
    if (mp.qpw > cap) mp.qpw = cap;
    mp.nchunk = (lanes + mp.lpq - 1) / mp.lpq;
    return mp;
}

inline int rows_grid(long rows, const RowMap &mp)
{
    const long groups = (rows + mp.qpw - 1) / mp.qpw;
    // every wave loads ~40 per-channel constants before its first row: give it >= 4 row groups when there is enough
    // work, but keep >= 2 workgroups per CU busy and never more than 8
    long per_chunk = (groups + kWaves * 4 - 1) / (kWaves * 4);
    const long lo = 512, hi = 256L * 8;
    if (per_chunk < lo) per_chunk = std::min(lo, (groups + kWaves - 1) / kWaves);
    if (per_chunk > hi) per_chunk = hi;
    long blocks = per_chunk * mp.nchunk;
    const long unit = 8L * mp.nchunk;  // whole chunk sets on each of the 8 XCDs (pcm_sa_fwd_kernel's mapping)
    blocks = (blocks + unit - 1) / unit * unit;
    return (int)blocks;
}

inline size_t rows_smem(const RowMap &mp, int K, int nv)
{
    const size_t stage = (size_t)kWaves * mp.qpw * K * sizeof(float4);
    const size_t red = (size_t)kWaves * mp.qpw * nv * mp.vec * mp.lpq * sizeof(float);
    return stage > red ? stage : red;
}

constexpr size_t kLdsBudget = 156 * 1024;  // of the CU's 160 KiB

template <typename KFN>
int allow_lds(KFN kfn, size_t bytes)
{
    if (bytes <= 64 * 1024) return PCM_OK;
    return pcm_status(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

}  // namespace

// number of partial-row slots the forward/backward kernels will write for `units` rows of width H
extern "C" int pcm_sa_fused_slots(int units, int H, int bf16, int K)
{
    const RowMap mp = row_map(H, bf16, K);
    return rows_grid(units, mp) / mp.nchunk;
}

// channels per workgroup of the LDS-staged bwd1 (0: does not fit / not applicable -> global-atomic kernel)
extern "C" int pcm_sa_fused_bwd1_lds_channels(int H, int n_max)
{
    if (H % 4 != 0 || n_max <= 0) return 0;
    // prefer a tile of <= 64 KiB (two or more workgroups per CU) as long as the column slices stay >= 64 bytes wide
    for (int c : {32, 16})
        if (H % c == 0 && (size_t)n_max * c * 4 + 16 * 5 * c * 4 <= 64 * 1024) return c;
    for (int c : {32, 16, 8, 4})
        if (H % c == 0 && (size_t)n_max * c * 4 + 16 * 5 * c * 4 <= kLdsBudget) return c;
    return 0;
}

#define PCM_SA_ST ((hipStream_t)stream)

// partial (nslots, VH) -> out (VH).  Above 256 slots: two levels through `scratch` (kRedGroups * VH floats; callers pass the
// tail of the partial-row buffer, which they allocate pcm_sa_fused_reduce_scratch_rows() rows longer than the slots need).
extern "C" int pcm_sa_fused_reduce_scratch_rows(void) { return kRedGroups; }

extern "C" int pcm_sa_reduce_rows_hip(int nslots, int VH, const float *partial, float *scratch, float *out, void *stream)
{
    if (nslots <= 0 || VH <= 0) return PCM_ERR_BAD_ARG;
    const int bx = (VH + 63) / 64;
    if (nslots <= 256 || scratch == nullptr) {
        hipLaunchKernelGGL(pcm_sa_reduce_kernel, dim3(bx, 1), dim3(64 * kRedWaves), 0, PCM_SA_ST, nslots, VH, partial, out);
    } else {
        hipLaunchKernelGGL(pcm_sa_reduce_kernel, dim3(bx, kRedGroups), dim3(64 * kRedWaves), 0, PCM_SA_ST, nslots, VH, partial, scratch);
        hipLaunchKernelGGL(pcm_sa_reduce_kernel, dim3(bx, 1), dim3(64 * kRedWaves), 0, PCM_SA_ST, kRedGroups, VH, scratch, out);
    }
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_sa_fused_forward_hip(int m, int K, int H, int gf_is_bf16, const void *Gf, const void *ent,
                                        const float *Wp, const float *gamma, const float *beta, float eps,
                                        float momentum, float *running_mean, float *running_var, float *sel,
                                        unsigned char *asel, float *partial, float *sums, float *stat, float *z,
                                        int stage_mask, void *stream)
{
    // stage_mask: bit0 gather+stats, bit1 reduce, bit2 stats->affine, bit3 apply; <= 0 means all (used by
    // bench.py to time each kernel alone with HIP events)
    if (stage_mask <= 0) stage_mask = 0xF;
    if (m <= 0 || K <= 0 || K > kMaxK || H <= 0) return PCM_ERR_BAD_ARG;
    const RowMap mp = row_map(H, gf_is_bf16, K);
    const int grid = rows_grid(m, mp);
    const int nslots = grid / mp.nchunk;
    const size_t smem = rows_smem(mp, K, 2);
#define PCM_FWD(T, V)                                                                                                        \
    do {                                                                                                                     \
        if (stage_mask & 1)                                                                                                  \
            hipLaunchKernelGGL((pcm_sa_fwd_kernel<T, V>), dim3(grid), dim3(kBlock), smem, PCM_SA_ST, m, K, H, mp.lpq, mp.qpw, \
                               mp.nchunk, (const T *)Gf, (const float4 *)ent, Wp, gamma, sel, asel, partial);                \
    } while (0)
    if (gf_is_bf16) {
        if (mp.vec == 8) PCM_FWD(__hip_bfloat16, 8); else if (mp.vec == 4) PCM_FWD(__hip_bfloat16, 4); else PCM_FWD(__hip_bfloat16, 1);
    } else {
        if (mp.vec == 4) PCM_FWD(float, 4); else PCM_FWD(float, 1);
    }
#undef PCM_FWD
    int rc = PCM_LAUNCH_STATUS();
    if (rc) return rc;
    if (stage_mask & 2) pcm_sa_reduce_rows_hip(nslots, 2 * H, partial, partial + (size_t)nslots * 2 * H, sums, stream);
    if (stage_mask & 4) {
        if (gf_is_bf16)
            hipLaunchKernelGGL(pcm_sa_stats_kernel<__hip_bfloat16>, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, PCM_SA_ST, H,
                               (double)m * K, eps, momentum, (const __hip_bfloat16 *)Gf, (const float4 *)ent, sums, gamma, beta, stat, running_mean,
                               running_var);
        else
            hipLaunchKernelGGL(pcm_sa_stats_kernel<float>, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, PCM_SA_ST, H, (double)m * K,
                               eps, momentum, (const float *)Gf, (const float4 *)ent, sums, gamma, beta, stat, running_mean, running_var);
    }
    if (stage_mask & 8) {
        const long total = (long)m * H;
        const int av = H % 4 == 0 ? 4 : 1;
        long blocks = (total / av + kBlock - 1) / kBlock;
        if (blocks > 256L * 16) blocks = 256L * 16;
        if (av == 4)
            hipLaunchKernelGGL(pcm_sa_apply_kernel<4>, dim3((int)blocks), dim3(kBlock), 0, PCM_SA_ST, total, H, sel, stat, z);
        else
            hipLaunchKernelGGL(pcm_sa_apply_kernel<1>, dim3((int)blocks), dim3(kBlock), 0, PCM_SA_ST, total, H, sel, stat, z);
    }
    return PCM_LAUNCH_STATUS();
}

// index-only products of the neighbour lists: ent (m,K) 16-byte records (j, p_j - q_i), cnt (n), S (n,3), RM (12).
// cnt / S / RM must be zero on entry.  With the cloud layout (offset, new_offset, b, n_max) and many clouds the
// statistics use one LDS tile per cloud, otherwise global atomics.
// the 16-byte neighbour records alone (the reproducible index pass of csrc/sa_scatter.hip builds the rest from a sorted CSR)
extern "C" int pcm_sa_index_entries_hip(int m, int K, const float *p, const float *q, const int *idx, void *ent, void *stream)
{
    if (m <= 0 || K <= 0 || K > kMaxK) return PCM_ERR_BAD_ARG;
    const long rows = (long)m * K;
    long blocks = (rows + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pcm_sa_entries_kernel, dim3((int)blocks), dim3(kBlock), 0, PCM_SA_ST, rows, K, p, q, idx, (float4 *)ent);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_sa_index_hip(int m, int K, const float *p, const float *q, const int *idx, const int *offset,
                                const int *new_offset, int b, int n_max, void *ent, float *cnt, float *S, float *RM,
                                void *stream)
{
    if (m <= 0 || K <= 0 || K > kMaxK) return PCM_ERR_BAD_ARG;
    const long rows = (long)m * K;
    long blocks = (rows + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pcm_sa_entries_kernel, dim3((int)blocks), dim3(kBlock), 0, PCM_SA_ST, rows, K, p, q, idx, (float4 *)ent);
    const size_t lds = (size_t)n_max * 16 + 16 * 12 * 4;
    if (offset && new_offset && b >= 64 && n_max > 0 && lds <= kLdsBudget) {
        const int rc = allow_lds(pcm_sa_index_lds_kernel, lds);
        if (rc) return rc;
        const int threads = n_max >= 2048 ? 1024 : 256;
        hipLaunchKernelGGL(pcm_sa_index_lds_kernel, dim3(b), dim3(threads), lds, PCM_SA_ST, K, (const float4 *)ent, offset, new_offset,
                           cnt, S, RM);
        return PCM_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(pcm_sa_index_kernel, dim3((int)blocks), dim3(kBlock), 0, PCM_SA_ST, rows, (const float4 *)ent, cnt, S, RM);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_sa_fused_backward_hip(int m, int n, int K, int H, int gf_is_bf16, const void *Gf, const void *ent,
                                         const float *Wp, const float *stat, const float *dz,
                                         const float *sel, const unsigned char *asel, float *D, const float *cnt,
                                         const float *S, const float *RM, float *partial, float *red1, float *red2, void *dGf,
                                         float *dWp, float *dgamma, float *dbeta, const int *offset, const int *new_offset,
                                         int b, int n_max, const float *red_global, double count_override, int stage_mask,
                                         void *stream)
{
    // stage_mask: bit1 bwd1, bit2 reduce1, bit3 bwd2, bit4 reduce2, bit5 bwd3; <= 0 means all.  cnt / S / RM come from
    // pcm_sa_index_hip (same neighbour lists).
    if (stage_mask <= 0) stage_mask = 0x3E;
    if (m <= 0 || n <= 0 || K <= 0 || K > kMaxK || H <= 0) return PCM_ERR_BAD_ARG;
    // synchronised BatchNorm: the normalisation terms of the input gradient use the sums and the row count of ALL ranks
    // (red_global = all-reduced {sum delta, sum delta * yhat}); dgamma / dbeta stay the local sums of red1
    const double count = count_override > 0.0 ? count_override : (double)m * K;
    const float *redn = red_global ? red_global : red1;
    {
        // LDS-staged variant when the cloud layout is known and a cloud's D rows for >= 4 channels fit in LDS
        const int CH = (b > 0 && offset && new_offset) ? pcm_sa_fused_bwd1_lds_channels(H, n_max) : 0;
        int nslots;
        if (CH) {
            nslots = b;
            const size_t lds = (size_t)n_max * CH * 4 + (size_t)16 * 5 * CH * 4;
            const int nwork = b * (H / CH);
            const int grid = (nwork + 7) / 8 * 8;
            const int threads = lds > 64 * 1024 ? 1024 : (lds > 32 * 1024 ? 512 : 256);  // one big tile per CU: more waves to hide latency
#define PCM_B1L(C)                                                                                                            \
    do {                                                                                                                     \
        auto kfn = pcm_sa_bwd1_lds_kernel<C>;                                                                                 \
        const int rc_ = allow_lds(kfn, lds);                                                                                  \
        if (rc_) return rc_;                                                                                                  \
        if (stage_mask & 2)                                                                                                  \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(threads), lds, PCM_SA_ST, K, H, nwork, dz, sel, asel, stat,            \
                               (const float4 *)ent, offset, new_offset, D, partial);                                          \
    } while (0)
            if (CH == 32) PCM_B1L(32); else if (CH == 16) PCM_B1L(16); else if (CH == 8) PCM_B1L(8); else PCM_B1L(4);
#undef PCM_B1L
        } else {
            const RowMap mp = row_map(H, 0, 1);  // fp32 rows: VEC 4 / 1
            const int grid = rows_grid(m, mp);
            nslots = grid / mp.nchunk;
            const size_t smem = rows_smem(mp, 1, 5);
            if (!(stage_mask & 2)) {
            } else if (mp.vec == 4)
                hipLaunchKernelGGL((pcm_sa_bwd1_kernel<4>), dim3(grid), dim3(kBlock), smem, PCM_SA_ST, m, K, H, mp.lpq, mp.qpw, mp.nchunk,
                                   dz, sel, asel, stat, (const float4 *)ent, D, partial);
            else
                hipLaunchKernelGGL((pcm_sa_bwd1_kernel<1>), dim3(grid), dim3(kBlock), smem, PCM_SA_ST, m, K, H, mp.lpq, mp.qpw, mp.nchunk,
                                   dz, sel, asel, stat, (const float4 *)ent, D, partial);
        }
        if (stage_mask & 4) pcm_sa_reduce_rows_hip(nslots, 5 * H, partial, partial + (size_t)nslots * 5 * H, red1, stream);
    }
    {
        const RowMap mp = row_map(H, gf_is_bf16, 1);
        const int grid = rows_grid(n, mp);
        const int nslots = grid / mp.nchunk;
        const size_t smem = rows_smem(mp, 1, 3);
#define PCM_B2(T, V)                                                                                                         \
    do {                                                                                                                     \
        if (stage_mask & 8)                                                                                                  \
            hipLaunchKernelGGL((pcm_sa_bwd2_kernel<T, V>), dim3(grid), dim3(kBlock), smem, PCM_SA_ST, n, H, mp.lpq, mp.qpw, \
                               mp.nchunk, count, (const T *)Gf, D, cnt, S, Wp, stat, redn, (T *)dGf, partial);              \
    } while (0)
        if (gf_is_bf16) {
            if (mp.vec == 8) PCM_B2(__hip_bfloat16, 8); else if (mp.vec == 4) PCM_B2(__hip_bfloat16, 4); else PCM_B2(__hip_bfloat16, 1);
        } else {
            if (mp.vec == 4) PCM_B2(float, 4); else PCM_B2(float, 1);
        }
#undef PCM_B2
        if (stage_mask & 16) pcm_sa_reduce_rows_hip(nslots, 3 * H, partial, partial + (size_t)nslots * 3 * H, red2, stream);
    }
    if (stage_mask & 32) hipLaunchKernelGGL(pcm_sa_bwd3_kernel, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, PCM_SA_ST, H, count, stat, red1, redn,
                       red2, RM, Wp, dWp, dgamma, dbeta);
    return PCM_LAUNCH_STATUS();
}
