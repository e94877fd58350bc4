// tokens.hip -- element-wise glue around the attention in-projection, fused, for gfx950.  HBM bound.
//
// Every self-attention of the ACT transformer computes  q = k-input = src + pos,  v-input = src
// (/root/reference/src/models/components/act/transformer.py:244-249, 318-323) and, under bf16 autocast, casts both
// to bf16 before the projection GEMMs; backward casts the two input gradients back to fp32 and adds them, and
// reduces three (rows, E) gradients over rows for the projection biases.  Through the framework: add + 2 casts
// forward, 2 casts + 2 adds + 2 strided reductions + 2 cats backward -- per layer, 15 layers per step.  Here:
//
//   pcm_add_cast2 : sum16 = bf16(x + pos), x16 = bf16(x)            one launch (pos broadcast over the batch if shorter)
//   pcm_add2_cast : out   = f32(a) + f32(b)                          one launch
//   pcm_colsum    : out[t][c] = sum_rows g_t[row][c]  for up to 3 (rows, C) tensors, fp32 partials per row slot, fixed
//                   order (deterministic), final reduce in fp64, written in the bias dtype      two launches
//
// Bytes per element: add_cast2 8 read + 4 written; add2_cast 4 read + 4 written; colsum 2 read.
#include "pcm_elem.hpp"

namespace {

constexpr int kBlock = 256;

inline int ew_grid(long work)
{
    long blocks = (work + kBlock - 1) / kBlock;
    if (blocks > 256L * 16) blocks = 256L * 16;
    return (int)(blocks < 1 ? 1 : blocks);
}

__global__ __launch_bounds__(kBlock) void pcm_add_cast2_kernel(long n4, long pos4, const float *__restrict__ x,
                                                               const float *__restrict__ pos, __hip_bfloat16 *__restrict__ sum16,
                                                               __hip_bfloat16 *__restrict__ x16)
{
    asm volatile("" ::"s"(gridDim.x), "s"(n4), "s"(pos4), "s"(x), "s"(pos), "s"(sum16), "s"(x16));  // "Kernel heads", pcm_common.hpp
    const bool small = n4 <= 0xFFFFFFFFl && pos4 <= 0xFFFFFFFFl;  // 32-bit remainder (always, at this model's sizes)
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        float a[4], p[4], s[4];
        load4<float>(x + i * 4, a);
        load4<float>(pos + (small ? (long)((unsigned)i % (unsigned)pos4) : i % pos4) * 4, p);
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = a[u] + p[u];
        store4<__hip_bfloat16>(sum16 + i * 4, s);
        if (x16 != nullptr) store4<__hip_bfloat16>(x16 + i * 4, a);
    }
}

__global__ __launch_bounds__(kBlock) void pcm_add2_cast_kernel(long n4, const __hip_bfloat16 *__restrict__ a,
                                                               const __hip_bfloat16 *__restrict__ b, float *__restrict__ out,
                                                               float *__restrict__ a32, const float *__restrict__ c32,
                                                               const __hip_bfloat16 *__restrict__ a2)
{
    asm volatile("" ::"s"(gridDim.x), "s"(n4), "s"(a), "s"(b), "s"(out), "s"(a32), "s"(c32), "s"(a2));  // "Kernel heads", pcm_common.hpp
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        // all (up to four) loads of the piece are requested before the first sum (the optional addends' `if`s made each wait for its
        // own load in turn); the sums themselves are the same, in the same order
        float x[4], y[4], o[4], z[4] = {0.f, 0.f, 0.f, 0.f};
        uint2 r2 = make_uint2(0u, 0u);  // raw bits: unpacking inside the `if` would wait for the load there
        load4<__hip_bfloat16>(a + i * 4, x);
        load4<__hip_bfloat16>(b + i * 4, y);
        if (a2 != nullptr) r2 = *reinterpret_cast<const uint2 *>(a2 + i * 4);  // the first addend arrives in two parts (dq W_q + dk W_k of a batched product)
        if (c32 != nullptr) load4<float>(c32 + i * 4, z);                     // a third, fp32 addend: the residual branch's gradient of the same tensor
        if (a2 != nullptr) {
            const float x2[4] = {__uint_as_float(r2.x << 16), __uint_as_float(r2.x & 0xFFFF0000u), __uint_as_float(r2.y << 16),
                                 __uint_as_float(r2.y & 0xFFFF0000u)};
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] += x2[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = x[u] + y[u];
        if (c32 != nullptr) {
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] += z[u];
        }
        store4<float>(out + i * 4, o);
        if (a32 != nullptr) store4<float>(a32 + i * 4, x);  // the first addend alone, widened (the position gradient)
    }
}

struct ColsumArgs {
    const void *g[3];
    long ld[3];  // row stride in elements
};

// a thread owns 4 channels of one tensor; rows_per_pass = 256 / (C/4) rows in flight
template <typename T>
__device__ __forceinline__ void colsum_body(long rows, int C, long rows_per_slot, const T *__restrict__ g, long ld, int slot, int t,
                                            int ntensors, float *__restrict__ partial, float *lds)
{
    const int lpr = C / 4, rpp = kBlock / lpr;
    const int col4 = threadIdx.x % lpr, rsub = threadIdx.x / lpr;
    const bool act = rsub < rpp;
    const long r0 = (long)slot * rows_per_slot;
    const long r1 = r0 + rows_per_slot < rows ? r0 + rows_per_slot : rows;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (act) {
        // four rows in flight, added in row order (the one-row loop waited for every load in turn: >= 8 round trips in series per thread)
        long r = r0 + rsub;
        for (; r + 3 * rpp < r1; r += 4 * (long)rpp) {
            float v[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) load4<T>(g + (r + q * (long)rpp) * ld + col4 * 4, v[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += v[q][u];
        }
        for (; r < r1; r += rpp) {
            float v[4];
            load4<T>(g + r * ld + col4 * 4, v);
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += v[u];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) lds[u * kBlock + threadIdx.x] = s[u];
    __syncthreads();
    if (act && rsub == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc = 0.f;
            for (int rr = 0; rr < rpp; ++rr) acc += lds[u * kBlock + rr * lpr + col4];
            partial[((size_t)slot * ntensors + t) * C + col4 * 4 + u] = acc;
        }
    }
}

// grid (slots, ntensors)
template <typename T>
__global__ __launch_bounds__(kBlock) void pcm_colsum_kernel(long rows, int C, long rows_per_slot, ColsumArgs args,
                                                            float *__restrict__ partial)
{
    __shared__ float lds[4 * kBlock];
    const int t = blockIdx.y;
    colsum_body<T>(rows, C, rows_per_slot, (const T *)args.g[t], args.ld[t], blockIdx.x, t, gridDim.y, partial, lds);
}

// the first stages of several column sums in one launch (policy/deferred.py): job table by value, one workgroup = one
// (job, slot, tensor); same arithmetic as pcm_colsum_kernel
constexpr int kColsumBatch = 16;
struct ColsumJob {
    ColsumArgs a;
    long rows, rps;
    float *partial;
    int C, ntensors, is_bf16, blk0;
};
struct ColsumBatch {
    ColsumJob j[kColsumBatch];
    int n;
};
__global__ __launch_bounds__(kBlock) void pcm_colsum_batch_kernel(ColsumBatch b)
{
    __shared__ float lds[4 * kBlock];
    const int i = pcm_job_of((int)blockIdx.x, b.n, [&](int q) { return b.j[q].blk0; });
    const ColsumJob &J = b.j[i];
    asm volatile("" ::"s"(J.a.g[0]), "s"(J.a.g[1]), "s"(J.a.g[2]), "s"(J.a.ld[0]), "s"(J.a.ld[1]), "s"(J.a.ld[2]), "s"(J.rows), "s"(J.rps), "s"(J.partial), "s"(J.C), "s"(J.ntensors), "s"(J.is_bf16), "s"(J.blk0));  // one batch: "Kernel heads", pcm_common.hpp
    const int local = (int)blockIdx.x - J.blk0;
    const int slot = local / J.ntensors, t = local - slot * J.ntensors;
    const void *g = t == 0 ? J.a.g[0] : (t == 1 ? J.a.g[1] : J.a.g[2]);
    const long ld = t == 0 ? J.a.ld[0] : (t == 1 ? J.a.ld[1] : J.a.ld[2]);
    if (J.is_bf16)
        colsum_body<__hip_bfloat16>(J.rows, J.C, J.rps, (const __hip_bfloat16 *)g, ld, slot, t, J.ntensors, J.partial, lds);
    else
        colsum_body<float>(J.rows, J.C, J.rps, (const float *)g, ld, slot, t, J.ntensors, J.partial, lds);
}

template <typename TO>
__global__ __launch_bounds__(512) void pcm_colsum_reduce_kernel(int nslots, int VH, const float *__restrict__ partial,
                                                                TO *__restrict__ out)
{
    __shared__ double red[8][64];
    asm volatile("" ::"s"(nslots), "s"(VH), "s"(partial), "s"(out));  // "Kernel heads", pcm_common.hpp
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    double acc = 0.0;
    if (e < VH) acc = pcm_slot_sum(partial, (size_t)VH, e, wave, 8, nslots);
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && e < VH) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w][lane];
        if constexpr (sizeof(TO) == 2) out[e] = __float2bfloat16((float)t);
        else out[e] = (float)t;
    }
}

// Several closing reductions in ONE launch (policy/deferred.py): every workgroup finds its reduction in a table passed by
// value (so a captured graph keeps it) and sums 64 columns exactly as pcm_colsum_reduce_kernel / pcm_drln_reduce_kernel /
// pcm_ffn_reduce_kernel do -- same wave striding, same fp64 order -- so deferring a reduction never changes a bit.
constexpr int kReduceBatch = 24;
struct ReduceDesc {
    const float *partial;
    float *out;              // fp32 result (may be NULL)
    __hip_bfloat16 *out16;   // bf16 copy of the columns [from16, VH) (may be NULL)
    int nslots, VH, from16, blk0;  // blk0: first workgroup of this reduction
};
struct ReduceBatch {
    ReduceDesc d[kReduceBatch];
    int n;
};
__global__ __launch_bounds__(512) void pcm_reduce_batch_kernel(ReduceBatch b)
{
    __shared__ double red[8][64];
    const int i = pcm_job_of((int)blockIdx.x, b.n, [&](int j) { return b.d[j].blk0; });
    const ReduceDesc &D = b.d[i];
    asm volatile("" ::"s"(D.partial), "s"(D.out), "s"(D.out16), "s"(D.nslots), "s"(D.VH), "s"(D.from16), "s"(D.blk0));  // one batch: "Kernel heads", pcm_common.hpp
    const float *__restrict__ partial = D.partial;
    const int VH = D.VH, nslots = D.nslots;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = ((int)blockIdx.x - D.blk0) * 64 + lane;
    double acc = 0.0;
    if (e < VH) acc = pcm_slot_sum(partial, (size_t)VH, e, wave, 8, nslots);
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && e < VH) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w][lane];
        if (D.out) D.out[e] = (float)t;
        if (D.out16 && e >= D.from16) D.out16[e - D.from16] = __float2bfloat16((float)t);
    }
}

inline int colsum_slots_for(long rows, int C)
{
    const int rpp = kBlock / (C / 4);
    long slots = (rows + (long)rpp * 8 - 1) / ((long)rpp * 8);  // >= 8 rows per thread
    if (slots > 256) slots = 256;
    return (int)(slots < 1 ? 1 : slots);
}

}  // namespace

extern "C" int pcm_add_cast2_hip(long n, long pos_n, const float *x, const float *pos, void *sum_bf16, void *x_bf16, void *stream)
{
    if (n == 0) return PCM_OK;
    if (n < 0 || pos_n <= 0 || n % 4 || pos_n % 4 || n % pos_n) return PCM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pcm_add_cast2_kernel, dim3(ew_grid(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, n / 4, pos_n / 4, x, pos,
                       (__hip_bfloat16 *)sum_bf16, (__hip_bfloat16 *)x_bf16);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_add3_cast2_hip(long n, const void *a_bf16, const void *b_bf16, const float *c_f32, float *out, float *a_f32,
                                  void *stream)
{
    if (n == 0) return PCM_OK;
    if (n < 0 || n % 4) return PCM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pcm_add2_cast_kernel, dim3(ew_grid(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, n / 4,
                       (const __hip_bfloat16 *)a_bf16, (const __hip_bfloat16 *)b_bf16, out, a_f32, c_f32, (const __hip_bfloat16 *)nullptr);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_add4_cast2_hip(long n, const void *a_bf16, const void *a2_bf16, const void *b_bf16, const float *c_f32, float *out,
                                  float *a_f32, void *stream)
{
    // out = (f32(a) + f32(a2)) + f32(b) [+ c]; a_f32 (nullable) = f32(a) + f32(a2)
    if (n == 0) return PCM_OK;
    if (n < 0 || n % 4 || !a_bf16 || !a2_bf16 || !b_bf16 || !out) return PCM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pcm_add2_cast_kernel, dim3(ew_grid(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, n / 4,
                       (const __hip_bfloat16 *)a_bf16, (const __hip_bfloat16 *)b_bf16, out, a_f32, c_f32, (const __hip_bfloat16 *)a2_bf16);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_add2_cast2_hip(long n, const void *a_bf16, const void *b_bf16, float *out, float *a_f32, void *stream)
{
    return pcm_add3_cast2_hip(n, a_bf16, b_bf16, nullptr, out, a_f32, stream);
}

extern "C" int pcm_add2_cast_hip(long n, const void *a_bf16, const void *b_bf16, float *out, void *stream)
{
    return pcm_add2_cast2_hip(n, a_bf16, b_bf16, out, nullptr, stream);
}

__device__ __forceinline__ float pcm_to_float(float v) { return v; }
__device__ __forceinline__ float pcm_to_float(__hip_bfloat16 v) { return __bfloat162float(v); }
__device__ __forceinline__ void pcm_store(float *p, float v) { *p = v; }
__device__ __forceinline__ void pcm_store(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }

// ---- ACT training loss (act.py:281-291 + loss/misc.py:10-26) as ONE workgroup: the values, and the gradients per unit of upstream
// gradient, in a fixed summation order (same bits on every run).  ~27 framework launches of 1-element tensors otherwise.
//   action = mean_{b,q,a}( (a_hat - actions)^2 * !is_pad[b,q] )           (MSELoss(reduction="none"), mean over ALL elements)
//   kl     = mean_b sum_d -0.5 (1 + logvar - mu^2 - exp(logvar))
//   loss   = action + kl_weight * kl
// stats[3] = {loss, action, kl};  ga (n) = d action / d a_hat,  gmu / glv (bd) = d kl / d mu, d kl / d logvar  (all fp32)
template <typename TA, typename TL>
__global__ __launch_bounds__(1024) void pcm_act_loss_kernel(int n, int A, int bd, int B, const TA *__restrict__ a_hat,
                                                            const float *__restrict__ actions, const unsigned char *__restrict__ is_pad,
                                                            const TL *__restrict__ mu, const TL *__restrict__ logvar, float kl_weight,
                                                            float *__restrict__ stats, float *__restrict__ ga, float *__restrict__ gmu,
                                                            float *__restrict__ glv)
{
    __shared__ double red[2][1024];
    const int t = threadIdx.x;
    double sa = 0.0, sk = 0.0;
    const float inv_n = 1.f / (float)n, inv_b = 1.f / (float)B;
    for (int i = t; i < n; i += 1024) {
        const float e = pcm_to_float(a_hat[i]) - actions[i];
        const float w = is_pad[i / A] ? 0.f : 1.f;
        sa += (double)(e * e * w);
        ga[i] = 2.f * e * w * inv_n;
    }
    for (int i = t; i < bd; i += 1024) {
        const float m = pcm_to_float(mu[i]), lv = pcm_to_float(logvar[i]);
        const float ex = expf(lv);
        sk += (double)(-0.5f * (1.f + lv - m * m - ex));
        gmu[i] = m * inv_b;
        glv[i] = -0.5f * (1.f - ex) * inv_b;
    }
    red[0][t] = sa, red[1][t] = sk;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (t < off) red[0][t] += red[0][t + off], red[1][t] += red[1][t + off];
        __syncthreads();
    }
    if (t == 0) {
        const float action = (float)(red[0][0] / (double)n), kl = (float)(red[1][0] / (double)B);
        stats[0] = action + kl * kl_weight, stats[1] = action, stats[2] = kl;
    }
}

// gradients for the upstream gradients g[0..2] of (loss, action, kl) -- device scalars, NULL = 0:
//   d a_hat = (g0 + g1) ga,   d mu = (g0 kl_weight + g2) gmu,   d logvar = (g0 kl_weight + g2) glv,  rounded to the input dtypes
template <typename TA, typename TL>
__global__ __launch_bounds__(256) void pcm_act_loss_bwd_kernel(int n, int bd, const float *__restrict__ g0, const float *__restrict__ g1,
                                                                const float *__restrict__ g2, float kl_weight, const float *__restrict__ ga,
                                                                const float *__restrict__ gmu, const float *__restrict__ glv,
                                                                TA *__restrict__ da, TL *__restrict__ dmu, TL *__restrict__ dlv)
{
    const float a = (g0 ? g0[0] : 0.f) + (g1 ? g1[0] : 0.f), k = (g0 ? g0[0] : 0.f) * kl_weight + (g2 ? g2[0] : 0.f);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n + bd; i += gridDim.x * 256) {
        if (i < n) {
            pcm_store(da + i, a * ga[i]);
        } else {
            pcm_store(dmu + (i - n), k * gmu[i - n]);
            pcm_store(dlv + (i - n), k * glv[i - n]);
        }
    }
}

// sine position embedding, act.py:467-506 with its default arguments (layout: see include/pcm_pointops.h)
__global__ __launch_bounds__(256) void pcm_coord_embed_sine_kernel(long total, int H, int npf, const float *__restrict__ coord,
                                                                    const float *__restrict__ dim_t, float *__restrict__ out)
{
    const int k = npf / 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / H;
        const int c = (int)(e - r * H);
        float v = 0.f;
        if (c < 3 * npf) {
            const int a = c / npf, i = c - a * npf;
            const float x = coord[r * 3 + a];
            v = i < k ? sinf(x / dim_t[2 * i]) : cosf(x / dim_t[2 * (i - k) + 1]);
        }
        out[e] = v;
    }
}

extern "C" int pcm_act_loss_forward_hip(int n, int A, int bd, int B, int a_is_bf16, const void *a_hat, const float *actions,
                                        const unsigned char *is_pad, int l_is_bf16, const void *mu, const void *logvar, float kl_weight,
                                        float *stats, float *ga, float *gmu, float *glv, void *stream)
{
    if (n <= 0 || A <= 0 || n % A || bd <= 0 || B <= 0 || bd % B) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
#define PCM_L(TA, TL)                                                                                                       \
    hipLaunchKernelGGL((pcm_act_loss_kernel<TA, TL>), dim3(1), dim3(1024), 0, st, n, A, bd, B, (const TA *)a_hat, actions, is_pad, \
                       (const TL *)mu, (const TL *)logvar, kl_weight, stats, ga, gmu, glv)
    if (a_is_bf16) {
        if (l_is_bf16) PCM_L(__hip_bfloat16, __hip_bfloat16); else PCM_L(__hip_bfloat16, float);
    } else {
        if (l_is_bf16) PCM_L(float, __hip_bfloat16); else PCM_L(float, float);
    }
#undef PCM_L
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_act_loss_backward_hip(int n, int bd, const float *g_loss, const float *g_action, const float *g_kl, float kl_weight,
                                         const float *ga, const float *gmu, const float *glv, int a_is_bf16, void *da, int l_is_bf16,
                                         void *dmu, void *dlv, void *stream)
{
    if (n <= 0 || bd <= 0) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (n + bd + 255) / 256;
#define PCM_L(TA, TL)                                                                                                          \
    hipLaunchKernelGGL((pcm_act_loss_bwd_kernel<TA, TL>), dim3(blocks), dim3(256), 0, st, n, bd, g_loss, g_action, g_kl, kl_weight, ga, gmu, \
                       glv, (TA *)da, (TL *)dmu, (TL *)dlv)
    if (a_is_bf16) {
        if (l_is_bf16) PCM_L(__hip_bfloat16, __hip_bfloat16); else PCM_L(__hip_bfloat16, float);
    } else {
        if (l_is_bf16) PCM_L(float, __hip_bfloat16); else PCM_L(float, float);
    }
#undef PCM_L
    return PCM_LAUNCH_STATUS();
}

namespace {
// ---- CVAE latent head (act.py:175-181 + act/utils.py:36-39): mu | logvar = split(latent_info), z = mu + exp(logvar / 2) * eps,
// with the framework's roundings under bf16 autocast reproduced: logvar / 2 in the input dtype, exp and the product in fp32,
// the sum in fp32.  eps is given (parity tests) or drawn here from the counter hash of the dropout masks (Box-Muller on two
// hashes of (seed, site, element)): no framework RNG inside the captured step.  Also emits contiguous copies of mu and
// logvar for the KL term.  One launch each way (the framework chain: 9 forward, ~12 backward launches on 256 elements).
__device__ __forceinline__ float hash_normal(uint64_t seed, uint32_t site, uint32_t e)
{
    const uint32_t k = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ (site * 0x85EBCA6Bu);
    const uint32_t h1 = mix32(mix32((2u * e) ^ k) + k), h2 = mix32(mix32((2u * e + 1u) ^ k) + k);
    const float u1 = ((float)(h1 >> 8) + 1.f) * (1.f / 16777216.f);  // (0, 1]
    const float u2 = (float)(h2 >> 8) * (1.f / 16777216.f);          // [0, 1)
    return sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

template <typename T>
__global__ __launch_bounds__(256) void pcm_cvae_latent_fwd_kernel(int B, int D, const T *__restrict__ info, const float *__restrict__ eps_in,
                                                                  const long *__restrict__ seed_ptr, unsigned site, float *__restrict__ z,
                                                                  T *__restrict__ mu_c, T *__restrict__ lv_c, float *__restrict__ eps_out,
                                                                  float *__restrict__ std_out)
{
    const int n = B * D;
    const uint64_t seed = eps_in == nullptr ? (uint64_t)seed_ptr[0] : 0ull;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int b = i / D, d = i - b * D;
        const T m = info[(size_t)b * 2 * D + d], lv = info[(size_t)b * 2 * D + D + d];
        T half;
        pcm_store(&half, pcm_to_float(lv) / 2.f);  // logvar.div(2) in the input dtype
        const float sd = expf(pcm_to_float(half));
        const float e = eps_in != nullptr ? eps_in[i] : hash_normal(seed, site, (uint32_t)i);
        const float prod = sd * e;
        z[i] = pcm_to_float(m) + prod;
        mu_c[i] = m, lv_c[i] = lv;
        eps_out[i] = e, std_out[i] = sd;
    }
}

// d_info[b][d] = dz (rounded to T) + dmu ; d_info[b][D + d] = T((dz * eps) * std) / 2 + dlv   (each sum rounded to T: what the
// engine's accumulation of the two gradients of mu / logvar does)
template <typename T>
__global__ __launch_bounds__(256) void pcm_cvae_latent_bwd_kernel(int B, int D, const float *__restrict__ dz, const T *__restrict__ dmu,
                                                                  const T *__restrict__ dlv, const float *__restrict__ eps,
                                                                  const float *__restrict__ sd, T *__restrict__ dinfo)
{
    const int n = B * D;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int b = i / D, d = i - b * D;
        float gm = 0.f, gl = 0.f;
        if (dz != nullptr) {
            T t;
            pcm_store(&t, dz[i]);
            gm = pcm_to_float(t);
            const float a = dz[i] * eps[i];
            pcm_store(&t, a * sd[i]);
            T h;
            pcm_store(&h, pcm_to_float(t) / 2.f);
            gl = pcm_to_float(h);
        }
        if (dmu != nullptr) gm = dz != nullptr ? gm + pcm_to_float(dmu[i]) : pcm_to_float(dmu[i]);
        if (dlv != nullptr) gl = dz != nullptr ? gl + pcm_to_float(dlv[i]) : pcm_to_float(dlv[i]);
        pcm_store(dinfo + (size_t)b * 2 * D + d, gm);
        pcm_store(dinfo + (size_t)b * 2 * D + D + d, gl);
    }
}
}  // namespace

extern "C" int pcm_cvae_latent_forward_hip(int B, int D, int is_bf16, const void *latent_info, const float *eps_in, const long *seed,
                                           unsigned site, float *z, void *mu, void *logvar, float *eps_out, float *std_out, void *stream)
{
    if (B < 0 || D <= 0 || (long)B * D > 0x7FFFFFFF) return PCM_ERR_BAD_ARG;
    if (B == 0) return PCM_OK;
    if (!latent_info || !z || !mu || !logvar || !eps_out || !std_out || (!eps_in && !seed)) return PCM_ERR_BAD_ARG;
    const int n = B * D;
    const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    hipStream_t st = (hipStream_t)stream;
    if (is_bf16)
        hipLaunchKernelGGL(pcm_cvae_latent_fwd_kernel<__hip_bfloat16>, dim3(blocks), dim3(256), 0, st, B, D, (const __hip_bfloat16 *)latent_info,
                           eps_in, seed, site, z, (__hip_bfloat16 *)mu, (__hip_bfloat16 *)logvar, eps_out, std_out);
    else
        hipLaunchKernelGGL(pcm_cvae_latent_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, B, D, (const float *)latent_info, eps_in, seed,
                           site, z, (float *)mu, (float *)logvar, eps_out, std_out);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_cvae_latent_backward_hip(int B, int D, int is_bf16, const float *dz, const void *dmu, const void *dlogvar,
                                            const float *eps, const float *std_, void *d_latent_info, void *stream)
{
    if (B < 0 || D <= 0 || (long)B * D > 0x7FFFFFFF) return PCM_ERR_BAD_ARG;
    if (B == 0) return PCM_OK;
    if (!d_latent_info || (dz && (!eps || !std_))) return PCM_ERR_BAD_ARG;
    const int n = B * D;
    const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    hipStream_t st = (hipStream_t)stream;
    if (is_bf16)
        hipLaunchKernelGGL(pcm_cvae_latent_bwd_kernel<__hip_bfloat16>, dim3(blocks), dim3(256), 0, st, B, D, dz, (const __hip_bfloat16 *)dmu,
                           (const __hip_bfloat16 *)dlogvar, eps, std_, (__hip_bfloat16 *)d_latent_info);
    else
        hipLaunchKernelGGL(pcm_cvae_latent_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, B, D, dz, (const float *)dmu,
                           (const float *)dlogvar, eps, std_, (float *)d_latent_info);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_coord_embed_sine_hip(long m, int H, int npf, const float *coord, const float *dim_t, float *out, void *stream)
{
    if (m < 0 || H <= 0 || npf <= 0 || (npf & 1) || 3 * npf > H) return PCM_ERR_BAD_ARG;
    if (m == 0) return PCM_OK;
    const long total = m * (long)H;
    long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pcm_coord_embed_sine_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, total, H, npf, coord, dim_t, out);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_slab_sum_hip(int nslabs, long n, const float *partial, int out_is_bf16, void *out, void *stream)
{
    // out[e] = sum_s partial[s][e]: the closing reduction of a split-K product (policy/rows_linear.py), fp64 accumulation in a
    // fixed order, rounded ONCE to the output dtype
    if (nslabs <= 0 || n < 0 || n > 0x7FFFFFFF) return PCM_ERR_BAD_ARG;
    if (n == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    const int VH = (int)n;
    if (out_is_bf16)
        hipLaunchKernelGGL(pcm_colsum_reduce_kernel<__hip_bfloat16>, dim3((VH + 63) / 64), dim3(512), 0, s, nslabs, VH, partial,
                           (__hip_bfloat16 *)out);
    else
        hipLaunchKernelGGL(pcm_colsum_reduce_kernel<float>, dim3((VH + 63) / 64), dim3(512), 0, s, nslabs, VH, partial, (float *)out);
    return PCM_LAUNCH_STATUS();
}

namespace {
// Several device-to-device copies in ONE launch (a training step stages ~20 small input / index tensors before it replays its
// graphs: a copy launch each, or one multi-tensor launch per dtype at 11-14 us, otherwise).  Table by value; a workgroup moves
// 4 KiB: 16-byte vectors when source, destination and length allow, bytes otherwise.
constexpr int kCopyBatch = 32;
struct CopyJob {
    char *dst;
    const char *src;
    long nbytes;
    int blk0;
};
struct CopyBatch {
    CopyJob j[kCopyBatch];
    int n;
};
__global__ __launch_bounds__(256) void pcm_copy_batch_kernel(CopyBatch b)
{
    const int i = pcm_job_of((int)blockIdx.x, b.n, [&](int q) { return b.j[q].blk0; });
    const CopyJob &J = b.j[i];
    asm volatile("" ::"s"(J.dst), "s"(J.src), "s"(J.nbytes), "s"(J.blk0));  // one batch: "Kernel heads", pcm_common.hpp
    const long off = (long)((int)blockIdx.x - J.blk0) * 4096;
    const long len = J.nbytes - off < 4096 ? J.nbytes - off : 4096;
    char *d = J.dst + off;
    const char *s = J.src + off;
    if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0 && len == 4096) {
        reinterpret_cast<uint4 *>(d)[threadIdx.x] = reinterpret_cast<const uint4 *>(s)[threadIdx.x];
    } else {
        for (long k = threadIdx.x; k < len; k += 256) d[k] = s[k];
    }
}
}  // namespace

namespace {
constexpr int kIncrBatch = 64;
struct IncrBatch {
    long *p[kIncrBatch];
    int n;
};
__global__ void pcm_incr_i64_batch_kernel(IncrBatch b)
{
    const int i = threadIdx.x;
    if (i < b.n) *b.p[i] += 1;
}
}  // namespace

extern "C" int pcm_incr_i64_batch_hip(int n, void *const *counters, void *stream)
{
    // *counters[i] += 1 for n distinct device int64 counters, 64 per launch: BatchNorm's num_batches_tracked of every layer of a
    // forward pass (torch.nn.BatchNorm1d increments each with its own one-element launch, pointnet.py:32-46)
    if (n < 0 || (n > 0 && !counters)) return PCM_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i)
        if (!counters[i]) return PCM_ERR_BAD_ARG;
    if (n == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += kIncrBatch) {
        IncrBatch b;
        b.n = n - base < kIncrBatch ? n - base : kIncrBatch;
        for (int i = 0; i < b.n; ++i) b.p[i] = (long *)counters[base + i];
        hipLaunchKernelGGL(pcm_incr_i64_batch_kernel, dim3(1), dim3(64), 0, s, b);
    }
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_copy_batch_hip(int n, void *const *dst, const void *const *src, const long *nbytes, void *stream)
{
    // dst[i][0..nbytes[i]) = src[i][0..nbytes[i]) for n device buffers (host arrays of pointers / sizes), 32 per launch.
    // Buffers of one pair must not overlap.
    if (n < 0 || (n > 0 && (!dst || !src || !nbytes))) return PCM_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i)
        if (nbytes[i] < 0 || (nbytes[i] > 0 && (!dst[i] || !src[i]))) return PCM_ERR_BAD_ARG;
    if (n == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += kCopyBatch) {
        CopyBatch b;
        b.n = 0;
        long blocks = 0;
        for (int i = base; i < n && i < base + kCopyBatch; ++i) {
            if (nbytes[i] == 0) continue;
            CopyJob &J = b.j[b.n++];
            J.dst = (char *)dst[i], J.src = (const char *)src[i], J.nbytes = nbytes[i], J.blk0 = (int)blocks;
            blocks += (nbytes[i] + 4095) / 4096;
        }
        if (blocks > 0x7FFFFFFF) return PCM_ERR_BAD_ARG;
        if (b.n) hipLaunchKernelGGL(pcm_copy_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, b);
    }
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_colsum_batch_hip(int n, const long *rows, const int *C, const int *ntensors, const int *in_is_bf16,
                                    const void *const *g, const long *ld, void *const *partial, void *stream)
{
    // first stages of n column sums (the arguments of pcm_colsum_hip, as host arrays; g / ld hold 3 entries per job), 16 per
    // launch; partial[i] receives pcm_colsum_slots(rows[i], C[i]) rows of ntensors[i] * C[i] sums for pcm_reduce_batch_hip
    if (n < 0 || (n > 0 && (!rows || !C || !ntensors || !in_is_bf16 || !g || !ld || !partial))) return PCM_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i) {
        if (rows[i] <= 0 || ntensors[i] < 1 || ntensors[i] > 3 || !partial[i]) return PCM_ERR_BAD_ARG;
        if (C[i] <= 0 || C[i] % 4 || C[i] > 1024) return PCM_ERR_UNSUPPORTED;
        for (int t = 0; t < ntensors[i]; ++t)
            if (!g[3 * i + t]) return PCM_ERR_BAD_ARG;
    }
    if (n == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += kColsumBatch) {
        ColsumBatch b;
        b.n = 0;
        long blocks = 0;
        for (int i = base; i < n && b.n < kColsumBatch; ++i) {
            ColsumJob &J = b.j[b.n++];
            for (int t = 0; t < 3; ++t) J.a.g[t] = g[3 * i + t], J.a.ld[t] = ld[3 * i + t];
            const int slots = colsum_slots_for(rows[i], C[i]);
            J.rows = rows[i], J.rps = (rows[i] + slots - 1) / slots, J.partial = (float *)partial[i];
            J.C = C[i], J.ntensors = ntensors[i], J.is_bf16 = in_is_bf16[i], J.blk0 = (int)blocks;
            blocks += (long)slots * ntensors[i];
        }
        if (blocks > 0x7FFFFFFF) return PCM_ERR_BAD_ARG;
        hipLaunchKernelGGL(pcm_colsum_batch_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, b);
    }
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_reduce_batch_hip(int n, const void *const *partial, const int *nslots, const int *width, void *const *out_f32,
                                    void *const *out_bf16, const int *bf16_from, void *stream)
{
    // n closing reductions out[e] = sum_s partial[s * width + e] (s < nslots), kReduceBatch per launch.  out_f32[i] and / or
    // out_bf16[i] receive the sums (bf16: columns bf16_from[i]..width-1, stored from index 0).  Host arrays.
    if (n < 0 || (n > 0 && (!partial || !nslots || !width || !out_f32 || !out_bf16 || !bf16_from))) return PCM_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i)
        if (!partial[i] || nslots[i] <= 0 || width[i] < 0 || (!out_f32[i] && !out_bf16[i]) || bf16_from[i] < 0) return PCM_ERR_BAD_ARG;
    if (n == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += kReduceBatch) {
        ReduceBatch b;
        b.n = 0;
        long blocks = 0;
        for (int i = base; i < n && i < base + kReduceBatch; ++i) {  // (a zero-width job is skipped, never visited twice)
            if (width[i] == 0) continue;
            ReduceDesc &d = b.d[b.n++];
            d.partial = (const float *)partial[i], d.out = (float *)out_f32[i], d.out16 = (__hip_bfloat16 *)out_bf16[i];
            d.nslots = nslots[i], d.VH = width[i], d.from16 = bf16_from[i], d.blk0 = (int)blocks;
            blocks += (width[i] + 63) / 64;
        }
        if (blocks > 0x7FFFFFFF) return PCM_ERR_BAD_ARG;
        if (b.n) hipLaunchKernelGGL(pcm_reduce_batch_kernel, dim3((unsigned)blocks), dim3(512), 0, s, b);
    }
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_colsum_slots(long rows, int C)
{
    if (rows <= 0 || C <= 0 || C % 4 || C > 1024) return 0;
    return colsum_slots_for(rows, C);
}

extern "C" int pcm_colsum_hip(long rows, int C, int ntensors, int in_is_bf16, const void *g0, long ld0, const void *g1, long ld1,
                              const void *g2, long ld2, float *partial, int out_is_bf16, void *out, void *stream)
{
    if (rows <= 0 || ntensors < 1 || ntensors > 3) return PCM_ERR_BAD_ARG;
    if (C <= 0 || C % 4 || C > 1024) return PCM_ERR_UNSUPPORTED;
    ColsumArgs a;
    a.g[0] = g0, a.g[1] = g1, a.g[2] = g2;
    a.ld[0] = ld0, a.ld[1] = ld1, a.ld[2] = ld2;
    const int slots = colsum_slots_for(rows, C);
    const long rps = (rows + slots - 1) / slots;
    hipStream_t s = (hipStream_t)stream;
    if (in_is_bf16)
        hipLaunchKernelGGL(pcm_colsum_kernel<__hip_bfloat16>, dim3(slots, ntensors), dim3(kBlock), 0, s, rows, C, rps, a, partial);
    else
        hipLaunchKernelGGL(pcm_colsum_kernel<float>, dim3(slots, ntensors), dim3(kBlock), 0, s, rows, C, rps, a, partial);
    const int VH = ntensors * C;
    if (out == nullptr) return PCM_LAUNCH_STATUS();  // partial sums only: the caller closes them with pcm_reduce_batch_hip
    if (out_is_bf16)
        hipLaunchKernelGGL(pcm_colsum_reduce_kernel<__hip_bfloat16>, dim3((VH + 63) / 64), dim3(512), 0, s, slots, VH, partial,
                           (__hip_bfloat16 *)out);
    else
        hipLaunchKernelGGL(pcm_colsum_reduce_kernel<float>, dim3((VH + 63) / 64), dim3(512), 0, s, slots, VH, partial, (float *)out);
    return PCM_LAUNCH_STATUS();
}
