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
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        float a[4], p[4], s[4];
        load4<float>(x + i * 4, a);
        load4<float>(pos + (i % pos4) * 4, p);
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = a[u] + p[u];
        store4<__hip_bfloat16>(sum16 + i * 4, s);
        if (x16 != nullptr) store4<__hip_bfloat16>(x16 + i * 4, a);
    }
}

__global__ __launch_bounds__(kBlock) void pcm_add2_cast_kernel(long n4, const __hip_bfloat16 *__restrict__ a,
                                                               const __hip_bfloat16 *__restrict__ b, float *__restrict__ out,
                                                               float *__restrict__ a32, const float *__restrict__ c32)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        float x[4], y[4], o[4];
        load4<__hip_bfloat16>(a + i * 4, x);
        load4<__hip_bfloat16>(b + i * 4, y);
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = x[u] + y[u];
        if (c32 != nullptr) {  // a third, fp32 addend: the residual branch's gradient of the same tensor
            float z[4];
            load4<float>(c32 + i * 4, z);
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

// grid (slots, ntensors); a thread owns 4 channels of one tensor; rows_per_pass = 256 / (C/4) rows in flight
template <typename T>
__global__ __launch_bounds__(kBlock) void pcm_colsum_kernel(long rows, int C, long rows_per_slot, ColsumArgs args,
                                                            float *__restrict__ partial)
{
    __shared__ float lds[4 * kBlock];
    const int t = blockIdx.y;
    const T *__restrict__ g = (const T *)args.g[t];
    const long ld = args.ld[t];
    const int lpr = C / 4, rpp = kBlock / lpr;
    const int col4 = threadIdx.x % lpr, rsub = threadIdx.x / lpr;
    const bool act = rsub < rpp;
    const long r0 = (long)blockIdx.x * rows_per_slot;
    const long r1 = r0 + rows_per_slot < rows ? r0 + rows_per_slot : rows;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (act)
        for (long r = r0 + rsub; r < r1; r += rpp) {
            float v[4];
            load4<T>(g + r * ld + col4 * 4, v);
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += v[u];
        }
#pragma unroll
    for (int u = 0; u < 4; ++u) lds[u * kBlock + threadIdx.x] = s[u];
    __syncthreads();
    if (act && rsub == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc = 0.f;
            for (int rr = 0; rr < rpp; ++rr) acc += lds[u * kBlock + rr * lpr + col4];
            partial[((size_t)blockIdx.x * gridDim.y + t) * C + col4 * 4 + u] = acc;
        }
    }
}

template <typename TO>
__global__ __launch_bounds__(512) void pcm_colsum_reduce_kernel(int nslots, int VH, const float *__restrict__ partial,
                                                                TO *__restrict__ out)
{
    __shared__ double red[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    double acc = 0.0;
    if (e < VH)
        for (int s = wave; s < nslots; s += 8) acc += (double)partial[(size_t)s * VH + e];
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
                       (const __hip_bfloat16 *)a_bf16, (const __hip_bfloat16 *)b_bf16, out, a_f32, c_f32);
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
    if (out_is_bf16)
        hipLaunchKernelGGL(pcm_colsum_reduce_kernel<__hip_bfloat16>, dim3((VH + 63) / 64), dim3(512), 0, s, slots, VH, partial,
                           (__hip_bfloat16 *)out);
    else
        hipLaunchKernelGGL(pcm_colsum_reduce_kernel<float>, dim3((VH + 63) / 64), dim3(512), 0, s, slots, VH, partial, (float *)out);
    return PCM_LAUNCH_STATUS();
}
