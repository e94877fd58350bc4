// bnact.hip -- BatchNorm1d (batch statistics) over rows (n, C) WITHOUT the ReLU behind it, for gfx950.  HBM bound.
//
// The Diffusion Policy's projector ends with a bare BatchNorm (/root/reference/src/models/components/diffusion_policy/vision/
// pcd_obs_encoder.py:100-120: ... -> MaxPool1d(M) -> Conv1d(k=1) -> BatchNorm1d).  bnrelu.hip serves BatchNorm + ReLU; its gfx950 code is
// byte-identical to the build the round-4 hardware suite ran green (profiles/r04_device_digest.txt) and is therefore NOT edited.  This
// file re-states its kernels with the gate removed (z = a y + b; g = dz), same plans, same loops, same reduction order:
//   pcm_bn_act_forward_hip / pcm_bn_act_backward_hip (relu != 0)  ->  pcm_bn_relu_forward_hip / pcm_bn_relu_backward_hip (bnrelu.hip, unchanged)
//   (relu == 0)                                                   ->  the pcm_bnact_* kernels below (first hardware contact: round 6+)
// Round 5's variant of both (four rows in flight in the column sums, kernel-head statements) is csrc/next/bnrelu.hip (`make next`).
#include "pcm_elem.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxChunk = 1024;  // channels per block: 256 threads x 4
constexpr int kMaxSlots = 1024;

struct RowMap {
    int lpr, rpp, col4, rsub;
    bool active;
    __device__ RowMap(int chunkW)
    {
        lpr = chunkW / 4;       // lanes per row
        rpp = kBlock / lpr;     // rows in flight per pass
        col4 = threadIdx.x % lpr;
        rsub = threadIdx.x / lpr;
        active = rsub < rpp;
    }
};

// MODE 0: (y, y^2)        MODE 2: (g, g*xhat) with g = dz (no ReLU gate)
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void pcm_bnact_colsum_kernel(long n, int C, int chunkW, long rows_per_slot,
                                                               const T *__restrict__ y, const T *__restrict__ dz,
                                                               const float *__restrict__ stat, float *__restrict__ partial)
{
    __shared__ float lds[8 * kBlock];
    const RowMap mp(chunkW);
    const int c0 = blockIdx.y * chunkW + mp.col4 * 4;
    const bool act = mp.active && c0 < C;
    const long r_begin = (long)blockIdx.x * rows_per_slot;
    const long r_end = r_begin + rows_per_slot < n ? r_begin + rows_per_slot : n;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (act) {
        float mean[4], invstd[4], a[4], b[4];
        if (MODE != 0) {
            load4<float>(stat + c0, mean);
            load4<float>(stat + C + c0, invstd);
            load4<float>(stat + 2 * C + c0, a);
            load4<float>(stat + 3 * C + c0, b);
        }
        float sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) load4<T>(y + c0, sh);  // shift by row 0: sum (y - sh), sum (y - sh)^2 do not cancel when |mean| >> std
        for (long r = r_begin + mp.rsub; r < r_end; r += mp.rpp) {
            float v[4];
            load4<T>(y + r * C + c0, v);
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d = v[u] - sh[u];
                    s0[u] += d;
                    s1[u] += d * d;
                }
            } else {
                float d[4];
                load4<T>(dz + r * C + c0, d);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float g = d[u];
                    s0[u] += g;
                    s1[u] += g * ((v[u] - mean[u]) * invstd[u]);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        lds[u * kBlock + threadIdx.x] = s0[u];
        lds[(4 + u) * kBlock + threadIdx.x] = s1[u];
    }
    __syncthreads();
    if (act && mp.rsub == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float acc = 0.f;
                for (int rr = 0; rr < mp.rpp; ++rr) acc += lds[(k * 4 + u) * kBlock + rr * mp.lpr + mp.col4];
                partial[((size_t)blockIdx.x * 2 + k) * C + c0 + u] = acc;
            }
    }
}

// out[e] = sum over slots of partial[slot][e], e in [0, 2C), accumulated in fp64
constexpr int kRedWaves = 16;
__global__ __launch_bounds__(64 * kRedWaves) void pcm_bnact_reduce_kernel(int nslots, int VH, const float *__restrict__ partial,
                                                                         float *__restrict__ out)
{
    __shared__ double red[kRedWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // four independent chains: four loads in flight per lane
    if (e < VH) {
        int s = wave;
        for (; s + 3 * kRedWaves < nslots; s += 4 * kRedWaves) {
            a0 += (double)partial[(size_t)s * VH + e];
            a1 += (double)partial[(size_t)(s + kRedWaves) * VH + e];
            a2 += (double)partial[(size_t)(s + 2 * kRedWaves) * VH + e];
            a3 += (double)partial[(size_t)(s + 3 * kRedWaves) * VH + e];
        }
        for (; s < nslots; s += kRedWaves) a0 += (double)partial[(size_t)s * VH + e];
    }
    red[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && e < VH) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kRedWaves; ++w) t += red[w][lane];
        out[e] = (float)t;
    }
}

// sums[2][C] -> stat[4][C] = { mean, invstd, a = gamma*invstd, b = beta - a*mean } and the running-stat update
template <typename T>
__device__ __forceinline__ void bn_stats_of(int c, int C, double count, float eps, float momentum, const T *__restrict__ y, float s1,
                                            float s2, const float *__restrict__ gamma, const float *__restrict__ beta,
                                            float *__restrict__ stat, float *__restrict__ running_mean,
                                            float *__restrict__ running_var)
{
    float shv;
    if constexpr (sizeof(T) == 2) shv = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(y)[c] << 16);
    else shv = (float)y[c];
    const double dm = (double)s1 / count;  // mean of (y - shift), shift = row 0 of y (see pcm_bnact_colsum_kernel)
    const double mean = (double)shv + dm;
    double var = (double)s2 / count - dm * dm;  // biased: what the normalisation uses
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float a = gamma[c] * invstd;
    stat[c] = (float)mean;
    stat[C + c] = invstd;
    stat[2 * C + c] = a;
    stat[3 * C + c] = beta[c] - a * (float)mean;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// pcm_bnact_reduce_kernel + pcm_bnact_stats_kernel in one launch (single-rank training): a workgroup reduces both moments of 64
// channels -- the same chains and order as pcm_bnact_reduce_kernel, so the same bits -- and its first wave finishes the
// statistics.  One launch less per BatchNorm layer (6 per ACT step, each on the critical path of the tokenizer).
template <typename T>
__global__ __launch_bounds__(64 * kRedWaves) void pcm_bnact_reduce_stats_kernel(int nslots, int C, const float *__restrict__ partial,
                                                                               float *__restrict__ sums, double count, float eps,
                                                                               float momentum, const T *__restrict__ y,
                                                                               const float *__restrict__ gamma,
                                                                               const float *__restrict__ beta, float *__restrict__ stat,
                                                                               float *__restrict__ running_mean,
                                                                               float *__restrict__ running_var)
{
    __shared__ double red[2][kRedWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int VH = 2 * C;
#pragma unroll
    for (int mom = 0; mom < 2; ++mom) {
        const int e = mom * C + c;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (c < C) {
            int s = wave;
            for (; s + 3 * kRedWaves < nslots; s += 4 * kRedWaves) {
                a0 += (double)partial[(size_t)s * VH + e];
                a1 += (double)partial[(size_t)(s + kRedWaves) * VH + e];
                a2 += (double)partial[(size_t)(s + 2 * kRedWaves) * VH + e];
                a3 += (double)partial[(size_t)(s + 3 * kRedWaves) * VH + e];
            }
            for (; s < nslots; s += kRedWaves) a0 += (double)partial[(size_t)s * VH + e];
        }
        red[mom][wave][lane] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (wave == 0 && c < C) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int w = 0; w < kRedWaves; ++w) t1 += red[0][w][lane], t2 += red[1][w][lane];
        const float s1 = (float)t1, s2 = (float)t2;
        sums[c] = s1, sums[C + c] = s2;
        bn_stats_of<T>(c, C, count, eps, momentum, y, s1, s2, gamma, beta, stat, running_mean, running_var);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void pcm_bnact_stats_kernel(int C, double count, float eps, float momentum, const T *__restrict__ y,
                                                              const float *__restrict__ sums, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *__restrict__ stat,
                                                              float *__restrict__ running_mean, float *__restrict__ running_var)
{
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    bn_stats_of<T>(c, C, count, eps, momentum, y, sums[c], sums[C + c], gamma, beta, stat, running_mean, running_var);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void pcm_bnact_apply_kernel(long total4, int C, const T *__restrict__ y,
                                                                   const float *__restrict__ stat, T *__restrict__ z)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total4; i += (long)gridDim.x * kBlock) {
        const long e = i * 4;
        const int c = (int)(e % C);
        float v[4], a[4], b[4], o[4];
        load4<T>(y + e, v);
        load4<float>(stat + 2 * C + c, a);
        load4<float>(stat + 3 * C + c, b);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t = a[u] * v[u] + b[u];
            o[u] = t;
        }
        store4<T>(z + e, o);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void pcm_bnact_bwd_apply_kernel(long total4, int C, float inv_n, const T *__restrict__ y,
                                                                       const T *__restrict__ dz, const float *__restrict__ stat,
                                                                       const float *__restrict__ sums, T *__restrict__ dy)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total4; i += (long)gridDim.x * kBlock) {
        const long e = i * 4;
        const int c = (int)(e % C);
        float v[4], d[4], mean[4], invstd[4], a[4], b[4], sg[4], sgx[4], o[4];
        load4<T>(y + e, v);
        load4<T>(dz + e, d);
        load4<float>(stat + c, mean);
        load4<float>(stat + C + c, invstd);
        load4<float>(stat + 2 * C + c, a);
        load4<float>(stat + 3 * C + c, b);
        load4<float>(sums + c, sg);
        load4<float>(sums + C + c, sgx);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float g = d[u];
            const float xhat = (v[u] - mean[u]) * invstd[u];
            o[u] = a[u] * (g - sg[u] * inv_n - xhat * (sgx[u] * inv_n));
        }
        store4<T>(dy + e, o);
    }
}

struct Plan {
    int chunkW, nchunk, nslots;
    long rows_per_slot;
};

inline Plan plan_for(long n, int C)
{
    Plan p;
    p.chunkW = C < kMaxChunk ? C : kMaxChunk;
    p.nchunk = (C + p.chunkW - 1) / p.chunkW;
    const int rpp = kBlock / (p.chunkW / 4);
    long slots = (n + (long)rpp * 16 - 1) / ((long)rpp * 16);  // >= 16 rows per thread
    if (slots > kMaxSlots) slots = kMaxSlots;
    if (slots < 1) slots = 1;
    p.rows_per_slot = (n + slots - 1) / slots;
    p.nslots = (int)((n + p.rows_per_slot - 1) / p.rows_per_slot);
    return p;
}

inline int ew_grid(long total4)
{
    long blocks = (total4 + kBlock - 1) / kBlock;
    if (blocks > 256L * 16) blocks = 256L * 16;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int pcm_bn_relu_supported(long n, int C);
extern "C" int pcm_bn_relu_forward_hip(long n, int C, int is_bf16, const void *y, const float *gamma, const float *beta, float eps,
                                       float momentum, float *running_mean, float *running_var, int use_given_stat, float *partial,
                                       float *sums, float *stat, void *z, void *stream);
extern "C" int pcm_bn_relu_backward_hip(long n, int C, int is_bf16, const void *y, const void *dz, const float *stat, float *partial,
                                        float *sums, void *dy, int phase, double count, void *stream);

// BatchNorm1d over rows with (relu != 0) or without the ReLU behind it; arguments as pcm_bn_relu_forward_hip
extern "C" int pcm_bn_act_forward_hip(long n, int C, int is_bf16, int relu, const void *y, const float *gamma, const float *beta,
                                      float eps, float momentum, float *running_mean, float *running_var, int use_given_stat,
                                      float *partial, float *sums, float *stat, void *z, void *stream)
{
    if (relu)
        return pcm_bn_relu_forward_hip(n, C, is_bf16, y, gamma, beta, eps, momentum, running_mean, running_var, use_given_stat, partial,
                                       sums, stat, z, stream);
    if (n == 0) return PCM_OK;
    if (!pcm_bn_relu_supported(n, C)) return PCM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    using bf = __hip_bfloat16;
    const Plan p = plan_for(n, C);
    // use_given_stat: 0 = whole layer; 1 = apply only (stat given: eval mode, or statistics combined across ranks);
    // 2 = local sums only (sums[0] = sum (y - y[0]), sums[1] = sum (y - y[0])^2: the caller combines them across ranks)
    if (use_given_stat != 1) {
        if (!partial || !sums) return PCM_ERR_BAD_ARG;
        const dim3 grid(p.nslots, p.nchunk);
        if (is_bf16)
            hipLaunchKernelGGL((pcm_bnact_colsum_kernel<bf, 0>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot,
                               (const bf *)y, (const bf *)nullptr, (const float *)nullptr, partial);
        else
            hipLaunchKernelGGL((pcm_bnact_colsum_kernel<float, 0>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot,
                               (const float *)y, (const float *)nullptr, (const float *)nullptr, partial);
        if (use_given_stat == 0) {  // reduce + statistics in one launch
            if (is_bf16)
                hipLaunchKernelGGL(pcm_bnact_reduce_stats_kernel<bf>, dim3((C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, C, partial,
                                   sums, (double)n, eps, momentum, (const bf *)y, gamma, beta, stat, running_mean, running_var);
            else
                hipLaunchKernelGGL(pcm_bnact_reduce_stats_kernel<float>, dim3((C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, C,
                                   partial, sums, (double)n, eps, momentum, (const float *)y, gamma, beta, stat, running_mean,
                                   running_var);
        } else {
            hipLaunchKernelGGL(pcm_bnact_reduce_kernel, dim3((2 * C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, 2 * C, partial, sums);
            if (use_given_stat == 2) return PCM_LAUNCH_STATUS();
            if (is_bf16)
                hipLaunchKernelGGL(pcm_bnact_stats_kernel<bf>, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, s, C, (double)n, eps,
                                   momentum, (const bf *)y, sums, gamma, beta, stat, running_mean, running_var);
            else
                hipLaunchKernelGGL(pcm_bnact_stats_kernel<float>, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, s, C, (double)n, eps,
                                   momentum, (const float *)y, sums, gamma, beta, stat, running_mean, running_var);
        }
    }
    const long total4 = n * C / 4;
    if (is_bf16)
        hipLaunchKernelGGL(pcm_bnact_apply_kernel<bf>, dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, (const bf *)y, stat, (bf *)z);
    else
        hipLaunchKernelGGL(pcm_bnact_apply_kernel<float>, dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, (const float *)y, stat,
                           (float *)z);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_bn_act_backward_hip(long n, int C, int is_bf16, int relu, const void *y, const void *dz, const float *stat,
                                       float *partial, float *sums, void *dy, int phase, double count, void *stream)
{
    // phase: 0 = whole backward; 1 = local sums only (sums = {sum g, sum g * xhat}); 2 = apply only with the given sums
    // (all-reduced across ranks by the caller) and `count` = rows of the GLOBAL batch (<= 0: n)
    if (relu) return pcm_bn_relu_backward_hip(n, C, is_bf16, y, dz, stat, partial, sums, dy, phase, count, stream);
    if (n == 0) return PCM_OK;
    if (!pcm_bn_relu_supported(n, C)) return PCM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    using bf = __hip_bfloat16;
    const Plan p = plan_for(n, C);
    const dim3 grid(p.nslots, p.nchunk);
    const long total4 = n * C / 4;
    const float inv_n = (float)(1.0 / (count > 0.0 ? count : (double)n));
    if (is_bf16) {
        if (phase != 2) {
            hipLaunchKernelGGL((pcm_bnact_colsum_kernel<bf, 2>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot, (const bf *)y,
                               (const bf *)dz, stat, partial);
            hipLaunchKernelGGL(pcm_bnact_reduce_kernel, dim3((2 * C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, 2 * C, partial, sums);
        }
        if (phase != 1)
            hipLaunchKernelGGL(pcm_bnact_bwd_apply_kernel<bf>, dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, inv_n, (const bf *)y,
                               (const bf *)dz, stat, sums, (bf *)dy);
    } else {
        if (phase != 2) {
            hipLaunchKernelGGL((pcm_bnact_colsum_kernel<float, 2>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot,
                               (const float *)y, (const float *)dz, stat, partial);
            hipLaunchKernelGGL(pcm_bnact_reduce_kernel, dim3((2 * C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, 2 * C, partial, sums);
        }
        if (phase != 1)
            hipLaunchKernelGGL(pcm_bnact_bwd_apply_kernel<float>, dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, inv_n,
                               (const float *)y, (const float *)dz, stat, sums, (float *)dy);
    }
    return PCM_LAUNCH_STATUS();
}
