// bnrelu.hip -- fused BatchNorm1d (batch statistics) + ReLU over per-point features (n, C) for gfx950.  HBM bound.
//
// Every PointNet layer of the reference is  conv(k=1, no bias) -> BatchNorm1d(eps 1e-3, momentum 0.01) -> ReLU  on the
// packed (n, C) point features (/root/reference/src/models/components/pcd_encoder/pointnet.py:25-56, C = 64, 64, 64,
// 128, 512; n = all points of the batch, up to 131072 for the Diffusion-Policy workloads).  The framework serves the
// BatchNorm with its channels-last kernels at ~0.6 TB/s (rocprofv3: 2.8 ms of a 15 ms step at n = 131072) plus a
// separate ReLU each way.  Here, per layer:
//
//   forward : colsum (sum (y-y0), sum (y-y0)^2 per column, y0 = row 0: no cancellation when |mean| >> std; per-slot fp32
//             partials) -> fp64 reduce -> stats (mean, invstd,
//             a = gamma*invstd, b = beta - a*mean, running-stat update) -> apply  z = max(a*y + b, 0)
//   backward: colsum (sum g, sum g*xhat with g = dz * [a*y+b > 0], xhat = (y-mean)*invstd) -> reduce ->
//             apply  dy = a * (g - mean(g) - xhat * mean(g*xhat));   dbeta = sum g, dgamma = sum g*xhat
//
// y is read twice each way and never re-materialised (the ReLU mask is recomputed from y); no atomics, fixed
// reduction order (deterministic).  y / z / dz / dy are bf16 under autocast (like torch's batch_norm, which keeps the
// input dtype), statistics are fp32 with an fp64 cross-slot reduction.
// Algorithmic bytes per element of y (bf16): forward 2+2 read + 2 written; backward (2+2)*2 read + 2 written.
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

// MODE 0: (y, y^2)        MODE 1: (g, g*xhat), g = dz behind the ReLU gate        MODE 2: the same without a gate (BatchNorm alone)
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void pcm_bn_colsum_kernel(long n, int C, int chunkW, long rows_per_slot,
                                                               const T *__restrict__ y, const T *__restrict__ dz,
                                                               const float *__restrict__ stat, float *__restrict__ partial)
{
    __shared__ float lds[8 * kBlock];
    asm volatile("" ::"s"(n), "s"(C), "s"(chunkW), "s"(rows_per_slot), "s"(y), "s"(dz), "s"(stat), "s"(partial));  // "Kernel heads", pcm_common.hpp
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
        auto accumulate = [&](const float(&v)[4], const float(&d)[4]) {
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dd = v[u] - sh[u];
                    s0[u] += dd;
                    s1[u] += dd * dd;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float g;
                    if constexpr (MODE == 2) g = d[u];
                    else g = (a[u] * v[u] + b[u] > 0.f) ? d[u] : 0.f;
                    s0[u] += g;
                    s1[u] += g * ((v[u] - mean[u]) * invstd[u]);
                }
            }
        };
        // four rows in flight, accumulated in row order: same sums as the one-row loop, which waited for every row's loads in turn
        // (tools/isa_load_chains.py: >= 8 exposed round trips in series per thread)
        long r = r_begin + mp.rsub;
        for (; r + 3 * mp.rpp < r_end; r += 4 * (long)mp.rpp) {
            float v[4][4], d[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                load4<T>(y + (r + q * (long)mp.rpp) * C + c0, v[q]);
                if (MODE != 0) load4<T>(dz + (r + q * (long)mp.rpp) * C + c0, d[q]);
                else d[q][0] = d[q][1] = d[q][2] = d[q][3] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) accumulate(v[q], d[q]);
        }
        for (; r < r_end; r += mp.rpp) {
            float v[4], d[4] = {0.f, 0.f, 0.f, 0.f};
            load4<T>(y + r * C + c0, v);
            if (MODE != 0) load4<T>(dz + r * C + c0, d);
            accumulate(v, d);
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
__global__ __launch_bounds__(64 * kRedWaves) void pcm_bn_reduce_kernel(int nslots, int VH, const float *__restrict__ partial,
                                                                         float *__restrict__ out)
{
    __shared__ double red[kRedWaves][64];
    asm volatile("" ::"s"(nslots), "s"(VH), "s"(partial), "s"(out));  // "Kernel heads", pcm_common.hpp
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
    const double dm = (double)s1 / count;  // mean of (y - shift), shift = row 0 of y (see pcm_bn_colsum_kernel)
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

// pcm_bn_reduce_kernel + pcm_bn_stats_kernel in one launch (single-rank training): a workgroup reduces both moments of 64
// channels -- the same chains and order as pcm_bn_reduce_kernel, so the same bits -- and its first wave finishes the
// statistics.  One launch less per BatchNorm layer (6 per ACT step, each on the critical path of the tokenizer).
template <typename T>
__global__ __launch_bounds__(64 * kRedWaves) void pcm_bn_reduce_stats_kernel(int nslots, int C, const float *__restrict__ partial,
                                                                               float *__restrict__ sums, double count, float eps,
                                                                               float momentum, const T *__restrict__ y,
                                                                               const float *__restrict__ gamma,
                                                                               const float *__restrict__ beta, float *__restrict__ stat,
                                                                               float *__restrict__ running_mean,
                                                                               float *__restrict__ running_var)
{
    __shared__ double red[2][kRedWaves][64];
    asm volatile("" ::"s"(nslots), "s"(C), "s"(partial), "s"(sums), "s"(count), "s"(eps), "s"(momentum), "s"(y), "s"(gamma), "s"(beta), "s"(stat), "s"(running_mean), "s"(running_var));  // "Kernel heads", pcm_common.hpp
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
__global__ __launch_bounds__(kBlock) void pcm_bn_stats_kernel(int C, double count, float eps, float momentum, const T *__restrict__ y,
                                                              const float *__restrict__ sums, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *__restrict__ stat,
                                                              float *__restrict__ running_mean, float *__restrict__ running_var)
{
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    bn_stats_of<T>(c, C, count, eps, momentum, y, sums[c], sums[C + c], gamma, beta, stat, running_mean, running_var);
}

template <typename T, bool RELU = true>
__global__ __launch_bounds__(kBlock) void pcm_bn_relu_apply_kernel(long total4, int C, const T *__restrict__ y,
                                                                   const float *__restrict__ stat, T *__restrict__ z)
{
    asm volatile("" ::"s"(gridDim.x), "s"(total4), "s"(C), "s"(y), "s"(stat), "s"(z));  // "Kernel heads", pcm_common.hpp
    const unsigned c4n = (unsigned)C / 4u;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total4; i += (long)gridDim.x * kBlock) {
        const long e = i * 4;
        // channel of element e = 4 i: a 32-bit remainder whenever the tensor has fewer than 2^32 four-element pieces (always, here)
        const int c = total4 <= 0xFFFFFFFFl ? (int)(((unsigned)i % c4n) * 4u) : (int)(e % C);
        float v[4], a[4], b[4], o[4];
        load4<T>(y + e, v);
        load4<float>(stat + 2 * C + c, a);
        load4<float>(stat + 3 * C + c, b);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t = a[u] * v[u] + b[u];
            if constexpr (RELU) o[u] = t > 0.f ? t : 0.f;
            else o[u] = t;
        }
        store4<T>(z + e, o);
    }
}

template <typename T, bool RELU = true>
__global__ __launch_bounds__(kBlock) void pcm_bn_relu_bwd_apply_kernel(long total4, int C, float inv_n, const T *__restrict__ y,
                                                                       const T *__restrict__ dz, const float *__restrict__ stat,
                                                                       const float *__restrict__ sums, T *__restrict__ dy)
{
    asm volatile("" ::"s"(gridDim.x), "s"(total4), "s"(C), "s"(inv_n), "s"(y), "s"(dz), "s"(stat), "s"(sums), "s"(dy));  // "Kernel heads", pcm_common.hpp
    const unsigned c4n = (unsigned)C / 4u;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total4; i += (long)gridDim.x * kBlock) {
        const long e = i * 4;
        const int c = total4 <= 0xFFFFFFFFl ? (int)(((unsigned)i % c4n) * 4u) : (int)(e % C);
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
            float g;
            if constexpr (RELU) g = (a[u] * v[u] + b[u] > 0.f) ? d[u] : 0.f;
            else g = d[u];
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

namespace {
// ---- synchronised BatchNorm (policy/sync_bn.py): the statistics exchange around the collective as two launches --------------
// pack: the rank's per-channel mean, sum of squared deviations and row count from the kernels' shifted sums
//   d = S1 / n; mean = shift + d; M2 = S2 - S1 * d        (n == 0: an empty rank contributes zeros with count 0)
//   shift = row `*row_index` (row 0 when the pointer is NULL; zeros when the index is negative) of the (rows, C) matrix `src`
//   (fp32 or bf16) that the forward kernel accumulated around: read here, not by a chain of framework indexing launches
__global__ __launch_bounds__(256) void pcm_bn_sync_pack_kernel(int C, double n, const float *__restrict__ sums, const void *__restrict__ src,
                                                               int src_is_bf16, const int *__restrict__ row_index, float *__restrict__ pack)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0) pack[2 * C] = (float)n;
    if (c >= C) return;
    if (n <= 0.0) {
        pack[c] = 0.f, pack[C + c] = 0.f;
        return;
    }
    const long row = row_index != nullptr ? (long)row_index[0] : 0;
    float sh = 0.f;
    if (row >= 0) {
        if (src_is_bf16) sh = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(src)[row * C + c] << 16);
        else sh = reinterpret_cast<const float *>(src)[row * C + c];
    }
    const float s1 = sums[c], s2 = sums[C + c];
    const float d = s1 / (float)n;
    pack[c] = sh + d;
    pack[C + c] = s2 - s1 * d;
}

// combine (Chan et al.): all ranks' (mean, M2, count) -> stat (4, C) = { mean, invstd, a = gamma invstd, b = beta - a mean } of the
// GLOBAL batch in fp64, the running-statistics update, and ratio = n_loc / N.  Every rank runs the same arithmetic on the same
// gathered numbers in rank order: identical results everywhere.
__global__ __launch_bounds__(256) void pcm_bn_sync_combine_kernel(int W, int C, const float *__restrict__ all, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float eps, float momentum,
                                                                  float *__restrict__ running_mean, float *__restrict__ running_var,
                                                                  double n_loc, float *__restrict__ stat, float *__restrict__ ratio)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int P = 2 * C + 1;
    double n = 0.0;
    for (int r = 0; r < W; ++r) n += (double)all[(size_t)r * P + 2 * C];
    if (c == 0) ratio[0] = (float)(n_loc / n);
    if (c >= C) return;
    double mean = 0.0;
    for (int r = 0; r < W; ++r) mean += (double)all[(size_t)r * P + c] * (double)all[(size_t)r * P + 2 * C];
    mean /= n;
    double m2 = 0.0;
    for (int r = 0; r < W; ++r) {
        const double dm = (double)all[(size_t)r * P + c] - mean;
        m2 += (double)all[(size_t)r * P + C + c] + (double)all[(size_t)r * P + 2 * C] * dm * dm;
    }
    double var = m2 / n;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double a = (double)gamma[c] * invstd;
    stat[c] = (float)mean;
    stat[C + c] = (float)invstd;
    stat[2 * C + c] = (float)a;
    stat[3 * C + c] = (float)((double)beta[c] - a * mean);
    if (running_mean != nullptr) {
        const double unbiased = var * (n / (n - 1.0 > 1.0 ? n - 1.0 : 1.0));
        running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * (float)mean;
        running_var[c] = running_var[c] * (1.f - momentum) + momentum * (float)unbiased;
    }
}
}  // namespace

extern "C" int pcm_bn_sync_pack_hip(int C, double count, const float *sums, const void *src, int src_is_bf16, const int *row_index,
                                    float *pack, void *stream)
{
    if (C <= 0 || count < 0.0 || !pack || (count > 0.0 && (!sums || !src))) return PCM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pcm_bn_sync_pack_kernel, dim3((C + 256) / 256), dim3(256), 0, (hipStream_t)stream, C, count, sums, src, src_is_bf16,
                       row_index, pack);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_bn_sync_combine_hip(int W, int C, const float *gathered, const float *gamma, const float *beta, float eps, float momentum,
                                       float *running_mean, float *running_var, double count_local, float *stat, float *ratio,
                                       void *stream)
{
    if (W <= 0 || C <= 0 || !gathered || !gamma || !beta || !stat || !ratio || (running_mean == nullptr) != (running_var == nullptr))
        return PCM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pcm_bn_sync_combine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, C, gathered, gamma, beta, eps,
                       momentum, running_mean, running_var, count_local, stat, ratio);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_bn_relu_supported(long n, int C)
{
    return (n > 0 && C > 0 && C % 4 == 0 && (C <= kMaxChunk || C % kMaxChunk == 0)) ? 1 : 0;
}

extern "C" int pcm_bn_relu_slots(long n, int C)
{
    if (!pcm_bn_relu_supported(n, C)) return 0;
    return plan_for(n, C).nslots;
}

// BatchNorm1d over rows with (relu != 0) or without the ReLU behind it: the Diffusion Policy's projector ends with a bare BatchNorm
// (/root/reference/src/models/components/diffusion_policy/vision/pcd_obs_encoder.py:100-120)
extern "C" int pcm_bn_act_forward_hip(long n, int C, int is_bf16, int relu, const void *y, const float *gamma, const float *beta,
                                      float eps, float momentum, float *running_mean, float *running_var, int use_given_stat,
                                      float *partial, float *sums, float *stat, void *z, void *stream)
{
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
            hipLaunchKernelGGL((pcm_bn_colsum_kernel<bf, 0>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot,
                               (const bf *)y, (const bf *)nullptr, (const float *)nullptr, partial);
        else
            hipLaunchKernelGGL((pcm_bn_colsum_kernel<float, 0>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot,
                               (const float *)y, (const float *)nullptr, (const float *)nullptr, partial);
        if (use_given_stat == 0) {  // reduce + statistics in one launch
            if (is_bf16)
                hipLaunchKernelGGL(pcm_bn_reduce_stats_kernel<bf>, dim3((C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, C, partial,
                                   sums, (double)n, eps, momentum, (const bf *)y, gamma, beta, stat, running_mean, running_var);
            else
                hipLaunchKernelGGL(pcm_bn_reduce_stats_kernel<float>, dim3((C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, C,
                                   partial, sums, (double)n, eps, momentum, (const float *)y, gamma, beta, stat, running_mean,
                                   running_var);
        } else {
        hipLaunchKernelGGL(pcm_bn_reduce_kernel, dim3((2 * C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, 2 * C, partial, sums);
        if (use_given_stat == 2) return PCM_LAUNCH_STATUS();
        if (is_bf16)
            hipLaunchKernelGGL(pcm_bn_stats_kernel<bf>, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, s, C, (double)n, eps, momentum,
                               (const bf *)y, sums, gamma, beta, stat, running_mean, running_var);
        else
            hipLaunchKernelGGL(pcm_bn_stats_kernel<float>, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, s, C, (double)n, eps,
                               momentum, (const float *)y, sums, gamma, beta, stat, running_mean, running_var);
        }
    }
    const long total4 = n * C / 4;
    if (is_bf16 && relu)
        hipLaunchKernelGGL(pcm_bn_relu_apply_kernel<bf>, dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, (const bf *)y, stat, (bf *)z);
    else if (is_bf16)
        hipLaunchKernelGGL((pcm_bn_relu_apply_kernel<bf, false>), dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, (const bf *)y, stat,
                           (bf *)z);
    else if (relu)
        hipLaunchKernelGGL(pcm_bn_relu_apply_kernel<float>, dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, (const float *)y, stat,
                           (float *)z);
    else
        hipLaunchKernelGGL((pcm_bn_relu_apply_kernel<float, false>), dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, (const float *)y,
                           stat, (float *)z);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_bn_relu_forward_hip(long n, int C, int is_bf16, const void *y, const float *gamma, const float *beta,
                                       float eps, float momentum, float *running_mean, float *running_var,
                                       int use_given_stat, float *partial, float *sums, float *stat, void *z, void *stream)
{
    return pcm_bn_act_forward_hip(n, C, is_bf16, 1, y, gamma, beta, eps, momentum, running_mean, running_var, use_given_stat, partial, sums,
                                  stat, z, stream);
}


extern "C" int pcm_bn_act_backward_hip(long n, int C, int is_bf16, int relu, const void *y, const void *dz, const float *stat,
                                       float *partial, float *sums, void *dy, int phase, double count, void *stream)
{
    // phase: 0 = whole backward; 1 = local sums only (sums = {sum g, sum g * xhat}); 2 = apply only with the given sums
    // (all-reduced across ranks by the caller) and `count` = rows of the GLOBAL batch (<= 0: n)
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
            if (relu)
                hipLaunchKernelGGL((pcm_bn_colsum_kernel<bf, 1>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot, (const bf *)y,
                                   (const bf *)dz, stat, partial);
            else
                hipLaunchKernelGGL((pcm_bn_colsum_kernel<bf, 2>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot, (const bf *)y,
                                   (const bf *)dz, stat, partial);
            hipLaunchKernelGGL(pcm_bn_reduce_kernel, dim3((2 * C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, 2 * C, partial, sums);
        }
        if (phase != 1 && relu)
            hipLaunchKernelGGL(pcm_bn_relu_bwd_apply_kernel<bf>, dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, inv_n, (const bf *)y,
                               (const bf *)dz, stat, sums, (bf *)dy);
        else if (phase != 1)
            hipLaunchKernelGGL((pcm_bn_relu_bwd_apply_kernel<bf, false>), dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, inv_n,
                               (const bf *)y, (const bf *)dz, stat, sums, (bf *)dy);
    } else {
        if (phase != 2) {
            if (relu)
                hipLaunchKernelGGL((pcm_bn_colsum_kernel<float, 1>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot,
                                   (const float *)y, (const float *)dz, stat, partial);
            else
                hipLaunchKernelGGL((pcm_bn_colsum_kernel<float, 2>), grid, dim3(kBlock), 0, s, n, C, p.chunkW, p.rows_per_slot,
                                   (const float *)y, (const float *)dz, stat, partial);
            hipLaunchKernelGGL(pcm_bn_reduce_kernel, dim3((2 * C + 63) / 64), dim3(64 * kRedWaves), 0, s, p.nslots, 2 * C, partial, sums);
        }
        if (phase != 1 && relu)
            hipLaunchKernelGGL(pcm_bn_relu_bwd_apply_kernel<float>, dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, inv_n,
                               (const float *)y, (const float *)dz, stat, sums, (float *)dy);
        else if (phase != 1)
            hipLaunchKernelGGL((pcm_bn_relu_bwd_apply_kernel<float, false>), dim3(ew_grid(total4)), dim3(kBlock), 0, s, total4, C, inv_n,
                               (const float *)y, (const float *)dz, stat, sums, (float *)dy);
    }
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_bn_relu_backward_hip(long n, int C, int is_bf16, const void *y, const void *dz, const float *stat,
                                        float *partial, float *sums, void *dy, int phase, double count, void *stream)
{
    return pcm_bn_act_backward_hip(n, C, is_bf16, 1, y, dz, stat, partial, sums, dy, phase, count, stream);
}
