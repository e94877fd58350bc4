// gnmish.hip -- channels-last building blocks of the Diffusion-Policy 1-D U-Net for gfx950 (MI355X).  HBM bound.
//
// The reference's Conv1dBlock is  Conv1d -> GroupNorm(8) -> Mish  on (B, C, T) tensors with T = 16 / 8 / 4 and
// C = 512 ... 4096 (/root/reference/src/models/components/diffusion_policy/diffusion/conv1d_components.py:25-45),
// optionally followed by FiLM  out = scale * out + bias  and, for the second block of a residual block, by the
// residual add (conditional_unet1d.py:56-75).  Through the framework that is pad + unfold + copy + cast for the
// convolution input, transpose-copy + 3 GroupNorm launches + Mish + 2 FiLM launches after it, and twice that
// backward.  Here activations stay channels-last (B, T, C) -- exactly what the im2col GEMM consumes and produces --
// and the chain is TWO kernels per block each way:
//
//   pcm_im2col_cl : x (B, T, C) -> cols (B*L_out, C*K), column c*K + k = x[b, l*stride + k - pad, c] (0 outside);
//                   the column order matches nn.Conv1d's weight (C_out, C_in, K) viewed as (C_out, C_in*K);
//                   output bf16 under autocast (the cast is fused).  pcm_col2im_cl is its adjoint.
//   pcm_gn_mish   : y = [scale *] mish(GroupNorm(x)) [+ bias] [+ res]  with per-(b, group) statistics over T x C/G
//                   elements; one workgroup per (b, group) keeps the group in LDS (one HBM read, one write).
//                   Backward recomputes mish from x, returns dx, per-sample partial (dgamma, dbeta) and the FiLM
//                   gradients; deterministic (no atomics).
//
// Algorithmic bytes per element (fp32): gn_mish forward 4 (x) + 4 (y) [+ 4 res]; backward 4 (dy) + 4 (x) + 4 (dx).
// im2col: C*T*4 read + C*K*L_out*{2,4} written per sample.
#include "pcm_elem.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGroupElems = 6656;  // T * C/G floats staged in LDS (x2 + accumulators + 2 per channel in backward < 64 KiB)

template <typename T>
__device__ __forceinline__ float ldf(const T *p);
template <>
__device__ __forceinline__ float ldf<float>(const float *p)
{
    return *p;
}
template <>
__device__ __forceinline__ float ldf<__hip_bfloat16>(const __hip_bfloat16 *p)
{
    return __bfloat162float(*p);
}
template <typename T>
__device__ __forceinline__ void stf(T *p, float v);
template <>
__device__ __forceinline__ void stf<float>(float *p, float v)
{
    *p = v;
}
template <>
__device__ __forceinline__ void stf<__hip_bfloat16>(__hip_bfloat16 *p, float v)
{
    *p = __float2bfloat16(v);
}

// sum over the 256 threads of the block; `red` is 4 floats of LDS; every thread gets the total
__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float mish_f(float z)
{
    return z * tanhf(log1pf(expf(z)));
}

// thread -> (channel j, first row t0) mapping shared by forward and backward: W = min(cg, 256) threads per row,
// R = 256 / W rows in flight; threads >= W*R idle (they still take part in the barriers)
struct GroupMap {
    int W, R, j0, t0;
    bool active;
    __device__ GroupMap(int cg)
    {
        W = cg < kBlock ? cg : kBlock;
        R = kBlock / W;
        active = (int)threadIdx.x < W * R;
        j0 = threadIdx.x % W;
        t0 = threadIdx.x / W;
    }
};

// film_mode: 0 none, 1 scale & bias (film = (B, 2, C)), 2 bias only (film = (B, C))
template <typename TX, typename TF, typename TR>
__global__ __launch_bounds__(kBlock) void pcm_gn_mish_fwd_kernel(int T, int C, int G, const TX *__restrict__ x,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 float eps, int film_mode, const TF *__restrict__ film,
                                                                 const TR *__restrict__ res, const float *__restrict__ cbias,
                                                                 float *__restrict__ y, float *__restrict__ mean_out,
                                                                 float *__restrict__ rstd_out)
{
    extern __shared__ float lds[];  // [T*cg] group values, then 4 floats for reductions
    const int cg = C / G, n = T * cg;
    float *vals = lds, *red = lds + n;
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const GroupMap mp(cg);
    const long base = (long)b * T * C + (long)g * cg;
    float s = 0.f;
    if (mp.active)
        for (int j = mp.j0; j < cg; j += mp.W) {
            const float cb = cbias ? cbias[g * cg + j] : 0.f;  // the producing convolution's bias, folded in here
            for (int t = mp.t0; t < T; t += mp.R) {
                const float v = ldf<TX>(x + base + (long)t * C + j) + cb;
                vals[t * cg + j] = v;  // read back only by this thread
                s += v;
            }
        }
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    if (mp.active)
        for (int j = mp.j0; j < cg; j += mp.W)
            for (int t = mp.t0; t < T; t += mp.R) {
                const float d = vals[t * cg + j] - mean;
                q += d * d;
            }
    const float var = block_sum(q, red) / (float)n;
    const float rstd = 1.f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
        mean_out[blockIdx.x] = mean;
        rstd_out[blockIdx.x] = rstd;
    }
    if (!mp.active) return;
    for (int j = mp.j0; j < cg; j += mp.W) {
        const int c = g * cg + j;
        const float ga = gamma[c], be = beta[c];
        float sc = 1.f, bi = 0.f;
        if (film_mode == 1) {
            sc = ldf<TF>(film + (long)b * 2 * C + c);
            bi = ldf<TF>(film + (long)b * 2 * C + C + c);
        } else if (film_mode == 2) {
            bi = ldf<TF>(film + (long)b * C + c);
        }
        for (int t = mp.t0; t < T; t += mp.R) {
            const float z = (vals[t * cg + j] - mean) * rstd * ga + be;
            float o = mish_f(z);
            if (film_mode == 1) o = sc * o + bi;
            else if (film_mode == 2) o = o + bi;
            const long e = base + (long)t * C + j;
            if (res != nullptr) o = o + ldf<TR>(res + e);
            y[e] = o;
        }
    }
}

template <typename TX, typename TF>
__global__ __launch_bounds__(kBlock) void pcm_gn_mish_bwd_kernel(int T, int C, int G, const TX *__restrict__ x,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                                 int film_mode, const TF *__restrict__ film,
                                                                 const float *__restrict__ cbias, const float *__restrict__ dy,
                                                                 TX *__restrict__ dx, float *__restrict__ dgb_partial,
                                                                 float *__restrict__ dfilm)
{
    extern __shared__ float lds[];  // xhat[n] | dxhat[n] | acc[5][256] | red[4] | chan[2][cg]
    const int cg = C / G, n = T * cg;
    float *xh = lds, *dxh = lds + n, *acc = lds + 2 * n, *red = acc + 5 * kBlock, *chan = red + 4;
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const GroupMap mp(cg);
    const long base = (long)b * T * C + (long)g * cg;
    const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
    float s1 = 0.f, s2 = 0.f;
    const int jiters = (cg + mp.W - 1) / mp.W;  // uniform across the block (barriers inside)
    for (int it = 0; it < jiters; ++it) {
        const int j = mp.j0 + it * mp.W;
        const bool live = mp.active && j < cg;
        float a_dg = 0.f, a_db = 0.f, a_ds = 0.f, a_dbi = 0.f, a_xh = 0.f;
        if (live) {
            const int c = g * cg + j;
            const float ga = gamma[c], be = beta[c];
            const float sc = film_mode == 1 ? ldf<TF>(film + (long)b * 2 * C + c) : 1.f;
            const float cb = cbias ? cbias[c] : 0.f;
            for (int t = mp.t0; t < T; t += mp.R) {
                const long e = base + (long)t * C + j;
                const float xhat = (ldf<TX>(x + e) + cb - mean) * rstd;
                const float z = xhat * ga + be;
                const float ts = tanhf(log1pf(expf(z)));
                const float sig = 1.f / (1.f + expf(-z));
                const float dmdz = ts + z * sig * (1.f - ts * ts);
                const float g_out = dy[e];
                a_ds += g_out * (z * ts);
                a_dbi += g_out;
                const float dz = (g_out * sc) * dmdz;
                a_dg += dz * xhat;
                a_db += dz;
                a_xh += xhat;
                const float d = dz * ga;
                xh[t * cg + j] = xhat;
                dxh[t * cg + j] = d;
                s1 += d;
                s2 += d * xhat;
            }
        }
        // fold the R row-slots of each channel in a fixed order
        __syncthreads();
        acc[0 * kBlock + threadIdx.x] = a_dg;
        acc[1 * kBlock + threadIdx.x] = a_db;
        acc[2 * kBlock + threadIdx.x] = a_ds;
        acc[3 * kBlock + threadIdx.x] = a_dbi;
        acc[4 * kBlock + threadIdx.x] = a_xh;
        __syncthreads();
        if (live && mp.t0 == 0) {
            float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int rr = 0; rr < mp.R; ++rr)
#pragma unroll
                for (int k = 0; k < 5; ++k) r[k] += acc[k * kBlock + rr * mp.W + mp.j0];
            const int c = g * cg + j;
            chan[j] = gamma[c] * r[1];  // sum_t dxhat for this channel
            chan[cg + j] = r[4];        // sum_t xhat
            dgb_partial[(long)b * 3 * C + c] = r[0];
            dgb_partial[(long)b * 3 * C + C + c] = r[1];
            if (film_mode == 1) {
                dfilm[(long)b * 2 * C + c] = r[2];
                dfilm[(long)b * 2 * C + C + c] = r[3];
            } else if (film_mode == 2) {
                dfilm[(long)b * C + c] = r[3];
            }
        }
    }
    const float c1 = block_sum(s1, red) / (float)n;
    const float c2 = block_sum(s2, red) / (float)n;
    // sum_t dx of each channel = the gradient of a bias added in front of the normalisation (the convolution's bias)
    for (int j = threadIdx.x; j < cg; j += kBlock)
        dgb_partial[(long)b * 3 * C + 2 * C + g * cg + j] = rstd * (chan[j] - (float)T * c1 - c2 * chan[cg + j]);
    if (!mp.active) return;
    for (int j = mp.j0; j < cg; j += mp.W)
        for (int t = mp.t0; t < T; t += mp.R) {
            const float v = rstd * (dxh[t * cg + j] - c1 - xh[t * cg + j] * c2);
            stf<TX>(dx + base + (long)t * C + j, v);
        }
}

// ---- im2col / col2im, channels-last -------------------------------------------------------------------------
template <typename TI, typename TO, int VEC>
__global__ __launch_bounds__(kBlock) void pcm_im2col_cl_kernel(long total_v, int T, int C, int K, int stride, int pad, int Lout,
                                                               const TI *__restrict__ x, TO *__restrict__ cols)
{
    const int CK = C * K;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total_v; i += (long)gridDim.x * kBlock) {
        const long e0 = i * VEC;
        const long row = e0 / CK;
        int col = (int)(e0 - row * CK);
        const int b = (int)(row / Lout), l = (int)(row - (long)b * Lout);
        int c = col / K, k = col - c * K;
        float v[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            const int t = l * stride + k - pad;
            v[u] = (t >= 0 && t < T) ? ldf<TI>(x + ((long)b * T + t) * C + c) : 0.f;
            if (++k == K) k = 0, ++c;
        }
        if constexpr (VEC == 4) {
            store4<TO>(cols + e0, v);
        } else {
            stf<TO>(cols + e0, v[0]);
        }
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(kBlock) void pcm_col2im_cl_kernel(long total, int T, int C, int K, int stride, int pad, int Lout,
                                                               const TI *__restrict__ dcols, TO *__restrict__ dx)
{
    const long CK = (long)C * K;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int c = (int)(i % C);
        const long bt = i / C;
        const int t = (int)(bt % T);
        const long b = bt / T;
        float a = 0.f;
        for (int k = 0; k < K; ++k) {
            const int num = t + pad - k;
            if (num < 0) break;
            const int l = num / stride;
            if (l * stride != num || l >= Lout) continue;
            a += ldf<TI>(dcols + (b * Lout + l) * CK + (long)c * K + k);
        }
        stf<TO>(dx + i, a);
    }
}

inline int grid_for(long work)
{
    long blocks = (work + kBlock - 1) / kBlock;
    if (blocks > 256L * 16) blocks = 256L * 16;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int pcm_gn_mish_supported(int T, int C, int G)
{
    if (T <= 0 || C <= 0 || G <= 0 || C % G != 0) return 0;
    const long n = (long)T * (C / G);
    return (n <= kMaxGroupElems && C / G <= 4 * kBlock) ? 1 : 0;
}

extern "C" int pcm_gn_mish_forward_hip(int B, int T, int C, int G, int x_is_bf16, const void *x, const float *gamma,
                                       const float *beta, float eps, int film_mode, int film_is_bf16, const void *film,
                                       int res_is_bf16, const void *res, const float *conv_bias, float *y, float *mean, float *rstd,
                                       void *stream)
{
    if (B < 0) return PCM_ERR_BAD_ARG;
    if (B == 0) return PCM_OK;
    if (film_mode < 0 || film_mode > 2 || (film_mode && !film)) return PCM_ERR_BAD_ARG;
    if (!pcm_gn_mish_supported(T, C, G)) return PCM_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)T * (C / G) + 4) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    using bf = __hip_bfloat16;
#define PCM_GN_FWD(TX, TF, TR)                                                                                               \
    hipLaunchKernelGGL((pcm_gn_mish_fwd_kernel<TX, TF, TR>), dim3(B * G), dim3(kBlock), lds, s, T, C, G, (const TX *)x, gamma, \
                       beta, eps, film_mode, (const TF *)film, (const TR *)res, conv_bias, y, mean, rstd)
    const int key = (x_is_bf16 ? 4 : 0) | (film_is_bf16 ? 2 : 0) | (res_is_bf16 ? 1 : 0);
    switch (key) {
    case 0: PCM_GN_FWD(float, float, float); break;
    case 1: PCM_GN_FWD(float, float, bf); break;
    case 2: PCM_GN_FWD(float, bf, float); break;
    case 3: PCM_GN_FWD(float, bf, bf); break;
    case 4: PCM_GN_FWD(bf, float, float); break;
    case 5: PCM_GN_FWD(bf, float, bf); break;
    case 6: PCM_GN_FWD(bf, bf, float); break;
    default: PCM_GN_FWD(bf, bf, bf); break;
    }
#undef PCM_GN_FWD
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_gn_mish_backward_hip(int B, int T, int C, int G, int x_is_bf16, const void *x, const float *gamma,
                                        const float *beta, const float *mean, const float *rstd, int film_mode,
                                        int film_is_bf16, const void *film, const float *conv_bias, const float *dy, void *dx,
                                        float *dgb_partial, float *dfilm, void *stream)
{
    if (B < 0) return PCM_ERR_BAD_ARG;
    if (B == 0) return PCM_OK;
    if (film_mode < 0 || film_mode > 2 || (film_mode && (!film || !dfilm))) return PCM_ERR_BAD_ARG;
    if (!pcm_gn_mish_supported(T, C, G)) return PCM_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)2 * T * (C / G) + 5 * kBlock + 4 + 2 * (C / G)) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    using bf = __hip_bfloat16;
#define PCM_GN_BWD(TX, TF)                                                                                                   \
    hipLaunchKernelGGL((pcm_gn_mish_bwd_kernel<TX, TF>), dim3(B * G), dim3(kBlock), lds, s, T, C, G, (const TX *)x, gamma,   \
                       beta, mean, rstd, film_mode, (const TF *)film, conv_bias, dy, (TX *)dx, dgb_partial, dfilm)
    if (x_is_bf16) {
        if (film_is_bf16) PCM_GN_BWD(bf, bf); else PCM_GN_BWD(bf, float);
    } else {
        if (film_is_bf16) PCM_GN_BWD(float, bf); else PCM_GN_BWD(float, float);
    }
#undef PCM_GN_BWD
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_im2col_cl_hip(int B, int T, int C, int K, int stride, int pad, int x_is_bf16, const void *x,
                                 int out_is_bf16, void *cols, void *stream)
{
    if (B < 0 || T <= 0 || C <= 0 || K <= 0 || stride <= 0 || pad < 0) return PCM_ERR_BAD_ARG;
    const int Lout = (T + 2 * pad - K) / stride + 1;
    if (Lout <= 0) return PCM_ERR_BAD_ARG;
    const long total = (long)B * Lout * C * K;
    if (total == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    using bf = __hip_bfloat16;
    const bool v4 = ((long)C * K) % 4 == 0;
#define PCM_I2C(TI, TO)                                                                                                      \
    do {                                                                                                                     \
        if (v4)                                                                                                              \
            hipLaunchKernelGGL((pcm_im2col_cl_kernel<TI, TO, 4>), dim3(grid_for(total / 4)), dim3(kBlock), 0, s, total / 4, T, C, \
                               K, stride, pad, Lout, (const TI *)x, (TO *)cols);                                             \
        else                                                                                                                 \
            hipLaunchKernelGGL((pcm_im2col_cl_kernel<TI, TO, 1>), dim3(grid_for(total)), dim3(kBlock), 0, s, total, T, C, K,  \
                               stride, pad, Lout, (const TI *)x, (TO *)cols);                                                \
    } while (0)
    if (x_is_bf16) {
        if (out_is_bf16) PCM_I2C(bf, bf); else PCM_I2C(bf, float);
    } else {
        if (out_is_bf16) PCM_I2C(float, bf); else PCM_I2C(float, float);
    }
#undef PCM_I2C
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_col2im_cl_hip(int B, int T, int C, int K, int stride, int pad, int cols_is_bf16, const void *dcols,
                                 int dx_is_bf16, void *dx, void *stream)
{
    if (B < 0 || T <= 0 || C <= 0 || K <= 0 || stride <= 0 || pad < 0) return PCM_ERR_BAD_ARG;
    const int Lout = (T + 2 * pad - K) / stride + 1;
    if (Lout <= 0) return PCM_ERR_BAD_ARG;
    const long total = (long)B * T * C;
    if (total == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    using bf = __hip_bfloat16;
#define PCM_C2I(TI, TO)                                                                                                      \
    hipLaunchKernelGGL((pcm_col2im_cl_kernel<TI, TO>), dim3(grid_for(total)), dim3(kBlock), 0, s, total, T, C, K, stride, pad, \
                       Lout, (const TI *)dcols, (TO *)dx)
    if (cols_is_bf16) {
        if (dx_is_bf16) PCM_C2I(bf, bf); else PCM_C2I(bf, float);
    } else {
        if (dx_is_bf16) PCM_C2I(float, bf); else PCM_C2I(float, float);
    }
#undef PCM_C2I
    return PCM_LAUNCH_STATUS();
}
