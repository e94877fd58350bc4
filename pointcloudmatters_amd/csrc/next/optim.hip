// optim.hip -- gradient-norm clipping + AdamW over ONE flat fp32 parameter buffer (gfx950).
//
// Replaces, on the training step of the hot path, what Lightning/torch run as separate multi-tensor
// passes in the reference:  clip_grad_norm_(0.5)  ->  AdamW.step()
//   (/root/reference/configs/trainer/ddp.yaml:12 gradient_clip_val,
//    /root/reference/src/models/maniskill2_act_bc_module.py:347-367 configure_optimizers,
//    /root/reference/configs/model/maniskill2_act_pcd_model.yaml:11-14 AdamW lr 5e-5 wd 0.05).
//
// Layout: all trainable parameters live back to back in one HBM buffer p[n]; g, m, v mirror it.
// Two launches per optimizer step, both HBM-streaming with 16-byte accesses:
//   pcm_grad_sumsq  : per-block partial sums of g^2 (fixed grid -> deterministic), 4 B/elem read
//   pcm_adamw_flat  : every block re-reduces the <=1024 partials (L2-resident, free), derives the
//                     clip coefficient and applies decoupled-weight-decay Adam: 16 B read + 12 B
//                     written per element.  Algorithmic bytes per step: 32 * n.
// Every step-dependent scalar (lr, beta1 -- OneCycleLR cycles it --, bias corrections, gradient
// scale) is read from a small DEVICE array, so the launches can sit inside a captured hipGraph and
// the host only rewrites that array between replays.
//
// Update rule = torch.optim.AdamW (single-tensor form), per element, fp32:
//   g   = grad * grad_scale * clip_coef          clip_coef = min(1, max_norm / (||grad*scale|| + 1e-6))
//   p  *= 1 - lr * wd
//   m  += (g - m) * (1 - beta1)                  (lerp)
//   v   = v * beta2 + (1 - beta2) * g * g
//   p  -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
#include <stdlib.h>
#include "pcm_common.hpp"

#include <hip/hip_bf16.h>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxPartials = 1024;

// hyper[] indices
enum { H_LR = 0, H_BETA1, H_BETA2, H_EPS, H_WD, H_BC1, H_BC2_SQRT, H_MAX_NORM, H_GRAD_SCALE, H_COUNT };

__device__ __forceinline__ float block_sum(float v, float *red)
{
    // wave reduce (DPP butterfly on the float bits is not associative-safe to share with u32 helpers; use shfl)
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < kBlock / 64; ++w) s += red[w];  // same order in every thread
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(kBlock) void pcm_grad_sumsq_kernel(long n4, long n, const float *__restrict__ g,
                                                                float *__restrict__ partials)
{
    __shared__ float red[kBlock / 64];
    float acc = 0.f;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        const float4 x = g4[i];
        acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    if (blockIdx.x == 0) {  // tail (n not a multiple of 4)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += kBlock) acc += g[i] * g[i];
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float lr, float b1, float b2, float eps,
                                          float wd, float bc1, float bc2s, float gscale)
{
    g = g * gscale;
    p = p * (1.f - lr * wd);
    m = m + (g - m) * (1.f - b1);
    v = v * b2 + (1.f - b2) * g * g;
    const float denom = sqrtf(v) / bc2s + eps;
    p = p - (lr / bc1) * (m / denom);
}

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4 *p)
{
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store(const float4 &v, float4 *p)
{
    f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p));
}

template <int MODE>  // bit 0: nontemporal gradient loads, bit 1: nontemporal p / m / v stores (measured: 133 -> 121 us at 24.1 M)
__global__ __launch_bounds__(kBlock) void pcm_adamw_flat_kernel(long n4, long n, float *__restrict__ p,
                                                                const float *__restrict__ g, float *__restrict__ m,
                                                                float *__restrict__ v, const float *__restrict__ hyper,
                                                                const float *__restrict__ partials, int npartials,
                                                                float *__restrict__ norm_out,
                                                                __hip_bfloat16 *__restrict__ p_bf16)
{
    __shared__ float red[kBlock / 64];
    const float lr = hyper[H_LR], b1 = hyper[H_BETA1], b2 = hyper[H_BETA2], eps = hyper[H_EPS], wd = hyper[H_WD];
    const float bc1 = hyper[H_BC1], bc2s = hyper[H_BC2_SQRT], max_norm = hyper[H_MAX_NORM], scale = hyper[H_GRAD_SCALE];
    float gscale = scale;
    if (npartials > 0) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < npartials; i += kBlock) acc += partials[i];
        const float total = sqrtf(block_sum(acc, red)) * fabsf(scale);  // ||grad * scale||_2
        if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
        if (max_norm > 0.f) {
            const float coef = max_norm / (total + 1e-6f);
            gscale = scale * (coef < 1.f ? coef : 1.f);
        }
    }
    float4 *p4 = reinterpret_cast<float4 *>(p);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    float4 *m4 = reinterpret_cast<float4 *>(m);
    float4 *v4 = reinterpret_cast<float4 *>(v);
    // two independent float4 quads per iteration (8 loads in flight per thread); g is read once and m / v / p are not
    // touched again before the next step, so they bypass the caches (nontemporal) and leave the MALL to the activations
    const long stride = (long)gridDim.x * kBlock;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += 2 * stride) {
        const long j = i + stride;
        const bool two = j < n4;
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = (MODE & 1) ? nt_load(g4 + i) : g4[i];
        float4 pq = pp, mq = mm, vq = vv, gq = gg;
        if (two) pq = p4[j], mq = m4[j], vq = v4[j], gq = (MODE & 1) ? nt_load(g4 + j) : g4[j];
        adam_elem(pp.x, gg.x, mm.x, vv.x, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
        adam_elem(pp.y, gg.y, mm.y, vv.y, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
        adam_elem(pp.z, gg.z, mm.z, vv.z, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
        adam_elem(pp.w, gg.w, mm.w, vv.w, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
        if (MODE & 2) nt_store(pp, p4 + i); else p4[i] = pp;
        if (MODE & 2) nt_store(mm, m4 + i); else m4[i] = mm;
        if (MODE & 2) nt_store(vv, v4 + i); else v4[i] = vv;
        if (p_bf16) {  // bf16 mirror of the weights for the next step's GEMMs (no per-weight cast kernels)
            *reinterpret_cast<uint2 *>(p_bf16 + i * 4) = make_uint2(pcm_cvt_pk_bf16(pp.x, pp.y), pcm_cvt_pk_bf16(pp.z, pp.w));
        }
        if (two) {
            adam_elem(pq.x, gq.x, mq.x, vq.x, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
            adam_elem(pq.y, gq.y, mq.y, vq.y, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
            adam_elem(pq.z, gq.z, mq.z, vq.z, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
            adam_elem(pq.w, gq.w, mq.w, vq.w, lr, b1, b2, eps, wd, bc1, bc2s, gscale);
            if (MODE & 2) nt_store(pq, p4 + j); else p4[j] = pq;
            if (MODE & 2) nt_store(mq, m4 + j); else m4[j] = mq;
            if (MODE & 2) nt_store(vq, v4 + j); else v4[j] = vq;
            if (p_bf16) {
                *reinterpret_cast<uint2 *>(p_bf16 + j * 4) = make_uint2(pcm_cvt_pk_bf16(pq.x, pq.y), pcm_cvt_pk_bf16(pq.z, pq.w));
            }
        }
    }
    if (blockIdx.x == 0) {
        for (long i = n4 * 4 + threadIdx.x; i < n; i += kBlock) {
            adam_elem(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, bc1, bc2s, gscale);
            if (p_bf16) p_bf16[i] = __float2bfloat16(p[i]);
        }
    }
}

inline int stream_grid(long n4)
{
    long blocks = (n4 + kBlock - 1) / kBlock;
    if (blocks > kMaxPartials) blocks = kMaxPartials;  // 4 workgroups per CU, grid-stride
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

extern "C" int pcm_optim_partials_capacity(void) { return kMaxPartials; }

extern "C" int pcm_grad_sumsq_hip(long n, const float *g, float *partials, int *npartials_out, void *stream)
{
    if (n < 0 || ((uintptr_t)g % 16) != 0) return PCM_ERR_BAD_ARG;
    if (n == 0) {  // no elements: no partial sums (pcm_adamw_flat_hip accepts npartials = 0)
        if (npartials_out) *npartials_out = 0;
        return PCM_OK;
    }
    const long n4 = n / 4;
    const int grid = stream_grid(n4);
    if (npartials_out) *npartials_out = grid;
    hipLaunchKernelGGL(pcm_grad_sumsq_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, n4, n, g, partials);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_adamw_flat_hip(long n, float *p, const float *g, float *m, float *v, const float *hyper,
                                  const float *partials, int npartials, float *norm_out, void *p_bf16, void *stream)
{
    if (n < 0 || npartials < 0 || npartials > kMaxPartials) return PCM_ERR_BAD_ARG;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) != 0) return PCM_ERR_BAD_ARG;
    if (n == 0) return PCM_OK;  // nothing to update (norm_out is left untouched)
    const long n4 = n / 4;
    // seven address streams (p, m, v, g in; p, m, v, bf16 mirror out): FEWER workgroups keep them more DRAM-page-friendly.
    // Measured (us, 24.1 M / 255.6 M parameters): 1024 blocks 122 / 1634, 512: 121 / 1416, 256: 140 / 1366.
    int grid = stream_grid(n4);
    const int cap = n4 >= (32L << 20) ? 256 : 512;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(pcm_adamw_flat_kernel<3>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, n4, n, p, g, m, v,
                       hyper, partials, npartials, norm_out, (__hip_bfloat16 *)p_bf16);
    return PCM_LAUNCH_STATUS();
}


// ---- gradient hand-off: the micro-batch's gradients into the flat fp32 buffer, ONE launch per 96 tensors ------------------------------
// What `loss.backward()` + DDP's bucket copies + `optimizer.zero_grad()` do tensor by tensor in the reference's loop
// (/root/reference/src/models/maniskill2_act_bc_module.py:64-86 under Lightning).  Round 3 used the framework's multi-tensor copies:
// five launches, 85 us for 24 M parameters at C2 (2.2 TB/s: 64 K-element chunks, one block each).  Here every job is cut into 8192-element
// chunks, one workgroup per chunk, 16-byte accesses; jobs travel BY VALUE in the kernel arguments, so a captured graph keeps them.
namespace {
constexpr int kXferBatch = 96;
constexpr int kXferChunk = 8192;
struct XferBatch {
    void *dst[kXferBatch];
    const void *src[kXferBatch];
    int numel[kXferBatch];
    int chunk0[kXferBatch + 1];  // first chunk of job i; chunk0[n] = grid
    unsigned char kind[kXferBatch];
    int n;
};

__global__ __launch_bounds__(256) void pcm_xfer_batch_kernel(XferBatch b)
{
    const int c = blockIdx.x;
    // the job whose chunk range holds c: one lane-indexed load of the table + a ballot (was a bisection: 7 dependent scalar loads)
    const int lo = pcm_job_of(c, b.n, [&](int j) { return b.chunk0[j]; });
    asm volatile("" ::"s"((int)b.kind[lo]), "s"(b.chunk0[lo]), "s"(b.numel[lo]), "s"(b.dst[lo]), "s"(b.src[lo]));  // one batch: "Kernel heads", pcm_common.hpp
    const int kind = b.kind[lo];
    const long e0 = (long)(c - b.chunk0[lo]) * kXferChunk;
    const int len = b.numel[lo] - e0 < kXferChunk ? (int)(b.numel[lo] - e0) : kXferChunk;
    if (kind == PCM_XFER_COPY_2B) {  // 2-byte elements, raw
        unsigned short *d = static_cast<unsigned short *>(b.dst[lo]) + e0;
        const unsigned short *s_ = static_cast<const unsigned short *>(b.src[lo]) + e0;
        const bool vec = ((((uintptr_t)d) | ((uintptr_t)s_)) & 15) == 0;
        if (vec && len == kXferChunk) {  // whole aligned chunk: every load in flight before the first store (the loop below: load, wait, store)
            uint4 q[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] = *reinterpret_cast<const uint4 *>(s_ + (threadIdx.x + 256 * t) * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<uint4 *>(d + (threadIdx.x + 256 * t) * 8) = q[t];
            return;
        }
        for (int i = threadIdx.x * 8; i < len; i += 256 * 8) {
            if (vec && i + 8 <= len) {
                *reinterpret_cast<uint4 *>(d + i) = *reinterpret_cast<const uint4 *>(s_ + i);
            } else {
                for (int k = i; k < len && k < i + 8; ++k) d[k] = s_[k];
            }
        }
        return;
    }
    float *d = static_cast<float *>(b.dst[lo]) + e0;
    const bool add = kind == PCM_XFER_ADD_BF16 || kind == PCM_XFER_ADD_F32;
    const bool dvec = (((uintptr_t)d) & 15) == 0;
    if (kind == PCM_XFER_ZERO) {
        for (int i = threadIdx.x * 4; i < len; i += 256 * 4) {
            if (dvec && i + 4 <= len) {
                *reinterpret_cast<float4 *>(d + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int k = i; k < len && k < i + 4; ++k) d[k] = 0.f;
            }
        }
    } else if (kind == PCM_XFER_SET_BF16 || kind == PCM_XFER_ADD_BF16) {
        const unsigned short *s_ = static_cast<const unsigned short *>(b.src[lo]) + e0;
        const bool vec = dvec && (((uintptr_t)s_) & 15) == 0;
        if (vec && len == kXferChunk) {  // whole aligned chunk: every load in flight before the first store; same sums
            uint4 q[4];
            float4 u[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] = *reinterpret_cast<const uint4 *>(s_ + (threadIdx.x + 256 * t) * 8);
            if (add) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    u[t][0] = *reinterpret_cast<const float4 *>(d + (threadIdx.x + 256 * t) * 8);
                    u[t][1] = *reinterpret_cast<const float4 *>(d + (threadIdx.x + 256 * t) * 8 + 4);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float4 a = make_float4(__uint_as_float(q[t].x << 16), __uint_as_float(q[t].x & 0xFFFF0000u), __uint_as_float(q[t].y << 16), __uint_as_float(q[t].y & 0xFFFF0000u));
                float4 c4 = make_float4(__uint_as_float(q[t].z << 16), __uint_as_float(q[t].z & 0xFFFF0000u), __uint_as_float(q[t].w << 16), __uint_as_float(q[t].w & 0xFFFF0000u));
                if (add) {
                    a.x += u[t][0].x, a.y += u[t][0].y, a.z += u[t][0].z, a.w += u[t][0].w;
                    c4.x += u[t][1].x, c4.y += u[t][1].y, c4.z += u[t][1].z, c4.w += u[t][1].w;
                }
                *reinterpret_cast<float4 *>(d + (threadIdx.x + 256 * t) * 8) = a;
                *reinterpret_cast<float4 *>(d + (threadIdx.x + 256 * t) * 8 + 4) = c4;
            }
            return;
        }
        for (int i = threadIdx.x * 8; i < len; i += 256 * 8) {
            if (vec && i + 8 <= len) {
                const uint4 q = *reinterpret_cast<const uint4 *>(s_ + i);
                float4 a = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xFFFF0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xFFFF0000u));
                float4 c4 = make_float4(__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xFFFF0000u), __uint_as_float(q.w << 16), __uint_as_float(q.w & 0xFFFF0000u));
                if (add) {
                    const float4 u = *reinterpret_cast<const float4 *>(d + i), v = *reinterpret_cast<const float4 *>(d + i + 4);
                    a.x += u.x, a.y += u.y, a.z += u.z, a.w += u.w, c4.x += v.x, c4.y += v.y, c4.z += v.z, c4.w += v.w;
                }
                *reinterpret_cast<float4 *>(d + i) = a;
                *reinterpret_cast<float4 *>(d + i + 4) = c4;
            } else {
                for (int k = i; k < len && k < i + 8; ++k) {
                    const float v = __uint_as_float((uint32_t)s_[k] << 16);
                    d[k] = add ? d[k] + v : v;
                }
            }
        }
    } else {  // fp32 source
        const float *s_ = static_cast<const float *>(b.src[lo]) + e0;
        const bool vec = dvec && (((uintptr_t)s_) & 15) == 0;
        if (vec && len == kXferChunk) {  // whole aligned chunk: every load in flight before the first store; same sums
            float4 a[8], u[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) a[t] = *reinterpret_cast<const float4 *>(s_ + (threadIdx.x + 256 * t) * 4);
            if (add) {
#pragma unroll
                for (int t = 0; t < 8; ++t) u[t] = *reinterpret_cast<const float4 *>(d + (threadIdx.x + 256 * t) * 4);
#pragma unroll
                for (int t = 0; t < 8; ++t) a[t].x += u[t].x, a[t].y += u[t].y, a[t].z += u[t].z, a[t].w += u[t].w;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) *reinterpret_cast<float4 *>(d + (threadIdx.x + 256 * t) * 4) = a[t];
            return;
        }
        for (int i = threadIdx.x * 4; i < len; i += 256 * 4) {
            if (vec && i + 4 <= len) {
                float4 a = *reinterpret_cast<const float4 *>(s_ + i);
                if (add) {
                    const float4 u = *reinterpret_cast<const float4 *>(d + i);
                    a.x += u.x, a.y += u.y, a.z += u.z, a.w += u.w;
                }
                *reinterpret_cast<float4 *>(d + i) = a;
            } else {
                for (int k = i; k < len && k < i + 4; ++k) d[k] = add ? d[k] + s_[k] : s_[k];
            }
        }
    }
}
}  // namespace

// n jobs (host arrays): kind[i] in PCM_XFER_* -- ZERO: dst[i][0..numel) = 0 (fp32; src ignored); SET_BF16 / SET_F32: dst (fp32) = src;
// ADD_BF16 / ADD_F32: dst (fp32) += src (the widened bf16 value is exact, so the sum is the one a cast-then-add gives); COPY_2B: raw copy
// of 2-byte elements.  Jobs must not overlap each other.  96 jobs per launch, one workgroup per 8192 elements.
extern "C" int pcm_xfer_batch_hip(int n, void *const *dst, const void *const *src, const long *numel, const int *kind, void *stream)
{
    if (n < 0 || (n > 0 && (!dst || !src || !numel || !kind))) return PCM_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i) {
        if (numel[i] < 0 || numel[i] > 0x7FFFFFFFL || kind[i] < PCM_XFER_ZERO || kind[i] > PCM_XFER_COPY_2B) return PCM_ERR_BAD_ARG;
        if (numel[i] > 0 && (!dst[i] || (kind[i] != PCM_XFER_ZERO && !src[i]))) return PCM_ERR_BAD_ARG;
    }
    if (n == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    int i = 0;
    bool launched = false;
    while (i < n) {
        XferBatch b;
        b.n = 0;
        long chunks = 0;
        for (; i < n && b.n < kXferBatch; ++i) {
            if (numel[i] == 0) continue;
            const int j = b.n++;
            b.dst[j] = dst[i], b.src[j] = src[i], b.numel[j] = (int)numel[i], b.kind[j] = (unsigned char)kind[i], b.chunk0[j] = (int)chunks;
            chunks += (numel[i] + kXferChunk - 1) / kXferChunk;
        }
        if (chunks > 0x7FFFFFFF) return PCM_ERR_BAD_ARG;
        b.chunk0[b.n] = (int)chunks;
        if (b.n) {
            hipLaunchKernelGGL(pcm_xfer_batch_kernel, dim3((unsigned)chunks), dim3(256), 0, s, b);
            launched = true;
        }
    }
    return launched ? PCM_LAUNCH_STATUS() : PCM_OK;  // only empty jobs: nothing was launched, nothing to report
}
