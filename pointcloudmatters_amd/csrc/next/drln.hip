// drln.hip -- fused  out = LayerNorm(x + dropout(y))  forward / backward for gfx950 (MI355X).  HBM bound.
//
// Every post-norm sub-layer of the ACT transformer ends with this chain
//   (/root/reference/src/models/components/act/transformer.py:250-256 encoder, :330-345 decoder:
//    `src = src + self.dropout1(src2); src = self.norm1(src)`), which PyTorch runs as
//    fused_dropout -> cast -> add -> layer_norm (4 launches forward, 6+ backward, 37 sub-layers per step).
// Here: ONE forward kernel and ONE backward kernel (+ the shared fp64 partial-row reduction for the
// affine gradients).
//
// Layout: x, out, s, dout, dx (R, E) fp32 row-major; y, dy (R, E) bf16 or fp32; gamma, beta (E).
// One wave owns one row (E <= 1024, E % 256 == 0): a lane holds E/64 values as float4 chunks, row
// statistics are two DPP/shuffle reductions, no LDS.  The dropout mask is never stored: it is
// re-derived in backward from a counter-based hash of (seed, site, element index); `seed` lives in
// device memory so a captured hipGraph draws fresh masks on every replay.
//   forward : s = x + keep * y / (1-p);  mu, rstd per row;  out = (s - mu) * rstd * gamma + beta
//   backward: g = dout * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  dy = keep * dx / (1-p)
//             dgamma = sum_rows dout * xhat,  dbeta = sum_rows dout,  dysum = sum_rows dy  (per-block partial rows,
//             fp64 reduce; dysum is the bias gradient of the projection that produced y)
// Algorithmic bytes per element: forward 4 (x) + 2 (y) + 4 (s) + 4 (out) = 14; backward 4 + 4 + 4 + 2 = 14.
#include "pcm_elem.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

template <typename T, int NCH>  // E = NCH * 256
__global__ __launch_bounds__(kBlock) void pcm_drln_fwd_kernel(long R, const float *__restrict__ x, const T *__restrict__ y,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta,
                                                              float eps, float p_drop, const long *__restrict__ seed_ptr,
                                                              unsigned site, float *__restrict__ s_out, float *__restrict__ out,
                                                              float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                              const float *__restrict__ pos, long pos_n,
                                                              __hip_bfloat16 *__restrict__ sum16, __hip_bfloat16 *__restrict__ x16)
{
    constexpr int E = NCH * 256;
    // every kernel argument in registers HERE: one batch of kernarg loads, one wait ("Kernel heads" in pcm_common.hpp)
    asm volatile("" ::"s"(gridDim.x), "s"(R), "s"(x), "s"(y), "s"(gamma), "s"(beta), "s"(eps), "s"(p_drop), "s"(seed_ptr), "s"(site), "s"(s_out), "s"(out), "s"(mean_out), "s"(rstd_out), "s"(pos), "s"(pos_n), "s"(sum16), "s"(x16));
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * kWaves + (threadIdx.x >> 6), nwaves = (long)gridDim.x * kWaves;
    const bool drop = p_drop > 0.f;
    const uint64_t seed = drop ? (uint64_t)seed_ptr[0] : 0ull;
    const uint32_t thr = drop ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
    const float scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    float g[NCH][4], b[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        load4<float>(gamma + c * 256 + lane * 4, g[c]);
        load4<float>(beta + c * 256 + lane * 4, b[c]);
    }
    for (long r = wave0; r < R; r += nwaves) {
        float s[NCH][4];
        float sum = 0.f;
        // the position rows of the emitted operand are requested together with x and y (the first version asked for them behind the
        // two reductions, one 64-bit modulo and one exposed round trip per chunk)
        float pv[NCH][4];
        if (sum16 != nullptr) {
            const long p0 = (r * E) % pos_n;  // E divides pos_n: a row never wraps
#pragma unroll
            for (int c = 0; c < NCH; ++c) load4<float>(pos + p0 + c * 256 + lane * 4, pv[c]);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const long e0 = r * E + c * 256 + lane * 4;
            float xv[4], yv[4];
            load4<float>(x + e0, xv);
            load4<T>(y + e0, yv);
            // the seed is first touched HERE, behind the row's loads in program order: its own load (issued at the kernel's entry) is
            // then waited for with the row's loads in flight instead of in front of them (the opaque asm keeps the compiler from hoisting
            // the hash constants -- and with them the wait -- back in front of the loop)
            uint64_t seed_r = seed;
            asm volatile("" : "+v"(seed_r));
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float yy = (keep_elem(seed_r, site, (uint64_t)(e0 + v), thr)) ? yv[v] * scale : 0.f;
                s[c][v] = xv[v] + yy;
                sum += s[c][v];
            }
        }
        const float mu = wave_sum(sum) * (1.f / E);
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float d = s[c][v] - mu;
                sq += d * d;
            }
        const float rstd = rsqrtf(wave_sum(sq) * (1.f / E) + eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const long e0 = r * E + c * 256 + lane * 4;
            float o[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) o[v] = (s[c][v] - mu) * rstd * g[c][v] + b[c][v];
            store4<float>(s_out + e0, s[c]);
            store4<float>(out + e0, o);
            // the consumer's bf16 operands, emitted here instead of by its own add + cast launch (pcm_add_cast2_hip):
            // sum16 = bf16(out + pos) (pos broadcast over the leading rows), x16 = bf16(out)
            if (sum16 != nullptr) {
                float q[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) q[v] = o[v] + pv[c][v];
                store4<__hip_bfloat16>(sum16 + e0, q);
            }
            if (x16 != nullptr) store4<__hip_bfloat16>(x16 + e0, o);
        }
        if (lane == 0) mean_out[r] = mu, rstd_out[r] = rstd;
    }
}

// partial layout [block][2][E] = { dgamma, dbeta }
template <typename T, int NCH>
__global__ __launch_bounds__(kBlock) void pcm_drln_bwd_kernel(long R, const float *__restrict__ dout, const float *__restrict__ dout2,
                                                              const float *__restrict__ s,
                                                              const float *__restrict__ mean, const float *__restrict__ rstd,
                                                              const float *__restrict__ gamma, float p_drop,
                                                              const long *__restrict__ seed_ptr, unsigned site,
                                                              float *__restrict__ dx, T *__restrict__ dy, float *__restrict__ partial)
{
    constexpr int E = NCH * 256;
    __shared__ float lds[kWaves][3][E];
    asm volatile("" ::"s"(gridDim.x), "s"(R), "s"(dout), "s"(dout2), "s"(s), "s"(mean), "s"(rstd), "s"(gamma), "s"(p_drop), "s"(seed_ptr), "s"(site), "s"(dx), "s"(dy), "s"(partial));  // "Kernel heads", pcm_common.hpp
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wave0 = (long)blockIdx.x * kWaves + wave, nwaves = (long)gridDim.x * kWaves;
    const bool drop = p_drop > 0.f;
    const uint64_t seed = drop ? (uint64_t)seed_ptr[0] : 0ull;
    const uint32_t thr = drop ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
    const float scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    float g[NCH][4], dg[NCH][4], db[NCH][4], dys[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        load4<float>(gamma + c * 256 + lane * 4, g[c]);
#pragma unroll
        for (int v = 0; v < 4; ++v) dg[c][v] = 0.f, db[c][v] = 0.f, dys[c][v] = 0.f;
    }
    for (long r = wave0; r < R; r += nwaves) {
        const float mu = mean[r], rs = rstd[r];
        float gd[NCH][4], xh[NCH][4];
        float s1 = 0.f, s2 = 0.f;
        // every load of the row is requested before the first value is used (written chunk by chunk, the second gradient's
        // `if` made the compiler wait for each chunk's loads in turn: NCH round trips in series instead of one)
        float dvs[NCH][4], svs[NCH][4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const long e0 = r * E + c * 256 + lane * 4;
            load4<float>(dout + e0, dvs[c]);
            load4<float>(s + e0, svs[c]);
        }
        if (dout2 != nullptr) {  // the output had two consumers: their gradients are summed here, not by an add launch
            float d2[NCH][4];
#pragma unroll
            for (int c = 0; c < NCH; ++c) load4<float>(dout2 + r * E + c * 256 + lane * 4, d2[c]);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int v = 0; v < 4; ++v) dvs[c][v] += d2[c][v];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float(&dv)[4] = dvs[c];
            const float(&sv)[4] = svs[c];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                xh[c][v] = (sv[v] - mu) * rs;
                gd[c][v] = dv[v] * g[c][v];
                s1 += gd[c][v];
                s2 += gd[c][v] * xh[c][v];
                dg[c][v] += dv[v] * xh[c][v];
                db[c][v] += dv[v];
            }
        }
        float m1 = wave_sum(s1) * (1.f / E), m2 = wave_sum(s2) * (1.f / E);
        // lib_next builds this file with packed fp32 (Makefile NEXT_PK_FILES).  Left to itself the compiler keeps (m1, m2) as ONE register pair
        // and broadcasts its halves with operand-half selects -- `x - m1` is an op_sel_hi form (exact), `xhat * m2` an OP_SEL form: the one round
        // 4's hardware reproducer found wrong beside another stream's MFMA work (2 sites, NCH = 1).  As two opaque scalars they cannot be paired.
        asm volatile("" : "+v"(m1));
        asm volatile("" : "+v"(m2));
        uint64_t seed_r = seed;  // first touched behind the row's loads: "Kernel heads", pcm_common.hpp
        asm volatile("" : "+v"(seed_r));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const long e0 = r * E + c * 256 + lane * 4;
            float o[4], oy[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                o[v] = rs * (gd[c][v] - m1 - xh[c][v] * m2);
                oy[v] = (keep_elem(seed_r, site, (uint64_t)(e0 + v), thr)) ? o[v] * scale : 0.f;
                dys[c][v] += oy[v];  // column sums of dy = the bias gradient of the projection that produced y
            }
            store4<float>(dx + e0, o);
            store4<T>(dy + e0, oy);
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            lds[wave][0][c * 256 + lane * 4 + v] = dg[c][v];
            lds[wave][1][c * 256 + lane * 4 + v] = db[c][v];
            lds[wave][2][c * 256 + lane * 4 + v] = dys[c][v];
        }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * E; e += kBlock) {
        const int t = e / E, h = e % E;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) acc += lds[w][t][h];
        partial[((size_t)blockIdx.x * 3 + t) * E + h] = acc;
    }
}

// out[e] = sum over blocks of partial[block][e] in fp64 (same scheme as the SA layer's reduction)
__global__ __launch_bounds__(512) void pcm_drln_reduce_kernel(int nslots, int VH, const float *__restrict__ partial,
                                                              float *__restrict__ out, __hip_bfloat16 *__restrict__ tail_bf16, int tail_from)
{
    __shared__ double red[8][64];
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
        out[e] = (float)t;
        // the column sums of dy are the bias gradient of a bf16 projection: emit them in bf16 too (saves a cast launch)
        if (tail_bf16 && e >= tail_from) tail_bf16[e - tail_from] = __float2bfloat16((float)t);
    }
}

inline int drln_grid(long R)
{
    long blocks = (R + kWaves - 1) / kWaves;
    if (blocks > 128) blocks = 128;  // waves stride over rows; few blocks keep the dgamma/dbeta partial rows few
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

extern "C" int pcm_drln_blocks(long R) { return drln_grid(R); }

extern "C" int pcm_drln_forward2_hip(long R, int E, int y_is_bf16, const float *x, const void *y, const float *gamma,
                                     const float *beta, float eps, float p_drop, const long *seed, unsigned site, float *s,
                                     float *out, float *mean, float *rstd, const float *pos, long pos_n, void *sum_bf16,
                                     void *out_bf16, void *stream);

extern "C" int pcm_drln_forward_hip(long R, int E, int y_is_bf16, const float *x, const void *y, const float *gamma,
                                    const float *beta, float eps, float p_drop, const long *seed, unsigned site, float *s,
                                    float *out, float *mean, float *rstd, void *stream)
{
    return pcm_drln_forward2_hip(R, E, y_is_bf16, x, y, gamma, beta, eps, p_drop, seed, site, s, out, mean, rstd, nullptr, 0, nullptr,
                                 nullptr, stream);
}

extern "C" int pcm_drln_forward2_hip(long R, int E, int y_is_bf16, const float *x, const void *y, const float *gamma,
                                     const float *beta, float eps, float p_drop, const long *seed, unsigned site, float *s,
                                     float *out, float *mean, float *rstd, const float *pos, long pos_n, void *sum_bf16,
                                     void *out_bf16, void *stream)
{
    if (sum_bf16 != nullptr && (pos == nullptr || pos_n <= 0 || pos_n % E || (R * E) % pos_n)) return PCM_ERR_BAD_ARG;
    if (R <= 0) return R == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (E % 256 != 0 || E > 1024 || E <= 0) return PCM_ERR_UNSUPPORTED;
    if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && seed == nullptr)) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = drln_grid(R);
#define PCM_F(T, N)                                                                                                         \
    hipLaunchKernelGGL((pcm_drln_fwd_kernel<T, N>), dim3(grid), dim3(kBlock), 0, st, R, x, (const T *)y, gamma, beta, eps,   \
                       p_drop, seed, site, s, out, mean, rstd, pos, pos_n, (__hip_bfloat16 *)sum_bf16, (__hip_bfloat16 *)out_bf16)
    const int n = E / 256;
    if (y_is_bf16) {
        if (n == 1) PCM_F(__hip_bfloat16, 1); else if (n == 2) PCM_F(__hip_bfloat16, 2); else if (n == 3) PCM_F(__hip_bfloat16, 3); else PCM_F(__hip_bfloat16, 4);
    } else {
        if (n == 1) PCM_F(float, 1); else if (n == 2) PCM_F(float, 2); else if (n == 3) PCM_F(float, 3); else PCM_F(float, 4);
    }
#undef PCM_F
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_drln_backward2_hip(long R, int E, int y_is_bf16, const float *dout, const float *dout2, const float *s,
                                      const float *mean, const float *rstd, const float *gamma, float p_drop, const long *seed,
                                      unsigned site, float *dx, void *dy, float *partial, float *dgamma_dbeta, void *dysum_bf16,
                                      void *stream);

extern "C" int pcm_drln_backward_hip(long R, int E, int y_is_bf16, const float *dout, const float *s, const float *mean,
                                     const float *rstd, const float *gamma, float p_drop, const long *seed, unsigned site,
                                     float *dx, void *dy, float *partial, float *dgamma_dbeta, void *dysum_bf16, void *stream)
{
    return pcm_drln_backward2_hip(R, E, y_is_bf16, dout, nullptr, s, mean, rstd, gamma, p_drop, seed, site, dx, dy, partial,
                                  dgamma_dbeta, dysum_bf16, stream);
}

extern "C" int pcm_drln_backward2_hip(long R, int E, int y_is_bf16, const float *dout, const float *dout2, const float *s,
                                      const float *mean, const float *rstd, const float *gamma, float p_drop, const long *seed,
                                      unsigned site, float *dx, void *dy, float *partial, float *dgamma_dbeta, void *dysum_bf16,
                                      void *stream)
{
    if (R <= 0) return R == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (E % 256 != 0 || E > 1024 || E <= 0) return PCM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = drln_grid(R);
#define PCM_B(T, N)                                                                                                         \
    hipLaunchKernelGGL((pcm_drln_bwd_kernel<T, N>), dim3(grid), dim3(kBlock), 0, st, R, dout, dout2, s, mean, rstd, gamma, p_drop,  \
                       seed, site, dx, (T *)dy, partial)
    const int n = E / 256;
    if (y_is_bf16) {
        if (n == 1) PCM_B(__hip_bfloat16, 1); else if (n == 2) PCM_B(__hip_bfloat16, 2); else if (n == 3) PCM_B(__hip_bfloat16, 3); else PCM_B(__hip_bfloat16, 4);
    } else {
        if (n == 1) PCM_B(float, 1); else if (n == 2) PCM_B(float, 2); else if (n == 3) PCM_B(float, 3); else PCM_B(float, 4);
    }
#undef PCM_B
    if (dgamma_dbeta == nullptr) return PCM_LAUNCH_STATUS();  // partial rows only: closed later by pcm_reduce_batch_hip
    hipLaunchKernelGGL(pcm_drln_reduce_kernel, dim3((3 * E + 63) / 64), dim3(512), 0, st, grid, 3 * E, partial, dgamma_dbeta,
                       (__hip_bfloat16 *)dysum_bf16, 2 * E);
    return PCM_LAUNCH_STATUS();
}
