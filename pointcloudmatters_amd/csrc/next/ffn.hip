// ffn.hip -- fused feed-forward sub-layer of the ACT transformer for gfx950 (MI355X):
//
//     out = LayerNorm( x + dropout_b( W2 . dropout_a( relu( W1 . x + b1 ) ) + b2 ) )
//
// Replaces `src2 = linear2(dropout(relu(linear1(src)))); src = norm2(src + dropout2(src2))`
//   (/root/reference/src/models/components/act/transformer.py:253-256 encoder, :342-345 decoder).
// The shipped configs use dim_feedforward = 32 (configs/model/maniskill2_act_pcd_model.yaml:34): two
// GEMMs with an inner dimension of 32 are launch-latency, not MFMA, work -- PyTorch spends 14 launches
// per layer forward+backward on them (casts, 2 tiny GEMMs, relu, dropout, 2 bias reductions ...).
// Here the whole sub-layer is ONE kernel forward and ONE backward (+ two weight-gradient GEMMs and the
// shared partial-row reduction), fp32 end to end.
//
// Mapping: both weight matrices live in LDS for the life of a workgroup (W1 as [F][E], W2 transposed to
// [F][E]: 128 KiB for E=512, F=32, reads are conflict-free 16-byte rows); one wave owns one row: a lane
// holds E/64 = 8 activations, accumulates its share of all F hidden pre-activations, and a 5-step
// halving butterfly (31 shuffles) leaves hidden unit j = lane>>1 in each lane; the second product
// broadcasts the F hidden values with readlane.  Row statistics for the LayerNorm are wave sums.
// Dropout masks are counter-based hashes of (seed, site, element) recomputed in backward (see drln.hip).
//
// Backward per row: LayerNorm backward -> ds; dy = mask_b ds; dh = (W2^T dy) . [hd > 0] / (1-p_a);
// dx = ds + W1^T dh.  It writes dy (R,E) and dh (R,F) so that dW2 = dy^T hd and dW1 = dh^T x are two
// small GEMMs on the host side, and per-block partial rows of {dgamma, dbeta, db2, db1}.
#include "pcm_elem.hpp"

// fp32 features (1e-4 tolerance), not the bit-exact index kernels: let the compiler form FMAs here
#pragma clang fp contract(fast)

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kPad = 4;  // floats of padding per transposed-W2 row: the transposing LDS store is 4-way instead of
                         // 32-way bank-conflicted, rows stay 16-byte aligned for ds_read_b128

// sum over the 64 lanes of F=32 per-lane partials; afterwards every lane holds the total of unit (lane >> 1)
__device__ __forceinline__ float butterfly32(float (&p)[32], int lane)
{
    float a[16], b[8], c[4], d[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const bool hi = lane & 32;
        const float send = hi ? p[i] : p[i + 16], keep = hi ? p[i + 16] : p[i];
        a[i] = keep + __shfl_xor(send, 32);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool hi = lane & 16;
        const float send = hi ? a[i] : a[i + 8], keep = hi ? a[i + 8] : a[i];
        b[i] = keep + __shfl_xor(send, 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool hi = lane & 8;
        const float send = hi ? b[i] : b[i + 4], keep = hi ? b[i + 4] : b[i];
        c[i] = keep + __shfl_xor(send, 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool hi = lane & 4;
        const float send = hi ? c[i] : c[i + 2], keep = hi ? c[i + 2] : c[i];
        d[i] = keep + __shfl_xor(send, 4);
    }
    const bool hi = lane & 2;
    const float send = hi ? d[0] : d[1], keep = hi ? d[1] : d[0];
    float v = keep + __shfl_xor(send, 2);
    return v + __shfl_xor(v, 1);
}

template <int E, int F>
__device__ __forceinline__ void stage_weights(float *w1, float *w2t, const float *__restrict__ W1, const float *__restrict__ W2)
{
    // fully unrolled, independent 16-byte loads: a rolled copy loop would pay one L2 round trip per iteration
    constexpr int N4 = F * E / 4, ITER = N4 / kThreads;
    static_assert(N4 % kThreads == 0, "weight matrices must tile the workgroup");
    const float4 *W1v = reinterpret_cast<const float4 *>(W1), *W2v = reinterpret_cast<const float4 *>(W2);
    float4 a[ITER], b[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) a[it] = W1v[it * kThreads + threadIdx.x], b[it] = W2v[it * kThreads + threadIdx.x];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i4 = it * kThreads + threadIdx.x;
        *reinterpret_cast<float4 *>(w1 + i4 * 4) = a[it];
        const int c = i4 / (F / 4), j0 = (i4 % (F / 4)) * 4;  // W2 is (E, F): 4 consecutive hidden units of column c
        w2t[(j0 + 0) * (E + kPad) + c] = b[it].x;
        w2t[(j0 + 1) * (E + kPad) + c] = b[it].y;
        w2t[(j0 + 2) * (E + kPad) + c] = b[it].z;
        w2t[(j0 + 3) * (E + kPad) + c] = b[it].w;
    }
    __syncthreads();
}

// column owned by value t of a lane: 256-column chunks, 4 consecutive columns per lane inside a chunk, so a
// wave's ds_read_b128 / global 16-byte accesses are contiguous (conflict-free, fully coalesced)
__device__ __forceinline__ int col_of(int lane, int t) { return (t >> 2) * 256 + lane * 4 + (t & 3); }

template <int E, int F>
__global__ __launch_bounds__(kThreads) void pcm_ffn_ln_fwd_kernel(long R, const float *__restrict__ x, const float *__restrict__ W1,
                                                                  const float *__restrict__ b1, const float *__restrict__ W2,
                                                                  const float *__restrict__ b2, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float eps, float pa, float pb,
                                                                  const long *__restrict__ seed_ptr, unsigned site_a, unsigned site_b,
                                                                  float *__restrict__ hd_out, float *__restrict__ s_out,
                                                                  float *__restrict__ out, float *__restrict__ mean_out,
                                                                  float *__restrict__ rstd_out, const float *__restrict__ pos,
                                                                  long pos_n, __hip_bfloat16 *__restrict__ sum16,
                                                                  __hip_bfloat16 *__restrict__ x16)
{
    static_assert(F == 32 && E % 256 == 0, "specialised for dim_feedforward = 32");
    constexpr int PER = E / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // every argument in registers at the entry, i.e. their loads travel UNDER the weight staging ("Kernel heads", pcm_common.hpp; the
    // compiler had put them -- R, the pointers, the seed through its pointer: five dependent scalar round trips -- behind the barrier)
    asm volatile("" ::"s"(gridDim.x), "s"(R), "s"(x), "s"(W1), "s"(b1), "s"(W2), "s"(b2), "s"(gamma), "s"(beta), "s"(eps), "s"(pa), "s"(pb), "s"(seed_ptr), "s"(site_a), "s"(site_b), "s"(hd_out), "s"(s_out), "s"(out), "s"(mean_out), "s"(rstd_out), "s"(pos), "s"(pos_n), "s"(sum16), "s"(x16));
    float *w1 = lds, *w2t = lds + F * E;
    stage_weights<E, F>(w1, w2t, W1, W2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 1;
    const bool da = pa > 0.f, db = pb > 0.f;
    const uint64_t seed = (da || db) ? (uint64_t)seed_ptr[0] : 0ull;
    const uint32_t thr_a = da ? (uint32_t)((double)pa * 4294967296.0) : 0u, thr_b = db ? (uint32_t)((double)pb * 4294967296.0) : 0u;
    const float sc_a = da ? 1.f / (1.f - pa) : 1.f, sc_b = db ? 1.f / (1.f - pb) : 1.f;
    float g[PER], bt[PER], bias2[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) g[t] = gamma[col_of(lane, t)], bt[t] = beta[col_of(lane, t)], bias2[t] = b2[col_of(lane, t)];
    const float bias1 = b1[j];
    for (long r = (long)blockIdx.x * kWaves + wave; r < R; r += (long)gridDim.x * kWaves) {
        // each lane always reads the same 2*F*PER weights; an offset the compiler cannot prove to be zero makes the
        // LDS addresses row-dependent, so it does not hoist 512 loop-invariant values into (spilled) VGPRs
        int zoff = (int)(r >> 40);
        asm volatile("" : "+v"(zoff));
        const float *w1r = w1 + zoff, *w2r = w2t + zoff;
        float xv[PER];
#pragma unroll
        for (int t = 0; t < PER; t += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + r * E + col_of(lane, t));
            xv[t] = v.x, xv[t + 1] = v.y, xv[t + 2] = v.z, xv[t + 3] = v.w;
        }
        float part[F];
#pragma unroll
        for (int j0 = 0; j0 < F; j0 += 8) {
            float4 wq[8][PER / 4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < PER; t += 4) wq[u][t / 4] = *reinterpret_cast<const float4 *>(w1r + (j0 + u) * E + col_of(lane, t));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < PER; t += 4) {
                    const float4 w = wq[u][t / 4];
                    acc += xv[t] * w.x + xv[t + 1] * w.y + xv[t + 2] * w.z + xv[t + 3] * w.w;
                }
                part[j0 + u] = acc;
            }
            // schedule of this group: its 16 LDS reads first, then its 128 VALU operations
            __builtin_amdgcn_sched_group_barrier(0x100, 8 * (PER / 4), 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8 * 2 * PER, 0);
        }
        float h = butterfly32(part, lane) + bias1;
        h = h > 0.f ? h : 0.f;
        const float hd = (keep_elem(seed, site_a, (uint64_t)(r * F + j), thr_a)) ? h * sc_a : 0.f;
        if ((lane & 1) == 0) hd_out[r * F + j] = hd;
        float s[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t) s[t] = bias2[t];
#pragma unroll
        for (int j0 = 0; j0 < F; j0 += 8) {
            float4 wq[8][PER / 4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < PER; t += 4) wq[u][t / 4] = *reinterpret_cast<const float4 *>(w2r + (j0 + u) * (E + kPad) + col_of(lane, t));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float hj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hd), 2 * (j0 + u)));
#pragma unroll
                for (int t = 0; t < PER; t += 4) {
                    const float4 w = wq[u][t / 4];
                    s[t] += hj * w.x, s[t + 1] += hj * w.y, s[t + 2] += hj * w.z, s[t + 3] += hj * w.w;
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8 * (PER / 4), 0);  // this group's 16 LDS reads first ...
            __builtin_amdgcn_sched_group_barrier(0x002, 8 * (2 * PER + 1), 0);  // ... then its arithmetic
        }
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const float y = (keep_elem(seed, site_b, (uint64_t)(r * E + col_of(lane, t)), thr_b)) ? s[t] * sc_b : 0.f;
            s[t] = xv[t] + y;
            sum += s[t];
        }
        const float mu = wave_sum(sum) * (1.f / E);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < PER; ++t) sq += (s[t] - mu) * (s[t] - mu);
        const float rstd = rsqrtf(wave_sum(sq) * (1.f / E) + eps);
#pragma unroll
        for (int t = 0; t < PER; t += 4) {
            float o[4], sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sv[u] = s[t + u], o[u] = (s[t + u] - mu) * rstd * g[t + u] + bt[t + u];
            store4<float>(s_out + r * E + col_of(lane, t), sv);
            store4<float>(out + r * E + col_of(lane, t), o);
            // the next layer's in-projection operands, emitted here (see pcm_drln_forward2_hip)
            if (sum16 != nullptr) {
                const long e0 = r * E + col_of(lane, t);
                float p[4], q[4];
                load4<float>(pos + (r * E) % pos_n + col_of(lane, t), p);  // E divides pos_n: a row never wraps
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = o[u] + p[u];
                store4<__hip_bfloat16>(sum16 + e0, q);
            }
            if (x16 != nullptr) store4<__hip_bfloat16>(x16 + r * E + col_of(lane, t), o);
        }
        if (lane == 0) mean_out[r] = mu, rstd_out[r] = rstd;
    }
}

// partial layout per block: [ dgamma(E) | dbeta(E) | db2(E) | db1(F) ]
template <int E, int F>
__global__ __launch_bounds__(kThreads) void pcm_ffn_ln_bwd_kernel(long R, const float *__restrict__ dout, const float *__restrict__ dout2,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ s, const float *__restrict__ mean,
                                                                  const float *__restrict__ rstd, const float *__restrict__ hd,
                                                                  const float *__restrict__ W1, const float *__restrict__ W2,
                                                                  const float *__restrict__ gamma, float pa, float pb,
                                                                  const long *__restrict__ seed_ptr, unsigned site_b,
                                                                  float *__restrict__ dx, float *__restrict__ dy,
                                                                  float *__restrict__ dh_out, float *__restrict__ partial)
{
    constexpr int PER = E / 64;
    constexpr int PW = 3 * E + F;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    asm volatile("" ::"s"(gridDim.x), "s"(R), "s"(dout), "s"(dout2), "s"(x), "s"(s), "s"(mean), "s"(rstd), "s"(hd), "s"(W1), "s"(W2), "s"(gamma), "s"(pa), "s"(pb), "s"(seed_ptr), "s"(site_b), "s"(dx), "s"(dy), "s"(dh_out), "s"(partial));  // "Kernel heads", pcm_common.hpp
    float *w1 = lds, *w2t = lds + F * E;
    stage_weights<E, F>(w1, w2t, W1, W2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 1;
    const bool db = pb > 0.f;
    const uint64_t seed = db ? (uint64_t)seed_ptr[0] : 0ull;
    const uint32_t thr_b = db ? (uint32_t)((double)pb * 4294967296.0) : 0u;
    const float sc_a = pa > 0.f ? 1.f / (1.f - pa) : 1.f, sc_b = db ? 1.f / (1.f - pb) : 1.f;
    float g[PER], dg[PER], dbt[PER], db2[PER];
    float db1 = 0.f;
#pragma unroll
    for (int t = 0; t < PER; ++t) g[t] = gamma[col_of(lane, t)], dg[t] = 0.f, dbt[t] = 0.f, db2[t] = 0.f;
    for (long r = (long)blockIdx.x * kWaves + wave; r < R; r += (long)gridDim.x * kWaves) {
        int zoff = (int)(r >> 40);  // see the forward kernel: keeps the LDS weight reads inside the row loop
        asm volatile("" : "+v"(zoff));
        const float *w1r = w1 + zoff, *w2r = w2t + zoff;
        const float mu = mean[r], rs = rstd[r];
        float gd[PER], xh[PER];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < PER; t += 4) {
            float dv[4], sv[4];
            load4<float>(dout + r * E + col_of(lane, t), dv);
            if (dout2 != nullptr) {  // second consumer's gradient, summed here instead of by an add launch
                float d2[4];
                load4<float>(dout2 + r * E + col_of(lane, t), d2);
#pragma unroll
                for (int u = 0; u < 4; ++u) dv[u] += d2[u];
            }
            load4<float>(s + r * E + col_of(lane, t), sv);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xh[t + u] = (sv[u] - mu) * rs;
                gd[t + u] = dv[u] * g[t + u];
                s1 += gd[t + u];
                s2 += gd[t + u] * xh[t + u];
                dg[t + u] += dv[u] * xh[t + u];
                dbt[t + u] += dv[u];
            }
        }
        const float m1 = wave_sum(s1) * (1.f / E), m2 = wave_sum(s2) * (1.f / E);
        float ds[PER], dyv[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            ds[t] = rs * (gd[t] - m1 - xh[t] * m2);
            dyv[t] = (keep_elem(seed, site_b, (uint64_t)(r * E + col_of(lane, t)), thr_b)) ? ds[t] * sc_b : 0.f;
            db2[t] += dyv[t];
        }
#pragma unroll
        for (int t = 0; t < PER; t += 4) {
            const float o[4] = {dyv[t], dyv[t + 1], dyv[t + 2], dyv[t + 3]};
            store4<float>(dy + r * E + col_of(lane, t), o);
        }
        float part[F];
#pragma unroll
        for (int j0 = 0; j0 < F; j0 += 8) {  // 16 LDS reads requested together, then consumed (see the forward kernel)
            float4 wq[8][PER / 4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < PER; t += 4) wq[u][t / 4] = *reinterpret_cast<const float4 *>(w2r + (j0 + u) * (E + kPad) + col_of(lane, t));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < PER; t += 4) {
                    const float4 w = wq[u][t / 4];
                    acc += dyv[t] * w.x + dyv[t + 1] * w.y + dyv[t + 2] * w.z + dyv[t + 3] * w.w;
                }
                part[j0 + u] = acc;
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8 * (PER / 4), 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8 * 2 * PER, 0);
        }
        const float dhd = butterfly32(part, lane);
        const float hdv = hd[r * F + j];
        const float dh = hdv > 0.f ? dhd * sc_a : 0.f;  // hd > 0  <=>  relu active AND kept by dropout_a
        if ((lane & 1) == 0) {
            dh_out[r * F + j] = dh;
            db1 += dh;
        }
#pragma unroll
        for (int j0 = 0; j0 < F; j0 += 8) {
            float4 wq[8][PER / 4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < PER; t += 4) wq[u][t / 4] = *reinterpret_cast<const float4 *>(w1r + (j0 + u) * E + col_of(lane, t));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float dj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dh), 2 * (j0 + u)));
#pragma unroll
                for (int t = 0; t < PER; t += 4) {
                    const float4 w = wq[u][t / 4];
                    ds[t] += dj * w.x, ds[t + 1] += dj * w.y, ds[t + 2] += dj * w.z, ds[t + 3] += dj * w.w;
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8 * (PER / 4), 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8 * (2 * PER + 1), 0);
        }
#pragma unroll
        for (int t = 0; t < PER; t += 4) {
            const float o[4] = {ds[t], ds[t + 1], ds[t + 2], ds[t + 3]};
            store4<float>(dx + r * E + col_of(lane, t), o);
        }
    }
    // block-level combine of the per-wave column sums in the (now idle) weight region of LDS
    __syncthreads();
    float *red = lds;  // [kWaves][PW]
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        red[wave * PW + col_of(lane, t)] = dg[t];
        red[wave * PW + E + col_of(lane, t)] = dbt[t];
        red[wave * PW + 2 * E + col_of(lane, t)] = db2[t];
    }
    if ((lane & 1) == 0) red[wave * PW + 3 * E + j] = db1;
    __syncthreads();
    for (int e = threadIdx.x; e < PW; e += kThreads) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) acc += red[w * PW + e];
        partial[(size_t)blockIdx.x * PW + e] = acc;
    }
}

__global__ __launch_bounds__(512) void pcm_ffn_reduce_kernel(int nslots, int VH, const float *__restrict__ partial, float *__restrict__ out)
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
    }
}

inline int ffn_grid(long R)
{
    long blocks = (R + kWaves - 1) / kWaves;
    if (blocks > 256) blocks = 256;  // 128 KiB of LDS per workgroup: one per CU
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <typename K>
int set_lds(K kernel, size_t bytes)
{
    return pcm_status(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

}  // namespace

extern "C" int pcm_ffn_ln_supported(int E, int F) { return (F == 32 && (E == 256 || E == 512)) ? 1 : 0; }

extern "C" int pcm_ffn_reduce_rows_hip(int nslots, int VH, const float *partial, float *out, void *stream)
{
    if (nslots <= 0 || VH <= 0 || !partial || !out) return PCM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pcm_ffn_reduce_kernel, dim3((VH + 63) / 64), dim3(512), 0, (hipStream_t)stream, nslots, VH, partial, out);
    return PCM_LAUNCH_STATUS();
}
extern "C" int pcm_ffn_ln_blocks(long R) { return ffn_grid(R); }

extern "C" int pcm_ffn_ln_forward2_hip(long R, int E, int F, const float *x, const float *W1, const float *b1, const float *W2,
                                       const float *b2, const float *gamma, const float *beta, float eps, float p_hidden,
                                       float p_out, const long *seed, unsigned site_a, unsigned site_b, float *hd, float *s,
                                       float *out, float *mean, float *rstd, const float *pos, long pos_n, void *sum_bf16,
                                       void *out_bf16, void *stream);

extern "C" int pcm_ffn_ln_forward_hip(long R, int E, int F, const float *x, const float *W1, const float *b1, const float *W2,
                                      const float *b2, const float *gamma, const float *beta, float eps, float p_hidden,
                                      float p_out, const long *seed, unsigned site_a, unsigned site_b, float *hd, float *s,
                                      float *out, float *mean, float *rstd, void *stream)
{
    return pcm_ffn_ln_forward2_hip(R, E, F, x, W1, b1, W2, b2, gamma, beta, eps, p_hidden, p_out, seed, site_a, site_b, hd, s, out,
                                   mean, rstd, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int pcm_ffn_ln_forward2_hip(long R, int E, int F, const float *x, const float *W1, const float *b1, const float *W2,
                                       const float *b2, const float *gamma, const float *beta, float eps, float p_hidden,
                                       float p_out, const long *seed, unsigned site_a, unsigned site_b, float *hd, float *s,
                                       float *out, float *mean, float *rstd, const float *pos, long pos_n, void *sum_bf16,
                                       void *out_bf16, void *stream)
{
    if (sum_bf16 != nullptr && (pos == nullptr || pos_n <= 0 || pos_n % E || (R * E) % pos_n)) return PCM_ERR_BAD_ARG;
    if (R <= 0) return R == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (!pcm_ffn_ln_supported(E, F)) return PCM_ERR_UNSUPPORTED;
    if ((p_hidden > 0.f || p_out > 0.f) && seed == nullptr) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)F * E + (size_t)F * (E + kPad)) * sizeof(float);
    const int grid = ffn_grid(R);
#define PCM_FF(EE)                                                                                                          \
    do {                                                                                                                     \
        auto k = pcm_ffn_ln_fwd_kernel<EE, 32>;                                                                              \
        int rc = set_lds(k, lds);                                                                                            \
        if (rc) return rc;                                                                                                   \
        hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds, st, R, x, W1, b1, W2, b2, gamma, beta, eps, p_hidden, p_out,  \
                           seed, site_a, site_b, hd, s, out, mean, rstd, pos, pos_n, (__hip_bfloat16 *)sum_bf16,             \
                           (__hip_bfloat16 *)out_bf16);                                                                      \
    } while (0)
    if (E == 512) PCM_FF(512); else PCM_FF(256);
#undef PCM_FF
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_ffn_ln_backward2_hip(long R, int E, int F, const float *dout, const float *dout2, const float *x, const float *s,
                                        const float *mean, const float *rstd, const float *hd, const float *W1, const float *W2,
                                        const float *gamma, float p_hidden, float p_out, const long *seed, unsigned site_b,
                                        float *dx, float *dy, float *dh, float *partial, float *sums, void *stream);

extern "C" int pcm_ffn_ln_backward_hip(long R, int E, int F, const float *dout, const float *x, const float *s, const float *mean,
                                       const float *rstd, const float *hd, const float *W1, const float *W2, const float *gamma,
                                       float p_hidden, float p_out, const long *seed, unsigned site_b, float *dx, float *dy,
                                       float *dh, float *partial, float *sums, void *stream)
{
    return pcm_ffn_ln_backward2_hip(R, E, F, dout, nullptr, x, s, mean, rstd, hd, W1, W2, gamma, p_hidden, p_out, seed, site_b, dx,
                                    dy, dh, partial, sums, stream);
}

extern "C" int pcm_ffn_ln_backward2_hip(long R, int E, int F, const float *dout, const float *dout2, const float *x, const float *s,
                                        const float *mean, const float *rstd, const float *hd, const float *W1, const float *W2,
                                        const float *gamma, float p_hidden, float p_out, const long *seed, unsigned site_b,
                                        float *dx, float *dy, float *dh, float *partial, float *sums, void *stream)
{
    if (R <= 0) return R == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (!pcm_ffn_ln_supported(E, F)) return PCM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)F * E + (size_t)F * (E + kPad)) * sizeof(float);
    const int grid = ffn_grid(R);
#define PCM_FB(EE)                                                                                                          \
    do {                                                                                                                     \
        auto k = pcm_ffn_ln_bwd_kernel<EE, 32>;                                                                              \
        int rc = set_lds(k, lds);                                                                                            \
        if (rc) return rc;                                                                                                   \
        hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds, st, R, dout, dout2, x, s, mean, rstd, hd, W1, W2, gamma, p_hidden,   \
                           p_out, seed, site_b, dx, dy, dh, partial);                                                        \
    } while (0)
    if (E == 512) PCM_FB(512); else PCM_FB(256);
#undef PCM_FB
    const int PW = 3 * E + F;
    if (sums == nullptr) return PCM_LAUNCH_STATUS();  // partial rows only: closed later by pcm_reduce_batch_hip
    hipLaunchKernelGGL(pcm_ffn_reduce_kernel, dim3((PW + 63) / 64), dim3(512), 0, st, grid, PW, partial, sums);
    return PCM_LAUNCH_STATUS();
}
