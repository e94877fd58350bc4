// ffn_mfma.hip -- the fused feed-forward sub-layer of the ACT transformer on the matrix cores (gfx950, bf16 autocast path):
//
//     out = LayerNorm( x + dropout_b( W2 . dropout_a( relu( W1 . x + b1 ) ) + b2 ) )
//
// Same sub-layer as csrc/ffn.hip (/root/reference/src/models/components/act/transformer.py:253-256 encoder, :342-345 decoder;
// dim_feedforward = 32, configs/model/maniskill2_act_pcd_model.yaml:34), same entry-point arguments (its own, cheaper dropout generator, see Drop below).
// ffn.hip is the fp32 path: every row re-reads both weight matrices from LDS (128 KiB staged per workgroup, 540 MB of LDS reads at
// 4120 rows) -- 14-21 us per launch, LDS-bandwidth bound, and 15 us even at 800 rows because 256 workgroups stage the weights.
// Under bf16 autocast the two products ARE bf16 GEMMs in the reference recipe, so here they run on v_mfma_f32_32x32x16_bf16:
//
//   * one workgroup (8 waves) per 32-row tile; wave w owns the channels [E/8 * w, E/8 * (w+1)): nothing is staged in LDS, the
//     weights are read straight from L2 as MFMA A operands (64 KiB per tile instead of 128 KiB per workgroup + per-row re-reads);
//   * everything is computed TRANSPOSED (lane = row of the tile, as in the attention kernels): H^T = W1 X^T per wave over its
//     K range, the eight partial 32 x 32 tiles meet in LDS (32 KiB), and -- because the K order of an MFMA is free as long as A and
//     B agree -- the accumulator layout of H^T (lane = row, registers = hidden units {4h..4h+3, 8+4h.., ...}) IS the B-operand
//     layout of the second product Y^T = W2 H^T: the hidden activations never leave registers;
//   * the same K permutation is applied to the first product, so the x values a lane loaded as B operand are exactly the ones it
//     needs again for the residual in the accumulator layout of Y^T: x is read from HBM once;
//   * LayerNorm statistics: in-lane sums over the lane's 32 channels, one cross-lane add (the other half of the row sits in
//     lane ^ 32) and an 8 x 32-float exchange through LDS;
//   * backward: the per-channel sums over rows (dgamma, dbeta, db2, db1) are sums over LANES in this layout: a halving butterfly
//     (30 shuffles + one for 32 registers) leaves two finished column sums per lane; dh^T = W2^T dy^T and dx^T = ds^T + W1^T dh^T are the
//     same two-product chain as the forward with the roles of the weights exchanged.
//
// Roundings follow the autocast recipe: h and y leave their GEMMs as bf16, dropout scales in bf16, the residual sum, the
// LayerNorm and every stored tensor (s, out, hd, dy, dh, dx) are fp32.  Algorithmic HBM traffic: forward R * (E * 12 + F * 4) B
// (+ 4 B * E per emitted bf16 operand pair), backward R * (E * 20 + F * 8) B -- 25.8 / 43.2 MB at 4120 rows.
#include "pcm_attn.hpp"

#pragma clang fp contract(fast)

#ifdef PCM_FFN_CLOCKS  // tools/mb/experiments/mb_ffn_clocks.py: where does a tile's time go?  (never defined in the product build)
__device__ long long pcm_ffn_clk[2][8][16];
#define STAMP(k, i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) pcm_ffn_clk[k][threadIdx.x >> 6][i] = clock64(); } while (0)
extern "C" int pcm_ffn_clocks_read(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcm_ffn_clk), sizeof(pcm_ffn_clk)); }
#else
#define STAMP(k, i) do { } while (0)
#endif

namespace {

constexpr int kNW = 8;          // waves per workgroup = per 32-row tile: a lane's serial work (its share of one row) is E / 16 values
constexpr int kT = 64 * kNW;   // -- with 4 waves the 64 values per lane cost ~6000 instructions of ONE wave, 10 us whatever the row count
constexpr int kF = 32;         // dim_feedforward

// Dropout mask of this kernel pair (forward and backward must agree with EACH OTHER; the fp32 kernels of csrc/ffn.hip use the
// per-element double hash of pcm_elem.hpp, 25 VALU operations incl. four quarter-rate 32-bit multiplies per element -- with 32-64
// elements per lane that alone was half of this kernel's time).  Here: one row key per lane (attn_rowbase: seed, site, row), one
// xorshift-multiply-xorshift word (pcm_attn.hpp mixp, 24-bit multiply) per PAIR of adjacent channels, 16-bit thresholds -- the
// generator of the attention kernels, whose statistics tests/test_host_logic.py checks.
struct Drop {
    uint32_t key, thr;  // key already advanced to this lane's first pair
    float scale;
    bool on;
};
__device__ __forceinline__ Drop make_drop(float p, const long *seed_ptr, uint32_t site, uint32_t row, uint32_t first_pair)
{
    Drop d;
    d.on = p > 0.f;
    const uint64_t seed = d.on ? (uint64_t)seed_ptr[0] : 0ull;
    d.thr = d.on ? (uint32_t)((double)p * 65536.0 + 0.5) : 0u;
    d.scale = d.on ? 1.f / (1.f - p) : 1.f;
    d.key = attn_rowbase(seed, site, row) + first_pair * 0x9E3779B1u;
    return d;
}
// keep bits of the two channels of pair (first_pair + dpair): dpair is a compile-time constant at every call site
__device__ __forceinline__ void keep2(const Drop &d, uint32_t dpair, bool &k0, bool &k1)
{
    const uint32_t bits = mixp(d.key + dpair * 0x9E3779B1u);
    k0 = (bits & 0xFFFFu) >= d.thr, k1 = (bits >> 16) >= d.thr;
}

__device__ __forceinline__ bf8 cvt8(const float4 &a, const float4 &b)
{
    return as_bf8(make_uint4(pcm_cvt_pk_bf16(a.x, a.y), pcm_cvt_pk_bf16(a.z, a.w), pcm_cvt_pk_bf16(b.x, b.y), pcm_cvt_pk_bf16(b.z, b.w)));
}
__device__ __forceinline__ float round_bf16(float v)
{
    return __uint_as_float(pcm_cvt_pk_bf16(v, 0.f) << 16);
}

// sum of the waves' 32 x 32 accumulator tiles: lane-major 16-byte slots, conflict-free ds_write_b128 / ds_read_b128
__device__ __forceinline__ void cross_wave_sum(f16v &acc, float *red, int wave, int lane)
{
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(red + ((wave * 4 + q) * 64 + lane) * 4) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 t = *reinterpret_cast<const float4 *>(red + ((0 * 4 + q) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < kNW; ++w) {
            const float4 u = *reinterpret_cast<const float4 *>(red + ((w * 4 + q) * 64 + lane) * 4);
            t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
        }
        acc[4 * q] = t.x, acc[4 * q + 1] = t.y, acc[4 * q + 2] = t.z, acc[4 * q + 3] = t.w;
    }
}

// row sum over the waves' channel ranges: v = this lane's sum over its own channels
__device__ __forceinline__ float row_total(float v, float *slot, int wave, int n)
{
    v += __shfl_xor(v, 32);  // the other half of the wave's channels for this row
    slot[wave * 32 + n] = v;
    __syncthreads();
    float t = slot[n];
#pragma unroll
    for (int w = 1; w < kNW; ++w) t += slot[w * 32 + n];
    return t;
}

// column sums in the lane = row layout: p[j] (j = lane-local channel index) summed over the 32 lanes of this half-wave;
// returns the finished sums of channels j = 2n and 2n + 1 (n = lane & 31).  NV = 64 or 32 or 16 values per lane.
template <int NV>
__device__ __forceinline__ void column_sums(float (&p)[NV], int lane, float &s0, float &s1)
{
    static_assert(NV == 64 || NV == 32 || NV == 16, "");
    // step k exchanges with lane ^ (16 >> k) and halves the number of live values; a lane whose bit is set keeps the upper half
    if constexpr (NV == 64) {
        const bool hi = lane & 16;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float send = hi ? p[i] : p[i + 32], keep = hi ? p[i + 32] : p[i];
            p[i] = keep + __shfl_xor(send, 16);
        }
    }
    if constexpr (NV >= 32) {
        constexpr int off = NV == 64 ? 8 : 16;
        const bool hi = lane & off;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float send = hi ? p[i] : p[i + 16], keep = hi ? p[i + 16] : p[i];
            p[i] = keep + __shfl_xor(send, off);
        }
    }
    {
        constexpr int off = NV == 64 ? 4 : (NV == 32 ? 8 : 16);
        const bool hi = lane & off;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float send = hi ? p[i] : p[i + 8], keep = hi ? p[i + 8] : p[i];
            p[i] = keep + __shfl_xor(send, off);
        }
    }
    {
        constexpr int off = NV == 64 ? 2 : (NV == 32 ? 4 : 8);
        const bool hi = lane & off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = hi ? p[i] : p[i + 4], keep = hi ? p[i + 4] : p[i];
            p[i] = keep + __shfl_xor(send, off);
        }
    }
    {
        constexpr int off = NV == 64 ? 1 : (NV == 32 ? 2 : 4);
        const bool hi = lane & off;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = hi ? p[i] : p[i + 2], keep = hi ? p[i + 2] : p[i];
            p[i] = keep + __shfl_xor(send, off);
        }
    }
    // NV == 64: 32 lanes x 2 values = 64 sums, done.  NV == 32: 16 lane groups hold the 32 sums twice over (lanes n and n ^ 1 still
    // differ): one more exchange completes them; NV == 16 likewise with two more.
    if constexpr (NV == 32) {
        p[0] += __shfl_xor(p[0], 1), p[1] += __shfl_xor(p[1], 1);
    }
    if constexpr (NV == 16) {
        p[0] += __shfl_xor(p[0], 2), p[1] += __shfl_xor(p[1], 2);
        p[0] += __shfl_xor(p[0], 1), p[1] += __shfl_xor(p[1], 1);
    }
    s0 = p[0], s1 = p[1];
}
// which lane-local channel indices the two sums of `column_sums<NV>` belong to: j0, j0 + 1
template <int NV>
__device__ __forceinline__ int column_index(int n)
{
    if constexpr (NV == 64) return 2 * n;
    if constexpr (NV == 32) return 2 * (n >> 1);   // every pair of lanes holds the same two sums
    return 2 * (n >> 2);
}

template <int E>
struct Geo {
    static constexpr int CW = E / kNW;  // channels per wave
    static constexpr int S = CW / 16;   // K slabs of the first product per wave
    static constexpr int MT = CW / 32;  // 32-channel M tiles of the second product per wave
    static constexpr int NV = MT * 16;  // values per lane (its share of one row)
};

// lane-local channel index j = mt * 16 + q * 4 + i  ->  channel
template <int E>
__device__ __forceinline__ int chan(int wave, int h, int j)
{
    return wave * Geo<E>::CW + 32 * (j >> 4) + 8 * ((j >> 2) & 3) + 4 * h + (j & 3);
}

// store predicate / index of the two column sums a lane ends up with (column_sums<NV>)
template <int NV>
__device__ __forceinline__ bool owns_sums(int n)
{
    return NV == 64 || (NV == 32 && (n & 1) == 0) || (NV == 16 && (n & 3) == 0);
}

template <int E>
__global__ __launch_bounds__(kT) void pcm_ffn_ln_mfma_fwd_kernel(long R, const float *__restrict__ x, const float *__restrict__ W1,
                                                                 const float *__restrict__ b1, const float *__restrict__ W2,
                                                                 const float *__restrict__ b2, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float eps, float pa, float pb,
                                                                 const long *__restrict__ seed_ptr, unsigned site_a, unsigned site_b,
                                                                 float *__restrict__ hd_out, float *__restrict__ s_out,
                                                                 float *__restrict__ out, float *__restrict__ mean_out,
                                                                 float *__restrict__ rstd_out, const float *__restrict__ pos, long pos_rows,
                                                                 __hip_bfloat16 *__restrict__ sum16, __hip_bfloat16 *__restrict__ x16)
{
    using G = Geo<E>;
    __shared__ __attribute__((aligned(16))) float red[kNW * 4 * 64 * 4];  // 32 KiB
    __shared__ float slot[2][kNW * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, n = lane & 31;
    const long row = (long)blockIdx.x * 32 + n;
    const bool ok = row < R;
    const long rr = ok ? row : R - 1;
    const int cbase = wave * G::CW;
    const int c_lane = cbase + 4 * h;  // first channel of this lane; its channels: c_lane + 32 mt + 8 q + i
    // hidden unit j = 8 q + 4 h + i -> pair (8 q + 4 h) / 2 + {0, 1};  channel c_lane + 32 mt + 8 q + i -> pair c_lane / 2 + 16 mt + 4 q + {0, 1}
    STAMP(0, 0);
    const Drop da = make_drop(pa, seed_ptr, site_a, (uint32_t)rr, (uint32_t)(2 * h));
    const Drop db = make_drop(pb, seed_ptr, site_b, (uint32_t)rr, (uint32_t)(c_lane >> 1));

    // ---- every global read of the tile's first half is issued up front: a tile is ONE pass of straight-line code per wave, so each
    // dependent round trip to L2 / HBM (~1-2 us with so few waves in flight) would be fully exposed ---------------------------------
    float4 xv[G::S][2], w1v[G::S][2], w2v[G::MT][4], b1v[4], b2v[G::MT][4];
    const float *xrow = x + rr * E + c_lane;
    const float *w1row = W1 + n * E + c_lane;  // A operand of the first product: m = hidden unit n
#pragma unroll
    for (int s = 0; s < G::S; ++s) {
        xv[s][0] = *reinterpret_cast<const float4 *>(xrow + 16 * s);
        xv[s][1] = *reinterpret_cast<const float4 *>(xrow + 16 * s + 8);
        w1v[s][0] = *reinterpret_cast<const float4 *>(w1row + 16 * s);
        w1v[s][1] = *reinterpret_cast<const float4 *>(w1row + 16 * s + 8);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) b1v[q] = *reinterpret_cast<const float4 *>(b1 + 8 * q + 4 * h);
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        const float *w2row = W2 + (cbase + 32 * mt + n) * kF + 4 * h;  // A operand of the second product: m = output channel
#pragma unroll
        for (int q = 0; q < 4; ++q) w2v[mt][q] = *reinterpret_cast<const float4 *>(w2row + 8 * q);
    }
    // ---- H^T (partial over this wave's channels) = W1[:, range] . X[rows, range]^T ----------------------------------------
    STAMP(0, 1);
    f16v acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int s = 0; s < G::S; ++s) acc1 = PCM_MFMA16(cvt8(w1v[s][0], w1v[s][1]), cvt8(xv[s][0], xv[s][1]), acc1);
    STAMP(0, 2);
    // the second product's bias: issued before the first barrier (the registers of the W1 operands are free now)
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) b2v[mt][q] = *reinterpret_cast<const float4 *>(b2 + c_lane + 32 * mt + 8 * q);
    cross_wave_sum(acc1, red, wave, lane);
    STAMP(0, 3);
    // ---- bias, relu, dropout_a; register r = 4 q + i <-> hidden unit 8 q + 4 h + i ----------------------------------------------
    float hd[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float bb[4] = {b1v[q].x, b1v[q].y, b1v[q].z, b1v[q].w};
        bool k[4];
        keep2(da, 4 * q, k[0], k[1]);
        keep2(da, 4 * q + 1, k[2], k[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = round_bf16(acc1[4 * q + i] + bb[i]);  // linear1 leaves its GEMM as bf16 under autocast
            v = v > 0.f ? v : 0.f;
            hd[4 * q + i] = k[i] ? round_bf16(v * da.scale) : 0.f;
        }
        if (wave == 0 && ok)
            *reinterpret_cast<float4 *>(hd_out + row * kF + 8 * q + 4 * h) = make_float4(hd[4 * q], hd[4 * q + 1], hd[4 * q + 2], hd[4 * q + 3]);
    }
    const bf8 hb0 = as_bf8(make_uint4(pcm_cvt_pk_bf16(hd[0], hd[1]), pcm_cvt_pk_bf16(hd[2], hd[3]), pcm_cvt_pk_bf16(hd[4], hd[5]), pcm_cvt_pk_bf16(hd[6], hd[7])));
    const bf8 hb1 = as_bf8(make_uint4(pcm_cvt_pk_bf16(hd[8], hd[9]), pcm_cvt_pk_bf16(hd[10], hd[11]), pcm_cvt_pk_bf16(hd[12], hd[13]), pcm_cvt_pk_bf16(hd[14], hd[15])));
    STAMP(0, 4);
    // ---- Y^T = W2[range, :] . Hd^T; residual, dropout_b, LayerNorm ------------------------------------------------------------
    float sv[G::NV];
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = PCM_MFMA16(cvt8(w2v[mt][0], w2v[mt][1]), hb0, acc);
        acc = PCM_MFMA16(cvt8(w2v[mt][2], w2v[mt][3]), hb1, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = b2v[mt][q];
            const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
            const float4 xq = xv[2 * mt + (q >> 1)][q & 1];
            const float xx[4] = {xq.x, xq.y, xq.z, xq.w};
            bool k[4];
            keep2(db, 16 * mt + 4 * q, k[0], k[1]);
            keep2(db, 16 * mt + 4 * q + 1, k[2], k[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y = round_bf16(acc[4 * q + i] + bb[i]);
                const float s = xx[i] + (k[i] ? round_bf16(y * db.scale) : 0.f);
                sv[mt * 16 + 4 * q + i] = s;
                sum += s;
            }
        }
    }
    STAMP(0, 5);
    // the epilogue's operands (LayerNorm affine, position rows of the emitted operands): issued before the statistics barriers
    float4 gv[G::MT][4], tv[G::MT][4];
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gv[mt][q] = *reinterpret_cast<const float4 *>(gamma + c_lane + 32 * mt + 8 * q);
            tv[mt][q] = *reinterpret_cast<const float4 *>(beta + c_lane + 32 * mt + 8 * q);
        }
    float4 pv[G::MT][4];
    if (sum16 != nullptr) {
        const float *pl = pos + (rr % pos_rows) * E + c_lane;
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) pv[mt][q] = *reinterpret_cast<const float4 *>(pl + 32 * mt + 8 * q);
    }
    const float mu = row_total(sum, slot[0], wave, n) * (1.f / E);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < G::NV; ++j) sq += (sv[j] - mu) * (sv[j] - mu);
    const float rstd = rsqrtf(row_total(sq, slot[1], wave, n) * (1.f / E) + eps);
    STAMP(0, 6);
    if (!ok) return;
    float *sl = s_out + row * E + c_lane, *ol = out + row * E + c_lane;
    __hip_bfloat16 *s16l = sum16 != nullptr ? sum16 + row * E + c_lane : nullptr, *x16l = x16 != nullptr ? x16 + row * E + c_lane : nullptr;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = 32 * mt + 8 * q;
            const float gg[4] = {gv[mt][q].x, gv[mt][q].y, gv[mt][q].z, gv[mt][q].w}, tt[4] = {tv[mt][q].x, tv[mt][q].y, tv[mt][q].z, tv[mt][q].w};
            float o[4], s4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) s4[i] = sv[mt * 16 + 4 * q + i], o[i] = (s4[i] - mu) * rstd * gg[i] + tt[i];
            store4<float>(sl + off, s4);
            store4<float>(ol + off, o);
            if (sum16 != nullptr) {  // the next layer's in-projection operands (see pcm_drln_forward2_hip)
                const float qv[4] = {o[0] + pv[mt][q].x, o[1] + pv[mt][q].y, o[2] + pv[mt][q].z, o[3] + pv[mt][q].w};
                store4<__hip_bfloat16>(s16l + off, qv);
            }
            if (x16 != nullptr) store4<__hip_bfloat16>(x16l + off, o);
        }
    }
    if (wave == 0 && h == 0) mean_out[row] = mu, rstd_out[row] = rstd;
    STAMP(0, 7);
}

// partial layout per workgroup (= per 32-row tile): [ dgamma(E) | dbeta(E) | db2(E) | db1(F) ]
template <int E>
__global__ __launch_bounds__(kT) void pcm_ffn_ln_mfma_bwd_kernel(long R, const float *__restrict__ dout, const float *__restrict__ dout2,
                                                                 const float *__restrict__ s, const float *__restrict__ mean,
                                                                 const float *__restrict__ rstd, const float *__restrict__ hd,
                                                                 const float *__restrict__ W1, const float *__restrict__ W2,
                                                                 const float *__restrict__ gamma, float pa, float pb,
                                                                 const long *__restrict__ seed_ptr, unsigned site_b,
                                                                 float *__restrict__ dx, float *__restrict__ dy,
                                                                 float *__restrict__ dh_out, float *__restrict__ partial)
{
    using G = Geo<E>;
    constexpr int PW = 3 * E + kF;
    __shared__ __attribute__((aligned(16))) float red[kNW * 4 * 64 * 4];
    __shared__ float slot[2][kNW * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, n = lane & 31;
    const long row = (long)blockIdx.x * 32 + n;
    const bool ok = row < R;
    const long rr = ok ? row : R - 1;
    const float okf = ok ? 1.f : 0.f;
    const int cbase = wave * G::CW;
    const int c_lane = cbase + 4 * h;
    const Drop db = make_drop(pb, seed_ptr, site_b, (uint32_t)rr, (uint32_t)(c_lane >> 1));
    const float sc_a = pa > 0.f ? 1.f / (1.f - pa) : 1.f;
    float *part = partial + (size_t)blockIdx.x * PW;
    const float mu = mean[rr], rs = rstd[rr];

    // ---- every global read is issued up front (see the forward kernel) --------------------------------------------------------------
    float dv[G::NV], xh[G::NV];
    float4 gv[G::MT][4], hv[4];
    float w2e[G::S][8], w1e[G::MT][16];
    {
        const float *dl = dout + rr * E + c_lane, *d2l = dout2 != nullptr ? dout2 + rr * E + c_lane : nullptr, *sl = s + rr * E + c_lane;
        float4 d4[G::MT][4], e4[G::MT][4], s4[G::MT][4];
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = 32 * mt + 8 * q;
                d4[mt][q] = *reinterpret_cast<const float4 *>(dl + off);
                e4[mt][q] = dout2 != nullptr ? *reinterpret_cast<const float4 *>(d2l + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                s4[mt][q] = *reinterpret_cast<const float4 *>(sl + off);
                gv[mt][q] = *reinterpret_cast<const float4 *>(gamma + c_lane + off);
            }
#pragma unroll
        for (int sl2 = 0; sl2 < G::S; ++sl2) {  // A operand of dh^T: m = hidden unit n, k = the 8 channels of the slab: W2[ch][n]
            const float *w2c = W2 + (c_lane + 16 * sl2) * kF + n;
#pragma unroll
            for (int t = 0; t < 4; ++t) w2e[sl2][t] = w2c[t * kF], w2e[sl2][4 + t] = w2c[(8 + t) * kF];
        }
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dd[4] = {d4[mt][q].x + e4[mt][q].x, d4[mt][q].y + e4[mt][q].y, d4[mt][q].z + e4[mt][q].z, d4[mt][q].w + e4[mt][q].w};
                const float ss[4] = {s4[mt][q].x, s4[mt][q].y, s4[mt][q].z, s4[mt][q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) dv[mt * 16 + 4 * q + i] = dd[i] * okf, xh[mt * 16 + 4 * q + i] = (ss[i] - mu) * rs;
            }
    }
    const int j0 = column_index<G::NV>(n);
    const int ch0 = c_lane + 32 * (j0 >> 4) + 8 * ((j0 >> 2) & 3) + (j0 & 3);  // channel of column sum j0 (j0 + 1: the next channel)
    {
        float t[G::NV], a, b;
#pragma unroll
        for (int j = 0; j < G::NV; ++j) t[j] = dv[j];
        column_sums<G::NV>(t, lane, a, b);  // dbeta
        if (owns_sums<G::NV>(n)) part[E + ch0] = a, part[E + ch0 + 1] = b;
#pragma unroll
        for (int j = 0; j < G::NV; ++j) t[j] = dv[j] * xh[j];
        column_sums<G::NV>(t, lane, a, b);  // dgamma
        if (owns_sums<G::NV>(n)) part[ch0] = a, part[ch0 + 1] = b;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gg[4] = {gv[mt][q].x, gv[mt][q].y, gv[mt][q].z, gv[mt][q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = mt * 16 + 4 * q + i;
                dv[j] *= gg[i];  // gd
                s1 += dv[j], s2 += dv[j] * xh[j];
            }
        }
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {  // A operand of dx^T: m = channel, k = hidden unit: W1[hid][ch]; issued before the barriers
        const float *w1c = W1 + (4 * h) * E + cbase + 32 * mt + n;
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8)
#pragma unroll
            for (int t = 0; t < 4; ++t) w1e[mt][4 * g8 + t] = w1c[(8 * g8 + t) * E];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4 *>(hd + rr * kF + 8 * q + 4 * h);
    const float m1 = row_total(s1, slot[0], wave, n) * (1.f / E), m2 = row_total(s2, slot[1], wave, n) * (1.f / E);
    // ds (kept in dv), dy = mask_b ds
    float dyv[G::NV];
    const float rso = okf * rs;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bool k[4];
            keep2(db, 16 * mt + 4 * q, k[0], k[1]);
            keep2(db, 16 * mt + 4 * q + 1, k[2], k[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = mt * 16 + 4 * q + i;
                dv[j] = rso * (dv[j] - m1 - xh[j] * m2);
                dyv[j] = k[i] ? dv[j] * db.scale : 0.f;
            }
        }
    if (ok) {
        float *dyl = dy + row * E + c_lane;
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float o[4] = {dyv[mt * 16 + 4 * q], dyv[mt * 16 + 4 * q + 1], dyv[mt * 16 + 4 * q + 2], dyv[mt * 16 + 4 * q + 3]};
                store4<float>(dyl + 32 * mt + 8 * q, o);
            }
    }
    {
        float t[G::NV], a, b;
#pragma unroll
        for (int j = 0; j < G::NV; ++j) t[j] = dyv[j];
        column_sums<G::NV>(t, lane, a, b);  // db2
        if (owns_sums<G::NV>(n)) part[2 * E + ch0] = a, part[2 * E + ch0 + 1] = b;
    }
    // ---- dh^T (partial over this wave's channels) = W2[range, :]^T . dy^T --------------------------------------------------------
    f16v acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int sl = 0; sl < G::S; ++sl) {
        const int jb = (sl >> 1) * 16 + 8 * (sl & 1);
        const bf8 bop = as_bf8(make_uint4(pcm_cvt_pk_bf16(dyv[jb], dyv[jb + 1]), pcm_cvt_pk_bf16(dyv[jb + 2], dyv[jb + 3]),
                                          pcm_cvt_pk_bf16(dyv[jb + 4], dyv[jb + 5]), pcm_cvt_pk_bf16(dyv[jb + 6], dyv[jb + 7])));
        const float4 a0 = make_float4(w2e[sl][0], w2e[sl][1], w2e[sl][2], w2e[sl][3]);
        const float4 a1 = make_float4(w2e[sl][4], w2e[sl][5], w2e[sl][6], w2e[sl][7]);
        acc1 = PCM_MFMA16(cvt8(a0, a1), bop, acc1);
    }
    cross_wave_sum(acc1, red, wave, lane);
    float dh[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float hh[4] = {hv[q].x, hv[q].y, hv[q].z, hv[q].w};
        const float sc = okf * sc_a;
#pragma unroll
        for (int i = 0; i < 4; ++i) dh[4 * q + i] = (hh[i] > 0.f) ? acc1[4 * q + i] * sc : 0.f;  // hd > 0 <=> relu active AND kept
        if (wave == 0 && ok)
            *reinterpret_cast<float4 *>(dh_out + row * kF + 8 * q + 4 * h) = make_float4(dh[4 * q], dh[4 * q + 1], dh[4 * q + 2], dh[4 * q + 3]);
    }
    if (wave == 0) {  // db1: 16 values per lane (register r <-> hidden unit (r & 3) + 8 (r >> 2) + 4 h), summed over the 32 rows
        float t[16], a, b;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = dh[r];
        column_sums<16>(t, lane, a, b);
        if ((n & 3) == 0) {
            const int r0 = column_index<16>(n);  // registers r0, r0 + 1 (r0 even: same q, consecutive hidden units)
            part[3 * E + (r0 & 3) + 8 * (r0 >> 2) + 4 * h] = a;
            part[3 * E + (r0 & 3) + 8 * (r0 >> 2) + 4 * h + 1] = b;
        }
    }
    const bf8 db0 = as_bf8(make_uint4(pcm_cvt_pk_bf16(dh[0], dh[1]), pcm_cvt_pk_bf16(dh[2], dh[3]), pcm_cvt_pk_bf16(dh[4], dh[5]), pcm_cvt_pk_bf16(dh[6], dh[7])));
    const bf8 db1v = as_bf8(make_uint4(pcm_cvt_pk_bf16(dh[8], dh[9]), pcm_cvt_pk_bf16(dh[10], dh[11]), pcm_cvt_pk_bf16(dh[12], dh[13]), pcm_cvt_pk_bf16(dh[14], dh[15])));
    // ---- dx^T = ds^T + W1[:, range]^T . dh^T ---------------------------------------------------------------------------------------
    float *dxl = dx + rr * E + c_lane;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        const float4 a0 = make_float4(w1e[mt][0], w1e[mt][1], w1e[mt][2], w1e[mt][3]);
        const float4 a1 = make_float4(w1e[mt][4], w1e[mt][5], w1e[mt][6], w1e[mt][7]);
        const float4 a2 = make_float4(w1e[mt][8], w1e[mt][9], w1e[mt][10], w1e[mt][11]);
        const float4 a3 = make_float4(w1e[mt][12], w1e[mt][13], w1e[mt][14], w1e[mt][15]);
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = PCM_MFMA16(cvt8(a0, a1), db0, acc);
        acc = PCM_MFMA16(cvt8(a2, a3), db1v, acc);
        if (ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = mt * 16 + 4 * q;
                const float o[4] = {dv[j] + acc[4 * q], dv[j + 1] + acc[4 * q + 1], dv[j + 2] + acc[4 * q + 2], dv[j + 3] + acc[4 * q + 3]};
                store4<float>(dxl + 32 * mt + 8 * q, o);
            }
        }
    }
}

}  // namespace

extern "C" int pcm_ffn_ln_mfma_supported(int E, int F) { return (F == kF && (E == 256 || E == 512)) ? 1 : 0; }
// partial rows written by the backward kernel: one per 32-row tile
extern "C" int pcm_ffn_ln_mfma_blocks(long R) { return (int)((R + 31) / 32 > 0 ? (R + 31) / 32 : 1); }

extern "C" int pcm_ffn_ln_mfma_forward_hip(long R, int E, int F, const float *x, const float *W1, const float *b1, const float *W2,
                                           const float *b2, const float *gamma, const float *beta, float eps, float p_hidden,
                                           float p_out, const long *seed, unsigned site_a, unsigned site_b, float *hd, float *s,
                                           float *out, float *mean, float *rstd, const float *pos, long pos_n, void *sum_bf16,
                                           void *out_bf16, void *stream)
{
    if (sum_bf16 != nullptr && (pos == nullptr || pos_n <= 0 || pos_n % E || (R * E) % pos_n)) return PCM_ERR_BAD_ARG;
    if (R <= 0) return R == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (!pcm_ffn_ln_mfma_supported(E, F)) return PCM_ERR_UNSUPPORTED;
    if ((p_hidden > 0.f || p_out > 0.f) && seed == nullptr) return PCM_ERR_BAD_ARG;
    if (!x || !W1 || !b1 || !W2 || !b2 || !gamma || !beta || !hd || !s || !out || !mean || !rstd) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = pcm_ffn_ln_mfma_blocks(R);
    if (E == 512)
        hipLaunchKernelGGL(pcm_ffn_ln_mfma_fwd_kernel<512>, dim3(grid), dim3(kT), 0, st, R, x, W1, b1, W2, b2, gamma, beta, eps, p_hidden,
                           p_out, seed, site_a, site_b, hd, s, out, mean, rstd, pos, pos_n / E, (__hip_bfloat16 *)sum_bf16,
                           (__hip_bfloat16 *)out_bf16);
    else
        hipLaunchKernelGGL(pcm_ffn_ln_mfma_fwd_kernel<256>, dim3(grid), dim3(kT), 0, st, R, x, W1, b1, W2, b2, gamma, beta, eps, p_hidden,
                           p_out, seed, site_a, site_b, hd, s, out, mean, rstd, pos, pos_n / E, (__hip_bfloat16 *)sum_bf16,
                           (__hip_bfloat16 *)out_bf16);
    return PCM_LAUNCH_STATUS();
}

// `partial`: pcm_ffn_ln_mfma_blocks(R) rows of 3 E + F floats; `sums` (3 E + F) or NULL (the caller closes the reduction later,
// policy/deferred.py).  `x` is not read (kept for the argument list of pcm_ffn_ln_backward2_hip).
extern "C" int pcm_ffn_reduce_rows_hip(int nslots, int VH, const float *partial, float *out, void *stream);

extern "C" int pcm_ffn_ln_mfma_backward_hip(long R, int E, int F, const float *dout, const float *dout2, const float *x, const float *s,
                                            const float *mean, const float *rstd, const float *hd, const float *W1, const float *W2,
                                            const float *gamma, float p_hidden, float p_out, const long *seed, unsigned site_b,
                                            float *dx, float *dy, float *dh, float *partial, float *sums, void *stream)
{
    (void)x;
    if (R <= 0) return R == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (!pcm_ffn_ln_mfma_supported(E, F)) return PCM_ERR_UNSUPPORTED;
    if (p_out > 0.f && seed == nullptr) return PCM_ERR_BAD_ARG;
    if (!dout || !s || !mean || !rstd || !hd || !W1 || !W2 || !gamma || !dx || !dy || !dh || !partial) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = pcm_ffn_ln_mfma_blocks(R);
    if (E == 512)
        hipLaunchKernelGGL(pcm_ffn_ln_mfma_bwd_kernel<512>, dim3(grid), dim3(kT), 0, st, R, dout, dout2, s, mean, rstd, hd, W1, W2, gamma,
                           p_hidden, p_out, seed, site_b, dx, dy, dh, partial);
    else
        hipLaunchKernelGGL(pcm_ffn_ln_mfma_bwd_kernel<256>, dim3(grid), dim3(kT), 0, st, R, dout, dout2, s, mean, rstd, hd, W1, W2, gamma,
                           p_hidden, p_out, seed, site_b, dx, dy, dh, partial);
    int rc = PCM_LAUNCH_STATUS();
    if (rc || sums == nullptr) return rc;
    return pcm_ffn_reduce_rows_hip(grid, 3 * E + F, partial, sums, stream);
}
