// ffn_mfma.hip -- the fused feed-forward sub-layer of the ACT transformer on the matrix cores (gfx950, bf16 autocast path):
//
//     out = LayerNorm( x + dropout_b( W2 . dropout_a( relu( W1 . x + b1 ) ) + b2 ) )
//
// Same sub-layer as csrc/ffn.hip (/root/reference/src/models/components/act/transformer.py:253-256 encoder, :342-345 decoder;
// dim_feedforward = 32, configs/model/maniskill2_act_pcd_model.yaml:34), same entry-point arguments (its own, cheaper dropout generator, see Drop below).
// ffn.hip is the fp32 path: every row re-reads both weight matrices from LDS (128 KiB staged per workgroup, 540 MB of LDS reads at
// 4120 rows) -- 20 us per launch at 4120 rows, LDS-bandwidth bound, and 9 us even at 800 rows because every workgroup stages the weights.
// Under bf16 autocast the two products ARE bf16 GEMMs in the reference recipe, so here they run on v_mfma_f32_16x16x32_bf16:
//
//   * one workgroup (4 waves, one per SIMD) per 16-ROW tile; wave w owns the channels [E/4 * w, E/4 * (w+1)).  [The first version used
//     32-row tiles on v_mfma_f32_32x32x16_bf16 with 8 waves: a tile is one pass of straight-line code, ~1500 instructions per wave,
//     and lived 30-36 k clocks = 14 us whatever the row count (phase clocks: tools/mb/experiments/mb_ffn_clocks.py) -- instruction-issue
//     bound on ONE CU while 129 / 25 tiles left half / nine tenths of the chip idle.  16-row tiles halve the values per lane and
//     double the tiles: 258 at 4120 rows = every CU, 50 at 800 rows.]
//   * nothing is staged in LDS: the weights are read straight from L2 as MFMA A operands;
//   * everything is computed TRANSPOSED (lane & 15 = row of the tile, g = lane >> 4 = k-group): H^T = W1 X^T per wave over its K
//     range, the four partial 32 x 16 tiles meet in LDS (8 KiB), and -- because the K order of an MFMA is free as long as A and B
//     agree -- the accumulator layout of H^T (lane (g, n): hidden units {4g..4g+3, 16+4g..16+4g+3} of row n) IS the B-operand layout
//     of the second product Y^T = W2 H^T: the hidden activations never leave registers;
//   * the same K permutation is applied to the first product (k-slot (g, i) of 32-channel step s <-> channel 32 s + 16 (i >> 2) + 4 g
//     + (i & 3)), so the x values a lane loaded as B operand are exactly the ones it needs again for the residual in the accumulator
//     layout of Y^T (M-tile mt = channels 16 mt .. 16 mt + 15: lane (g, n) holds channels 16 mt + 4 g + i of row n): x is read once;
//   * LayerNorm statistics: in-lane sums over the lane's channels, two cross-lane adds (the other k-groups of the row sit in
//     lane ^ 16, lane ^ 32) and a 4 x 16-float exchange through LDS;
//   * backward: the per-channel sums over rows (dgamma, dbeta, db2, db1) are sums over the 16 LANES n of a k-group in this layout: a
//     halving butterfly (4 steps) leaves two finished column sums per lane; dh^T = W2^T dy^T and dx^T = ds^T + W1^T dh^T are the
//     same two-product chain as the forward with the roles of the weights exchanged (their A operands are 8 strided loads each).
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

constexpr int kNW = 4;          // waves per workgroup = per 16-row tile, one per SIMD
constexpr int kT = 64 * kNW;
constexpr int kF = 32;          // dim_feedforward
constexpr int kTR = 16;         // rows per tile

typedef float f4v __attribute__((ext_vector_type(4)));
#define PCM_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)  // A: lane (m = l & 15, k = 8 (l >> 4) ..+7); C: col l & 15, rows 4 (l >> 4) + r

// Dropout mask of this kernel pair (forward and backward must agree with EACH OTHER; the fp32 kernels of csrc/ffn.hip use the
// per-element double hash of pcm_elem.hpp, 25 VALU operations incl. four quarter-rate 32-bit multiplies per element).  Here: one
// row key per lane (attn_rowbase: seed, site, row), one xorshift-multiply-xorshift word (pcm_attn.hpp mixp, 24-bit multiply) per
// PAIR of adjacent channels, 16-bit thresholds -- the generator of the attention kernels, whose statistics
// tests/test_host_logic.py checks.  The mask of (row, channel) does not depend on the tiling.
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
__device__ __forceinline__ bf8 cvt8(const float (&v)[8])
{
    return as_bf8(make_uint4(pcm_cvt_pk_bf16(v[0], v[1]), pcm_cvt_pk_bf16(v[2], v[3]), pcm_cvt_pk_bf16(v[4], v[5]), pcm_cvt_pk_bf16(v[6], v[7])));
}
__device__ __forceinline__ float round_bf16(float v)
{
    return __uint_as_float(pcm_cvt_pk_bf16(v, 0.f) << 16);
}
__device__ __forceinline__ void unpack(const float4 &v, float (&o)[4])
{
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
}

// sum of the waves' two 16 x 16 accumulator tiles (hidden units 0..15 | 16..31): lane-major 16-byte slots
__device__ __forceinline__ void cross_wave_sum(f4v (&acc)[2], float *red, int wave, int lane)
{
#pragma unroll
    for (int t = 0; t < 2; ++t)
        *reinterpret_cast<float4 *>(red + ((wave * 2 + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float4 s = *reinterpret_cast<const float4 *>(red + ((0 * 2 + t) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < kNW; ++w) {
            const float4 u = *reinterpret_cast<const float4 *>(red + ((w * 2 + t) * 64 + lane) * 4);
            s.x += u.x, s.y += u.y, s.z += u.z, s.w += u.w;
        }
        acc[t][0] = s.x, acc[t][1] = s.y, acc[t][2] = s.z, acc[t][3] = s.w;
    }
}

// row sum over the waves' channel ranges: v = this lane's sum over its own channels
__device__ __forceinline__ float row_total(float v, float *slot, int wave, int n)
{
    v += __shfl_xor(v, 16);  // the other k-groups of the wave's channels for this row
    v += __shfl_xor(v, 32);
    slot[wave * kTR + n] = v;
    __syncthreads();
    float t = slot[n];
#pragma unroll
    for (int w = 1; w < kNW; ++w) t += slot[w * kTR + n];
    return t;
}

// column sums in the lane & 15 = row layout: p[j] (j = lane-local value index) summed over the 16 lanes of this k-group.  Each halving
// step exchanges with lane ^ off and keeps the half of the values that the lane's bit selects, so the value index left in lane n has
// n's bits (3, 2, 1, 0) as its top bits.  NV = 32: two finished sums per lane, j = 2 n, 2 n + 1 (returned in s0, s1); NV = 16: one,
// j = n (s0); NV = 8: one per lane PAIR, j = n >> 1 (s0; lanes n and n ^ 1 hold the same sum).
template <int HALF, int OFF, int N>
__device__ __forceinline__ void halve_step(float (&p)[N], int lane)
{
    const bool hi = lane & OFF;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float send = hi ? p[i] : p[i + HALF], keep = hi ? p[i + HALF] : p[i];
        p[i] = keep + __shfl_xor(send, OFF);
    }
}
template <int NV>
__device__ __forceinline__ void column_sums16(float (&p)[NV], int lane, float &s0, float &s1)
{
    static_assert(NV == 32 || NV == 16 || NV == 8, "");
    if constexpr (NV == 32) {
        halve_step<16, 8>(p, lane), halve_step<8, 4>(p, lane), halve_step<4, 2>(p, lane), halve_step<2, 1>(p, lane);
        s0 = p[0], s1 = p[1];
    } else if constexpr (NV == 16) {
        halve_step<8, 8>(p, lane), halve_step<4, 4>(p, lane), halve_step<2, 2>(p, lane), halve_step<1, 1>(p, lane);
        s0 = p[0], s1 = 0.f;
    } else {
        halve_step<4, 8>(p, lane), halve_step<2, 4>(p, lane), halve_step<1, 2>(p, lane);
        p[0] += __shfl_xor(p[0], 1);
        s0 = p[0], s1 = 0.f;
    }
}

template <int E>
struct Geo {
    static constexpr int CW = E / kNW;  // channels per wave
    static constexpr int KS = CW / 32;  // K steps of the first product per wave
    static constexpr int MT = CW / 16;  // 16-channel M tiles of the second product per wave
    static constexpr int NV = MT * 4;   // values per lane (its share of one row): local index j = 4 mt + i <-> channel 16 mt + 4 g + i
};

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
    __shared__ __attribute__((aligned(16))) float red[kNW * 2 * 64 * 4];  // 8 KiB
    __shared__ float slot[2][kNW * kTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, n = lane & 15;
#ifdef PCM_FFN_CLOCKS  // the probe launches HALF the grid and runs two tiles per workgroup: is the second pass (warm code, cold data) faster?
    for (int it = 0; it < 2; ++it) {
    const long tile = blockIdx.x + (long)it * gridDim.x;
    if (tile * kTR >= R) break;
#else
    {
    const long tile = blockIdx.x;
    constexpr int it = 0;
#endif
    const long row = tile * kTR + n;
    const bool ok = row < R;
    const long rr = ok ? row : R - 1;
    const int c_lane = wave * G::CW + 4 * g;  // first channel of this lane; its channels: c_lane + 16 mt + i
    STAMP(it, 0);
    // ---- every global read of the tile's first half is issued up front: a tile is ONE pass of straight-line code per wave, so each
    // dependent round trip to L2 / HBM would be fully exposed ------------------------------------------------------------------------
    float4 xv[G::MT], w1v[2][G::MT], b1v[2];
    const float *xrow = x + rr * E + c_lane;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) xv[mt] = *reinterpret_cast<const float4 *>(xrow + 16 * mt);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float *w1row = W1 + (16 * t + n) * E + c_lane;  // A operand of the first product: m = hidden unit 16 t + n
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt) w1v[t][mt] = *reinterpret_cast<const float4 *>(w1row + 16 * mt);
        b1v[t] = *reinterpret_cast<const float4 *>(b1 + 16 * t + 4 * g);
    }
    // hidden unit 16 t + 4 g + i -> pair 8 t + 2 g + (i >> 1);  channel c_lane + 16 mt + i -> pair c_lane / 2 + 8 mt + (i >> 1)
    const Drop da = make_drop(pa, seed_ptr, site_a, (uint32_t)rr, (uint32_t)(2 * g));
    const Drop db = make_drop(pb, seed_ptr, site_b, (uint32_t)rr, (uint32_t)(c_lane >> 1));
    STAMP(it, 1);
    // ---- H^T (partial over this wave's channels) = W1[:, range] . X[rows, range]^T ----------------------------------------
    f4v acc1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
        const bf8 xb = cvt8(xv[2 * s], xv[2 * s + 1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc1[t] = PCM_MFMA32(cvt8(w1v[t][2 * s], w1v[t][2 * s + 1]), xb, acc1[t]);
    }
    STAMP(it, 2);
    // the second product's operands: issued before the first barrier (the registers of the W1 operands are free now)
    float4 w2v[G::MT][2], b2v[G::MT];
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        const float *w2row = W2 + (wave * G::CW + 16 * mt + n) * kF + 4 * g;  // A operand of the second product: m = output channel
        w2v[mt][0] = *reinterpret_cast<const float4 *>(w2row), w2v[mt][1] = *reinterpret_cast<const float4 *>(w2row + 16);
        b2v[mt] = *reinterpret_cast<const float4 *>(b2 + c_lane + 16 * mt);
    }
    cross_wave_sum(acc1, red, wave, lane);
    STAMP(it, 3);
    // ---- bias, relu, dropout_a; value (t, i) <-> hidden unit 16 t + 4 g + i -------------------------------------------------------
    float hd[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float bb[4];
        unpack(b1v[t], bb);
        bool k[4];
        keep2(da, 8 * t, k[0], k[1]);
        keep2(da, 8 * t + 1, k[2], k[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = round_bf16(acc1[t][i] + bb[i]);  // linear1 leaves its GEMM as bf16 under autocast
            v = v > 0.f ? v : 0.f;
            hd[4 * t + i] = k[i] ? round_bf16(v * da.scale) : 0.f;
        }
        if (wave == 0 && ok) *reinterpret_cast<float4 *>(hd_out + row * kF + 16 * t + 4 * g) = make_float4(hd[4 * t], hd[4 * t + 1], hd[4 * t + 2], hd[4 * t + 3]);
    }
    const bf8 hb = cvt8(hd);
    STAMP(it, 4);
    // ---- Y^T = W2[range, :] . Hd^T; residual, dropout_b, LayerNorm ------------------------------------------------------------
    float sv[G::NV];
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        f4v acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = 0.f;
        acc = PCM_MFMA32(cvt8(w2v[mt][0], w2v[mt][1]), hb, acc);
        float bb[4], xx[4];
        unpack(b2v[mt], bb);
        unpack(xv[mt], xx);
        bool k[4];
        keep2(db, 8 * mt, k[0], k[1]);
        keep2(db, 8 * mt + 1, k[2], k[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float y = round_bf16(acc[i] + bb[i]);
            const float s = xx[i] + (k[i] ? round_bf16(y * db.scale) : 0.f);
            sv[mt * 4 + i] = s;
            sum += s;
        }
    }
    STAMP(it, 5);
    // the epilogue's operands (LayerNorm affine, position rows of the emitted operands): issued before the statistics barriers
    float4 gv[G::MT], tv[G::MT], pv[G::MT];
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        gv[mt] = *reinterpret_cast<const float4 *>(gamma + c_lane + 16 * mt);
        tv[mt] = *reinterpret_cast<const float4 *>(beta + c_lane + 16 * mt);
    }
    if (sum16 != nullptr) {
        const float *pl = pos + (rr % pos_rows) * E + c_lane;
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt) pv[mt] = *reinterpret_cast<const float4 *>(pl + 16 * mt);
    }
    const float mu = row_total(sum, slot[0], wave, n) * (1.f / E);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < G::NV; ++j) sq += (sv[j] - mu) * (sv[j] - mu);
    const float rstd = rsqrtf(row_total(sq, slot[1], wave, n) * (1.f / E) + eps);
    STAMP(it, 6);
    if (ok) {
    float *sl = s_out + row * E + c_lane, *ol = out + row * E + c_lane;
    __hip_bfloat16 *s16l = sum16 != nullptr ? sum16 + row * E + c_lane : nullptr, *x16l = x16 != nullptr ? x16 + row * E + c_lane : nullptr;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        float gg[4], tt[4], o[4], s4[4];
        unpack(gv[mt], gg);
        unpack(tv[mt], tt);
#pragma unroll
        for (int i = 0; i < 4; ++i) s4[i] = sv[mt * 4 + i], o[i] = (s4[i] - mu) * rstd * gg[i] + tt[i];
        store4<float>(sl + 16 * mt, s4);
        store4<float>(ol + 16 * mt, o);
        if (sum16 != nullptr) {  // the next layer's in-projection operands (see pcm_drln_forward2_hip)
            const float qv[4] = {o[0] + pv[mt].x, o[1] + pv[mt].y, o[2] + pv[mt].z, o[3] + pv[mt].w};
            store4<__hip_bfloat16>(s16l + 16 * mt, qv);
        }
        if (x16 != nullptr) store4<__hip_bfloat16>(x16l + 16 * mt, o);
    }
    if (wave == 0 && g == 0) mean_out[row] = mu, rstd_out[row] = rstd;
    }
    STAMP(it, 7);
    (void)it;
    }
}

// partial layout per workgroup (= per 16-row tile): [ dgamma(E) | dbeta(E) | db2(E) | db1(F) ]
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
    __shared__ __attribute__((aligned(16))) float red[kNW * 2 * 64 * 4];
    __shared__ float slot[2][kNW * kTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, n = lane & 15;
    const long row = (long)blockIdx.x * kTR + n;
    const bool ok = row < R;
    const long rr = ok ? row : R - 1;
    const float okf = ok ? 1.f : 0.f;
    const int cbase = wave * G::CW;
    const int c_lane = cbase + 4 * g;
    const Drop db = make_drop(pb, seed_ptr, site_b, (uint32_t)rr, (uint32_t)(c_lane >> 1));
    const float sc_a = pa > 0.f ? 1.f / (1.f - pa) : 1.f;
    float *part = partial + (size_t)blockIdx.x * PW;
    const float mu = mean[rr], rs = rstd[rr];

    // ---- every global read is issued up front (see the forward kernel) --------------------------------------------------------------
    float dv[G::NV], xh[G::NV];
    float4 gv[G::MT], hv[2];
    float w2e[G::KS][2][8], w1e[G::MT][8];
    {
        const float *dl = dout + rr * E + c_lane, *d2l = dout2 != nullptr ? dout2 + rr * E + c_lane : nullptr, *sl = s + rr * E + c_lane;
        float4 d4[G::MT], e4[G::MT], s4[G::MT];
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt) {
            d4[mt] = *reinterpret_cast<const float4 *>(dl + 16 * mt);
            e4[mt] = dout2 != nullptr ? *reinterpret_cast<const float4 *>(d2l + 16 * mt) : make_float4(0.f, 0.f, 0.f, 0.f);
            s4[mt] = *reinterpret_cast<const float4 *>(sl + 16 * mt);
            gv[mt] = *reinterpret_cast<const float4 *>(gamma + c_lane + 16 * mt);
        }
        // A operand of dh^T: m = hidden unit 16 t + n, k-slot (g, i) of step sl2 = channel cbase + 32 sl2 + 16 (i >> 2) + 4 g + (i & 3): W2[ch][m]
#pragma unroll
        for (int sl2 = 0; sl2 < G::KS; ++sl2)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float *w2c = W2 + (c_lane + 32 * sl2) * kF + 16 * t + n;
#pragma unroll
                for (int i = 0; i < 4; ++i) w2e[sl2][t][i] = w2c[i * kF], w2e[sl2][t][4 + i] = w2c[(16 + i) * kF];
            }
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt) {
            float dd[4], ee[4], ss[4];
            unpack(d4[mt], dd);
            unpack(e4[mt], ee);
            unpack(s4[mt], ss);
#pragma unroll
            for (int i = 0; i < 4; ++i) dv[mt * 4 + i] = (dd[i] + ee[i]) * okf, xh[mt * 4 + i] = (ss[i] - mu) * rs;
        }
    }
    // the two column sums a lane ends up with (column_sums16): local indices j0, j0 + 1 -> channels ch0, ch0 + 1
    const int j0 = G::NV == 32 ? 2 * n : n;
    const int ch0 = c_lane + 16 * (j0 >> 2) + (j0 & 3);
    {
        float t[G::NV], a, b;
#pragma unroll
        for (int j = 0; j < G::NV; ++j) t[j] = dv[j];
        column_sums16<G::NV>(t, lane, a, b);  // dbeta
        part[E + ch0] = a;
        if (G::NV == 32) part[E + ch0 + 1] = b;
#pragma unroll
        for (int j = 0; j < G::NV; ++j) t[j] = dv[j] * xh[j];
        column_sums16<G::NV>(t, lane, a, b);  // dgamma
        part[ch0] = a;
        if (G::NV == 32) part[ch0 + 1] = b;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        float gg[4];
        unpack(gv[mt], gg);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = mt * 4 + i;
            dv[j] *= gg[i];  // gd
            s1 += dv[j], s2 += dv[j] * xh[j];
        }
    }
    // A operand of dx^T: m = channel cbase + 16 mt + n, k-slot (g, i) = hidden unit 16 (i >> 2) + 4 g + (i & 3): W1[hid][ch]; issued before the barriers
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        const float *w1c = W1 + (4 * g) * E + cbase + 16 * mt + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) w1e[mt][i] = w1c[i * E], w1e[mt][4 + i] = w1c[(16 + i) * E];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) hv[t] = *reinterpret_cast<const float4 *>(hd + rr * kF + 16 * t + 4 * g);
    const float m1 = row_total(s1, slot[0], wave, n) * (1.f / E), m2 = row_total(s2, slot[1], wave, n) * (1.f / E);
    // ds (kept in dv), dy = mask_b ds
    float dyv[G::NV];
    const float rso = okf * rs;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        bool k[4];
        keep2(db, 8 * mt, k[0], k[1]);
        keep2(db, 8 * mt + 1, k[2], k[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = mt * 4 + i;
            dv[j] = rso * (dv[j] - m1 - xh[j] * m2);
            dyv[j] = k[i] ? dv[j] * db.scale : 0.f;
        }
    }
    if (ok) {
        float *dyl = dy + row * E + c_lane;
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt) {
            const float o[4] = {dyv[mt * 4], dyv[mt * 4 + 1], dyv[mt * 4 + 2], dyv[mt * 4 + 3]};
            store4<float>(dyl + 16 * mt, o);
        }
    }
    {
        float t[G::NV], a, b;
#pragma unroll
        for (int j = 0; j < G::NV; ++j) t[j] = dyv[j];
        column_sums16<G::NV>(t, lane, a, b);  // db2
        part[2 * E + ch0] = a;
        if (G::NV == 32) part[2 * E + ch0 + 1] = b;
    }
    // ---- dh^T (partial over this wave's channels) = W2[range, :]^T . dy^T --------------------------------------------------------
    f4v acc1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[t][r] = 0.f;
#pragma unroll
    for (int sl = 0; sl < G::KS; ++sl) {
        float bo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bo[i] = dyv[8 * sl + i];  // M-tiles 2 sl, 2 sl + 1 = k-slots 0..3, 4..7 of this lane
        const bf8 bop = cvt8(bo);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc1[t] = PCM_MFMA32(cvt8(w2e[sl][t]), bop, acc1[t]);
    }
    cross_wave_sum(acc1, red, wave, lane);
    float dh[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float hh[4];
        unpack(hv[t], hh);
        const float sc = okf * sc_a;
#pragma unroll
        for (int i = 0; i < 4; ++i) dh[4 * t + i] = (hh[i] > 0.f) ? acc1[t][i] * sc : 0.f;  // hd > 0 <=> relu active AND kept
        if (wave == 0 && ok) *reinterpret_cast<float4 *>(dh_out + row * kF + 16 * t + 4 * g) = make_float4(dh[4 * t], dh[4 * t + 1], dh[4 * t + 2], dh[4 * t + 3]);
    }
    if (wave == 0) {  // db1: value (t, i) <-> hidden unit 16 t + 4 g + i, summed over the 16 rows; the sum of local index j = n >> 1 lands in lanes n, n ^ 1
        float t8[8], a, b;
#pragma unroll
        for (int r = 0; r < 8; ++r) t8[r] = dh[r];
        column_sums16<8>(t8, lane, a, b);
        if ((n & 1) == 0) {
            const int j = n >> 1;
            part[3 * E + 16 * (j >> 2) + 4 * g + (j & 3)] = a;
        }
    }
    const bf8 dhb = cvt8(dh);
    // ---- dx^T = ds^T + W1[:, range]^T . dh^T ---------------------------------------------------------------------------------------
    float *dxl = dx + rr * E + c_lane;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
        f4v acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = 0.f;
        acc = PCM_MFMA32(cvt8(w1e[mt]), dhb, acc);
        if (ok) {
            const int j = mt * 4;
            const float o[4] = {dv[j] + acc[0], dv[j + 1] + acc[1], dv[j + 2] + acc[2], dv[j + 3] + acc[3]};
            store4<float>(dxl + 16 * mt, o);
        }
    }
}

}  // namespace

extern "C" int pcm_ffn_ln_mfma_supported(int E, int F) { return (F == kF && (E == 256 || E == 512)) ? 1 : 0; }
// partial rows written by the backward kernel: one per 16-row tile
extern "C" int pcm_ffn_ln_mfma_blocks(long R) { return (int)((R + kTR - 1) / kTR > 0 ? (R + kTR - 1) / kTR : 1); }

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
#ifdef PCM_FFN_CLOCKS
    const int grid = (pcm_ffn_ln_mfma_blocks(R) + 1) / 2;
#else
    const int grid = pcm_ffn_ln_mfma_blocks(R);
#endif
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
