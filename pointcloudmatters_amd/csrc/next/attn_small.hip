// attn_small.hip -- multi-head attention for short sequences (head_dim 64; tuned for <= ~600 tokens), forward and
// backward, on the gfx950 matrix cores.
//
// 18 of the 22 attention calls of an ACT training step have 100-102 queries: the CVAE encoder (102 tokens), the
// decoder's self-attention (100 queries) and its cross-attention (100 queries x 515 memory tokens)
// (/root/reference/src/models/components/act/transformer.py:225-262, 296-346 via nn.MultiheadAttention).  The
// framework's flash kernels are sized for long sequences and spend 25 us forward / ~105 us backward on each of them;
// the arithmetic is 0.16 GFLOP.  Here a workgroup owns 32 query columns of one (batch, head) -- 256+ workgroups -- and
// its 4 waves split the key tiles (32 keys each, staged in wave-private LDS, next tile prefetched in registers) with
// private online-softmax states that are merged once at the end; everything on v_mfma_f32_32x32x8_bf16_1k.
//
// Layout trick (no LDS round trip between the two GEMMs of a tile): scores are computed TRANSPOSED,
//   S^T (keys x queries) = K_tile (A: lane = key row) . Q^T (B: lane = query column),
// whose accumulator layout -- lane = query column, registers = key rows {8g + 4*(lane>>5) + i} -- is exactly the B
// operand layout of the next MFMA over key slab g:  O^T (d x queries) += V^T (A: lane = d row) . P^T (B).
// A lane therefore owns ONE query for the whole kernel: the online-softmax state (max, sum) and the rescale are
// per-lane scalars.  Backward (one launch) uses the same trick twice: role-A workgroups (32 queries each) produce dQ,
// role-B workgroups (32 keys each, scores un-transposed so that lane = key column) produce dK and dV; no atomics, no
// cross-wave reductions.
//
// Dropout on the attention weights: counter-based hash of (device seed, call site, (b, h, q, key)) as in drln.hip,
// recomputed in backward.  key_padding_mask: (B, S) bytes, non-zero = ignore.  P and dS are rounded to bf16 for the
// second GEMM like every flash implementation.  q, k, v: bf16 with arbitrary batch / row strides and unit stride
// along the 64 head channels (head h at column h*64); out, dout: (B, L, H*64) contiguous; lse: (B, H, L) fp32.
#include "pcm_attn.hpp"

namespace {

constexpr int HD = 64;    // head dim
constexpr int KT = 32;    // keys (or queries) per tile = one MFMA tile edge
constexpr int LMAX = 128; // queries staged in LDS at a time by the dK/dV role
constexpr int RS = 72;    // row stride (u16) of the row-major (32 x 64) tiles: 144 B (16-byte aligned rows)
constexpr int NW = 4;     // waves per workgroup; they split the streamed dimension and meet once, at the end
constexpr int WG = 64 * NW;
constexpr int TILE_U16 = KT * RS;

// 4 elements of one column (rows r .. r+3 of a row-major tile), `p` = &tile[r][lane & 31 (+32)]: ONE ds_read_b64_tr_b16.
// Inside each 16-lane group lane i (= lane & 15) supplies the address of 4 contiguous elements of row r + i/4 at columns
// 4 (i % 4) .. +3 of the group's 16 columns; the hardware hands lane i the column i of that 4 x 16 block (semantics pinned
// by tools/mb/tr_probe.hip).  It replaces four 2-byte reads and their packing.
typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;
__device__ __forceinline__ s4 lds_col4(const u16 *p)
{
    const int i = threadIdx.x & 15;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p - i + (i >> 2) * RS + 4 * (i & 3)));
}
// one (32 x 64) bf16 tile = 256 chunks of 16 B, 4 per lane of ONE wave: chunk c = lane + 64*i -> row c>>3, cols (c&7)*8..
struct TileRegs {
    uint4 v[4];
};
__device__ __forceinline__ void tile_fetch(TileRegs &t, const u16 *__restrict__ base, long row_stride, int row0, int nrows, int lane)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i, r = c >> 3, d0 = (c & 7) * 8;
        t.v[i] = (row0 + r < nrows) ? *reinterpret_cast<const uint4 *>(base + (long)(row0 + r) * row_stride + d0) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void tile_store(const TileRegs &t, u16 *rowmajor, int lane)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i, r = c >> 3, d0 = (c & 7) * 8;
        uint2 *dst = reinterpret_cast<uint2 *>(rowmajor + r * RS + d0);
        dst[0] = make_uint2(t.v[i].x, t.v[i].y);
        dst[1] = make_uint2(t.v[i].z, t.v[i].w);
    }
}

// write one wave's (64 x 32) fp32 accumulator pair into the combine buffer part[64][32] (row = channel, column = lane&31)
__device__ __forceinline__ void spill_acc(float *part, const f16v &a0, const f16v &a1, int lane)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        part[crow(r, lane) * 32 + (lane & 31)] = a0[r];
        part[(32 + crow(r, lane)) * 32 + (lane & 31)] = a1[r];
    }
}

// one (32 keys x 32 queries) step of the forward pass for the wave's query columns: scores, online softmax, dropout, PV.
// v_mfma_f32_32x32x16_bf16: a lane holds 8 consecutive k per operand.  For S^T = K Q^T both operands are natural 16-byte
// reads.  For O^T += V^T P^T the B operand comes straight from the S^T accumulator: registers 8j..8j+7 of lane-half h are
// keys {16j+4h+i} and {16j+8+4h+i}; the A operand (V^T) is gathered with the SAME key order, so the k-sum is unchanged.
template <bool DROP>
__device__ __forceinline__ void fwd_tile(const u16 *Ks, const u16 *Vs, const bf8 (&qf)[4], f16v &o0, f16v &o1, float &m, float &lsum,
                                         int kt, int lane, const AttnParams &P, const unsigned char *mask, const DropCfg &dc, uint32_t rb)
{
    const int h8 = 8 * (lane >> 5), h4 = 4 * (lane >> 5);
    // the tile's 32 mask bytes: ONE load (lane t and t + 32 read key t), turned into a wave mask behind the matrix instructions below; the
    // 16 short-circuit conditions used to load `mask[key]` one by one, each with a full wait (tile_visible, pcm_attn.hpp)
    const unsigned mbyte = tile_mask_byte(mask, kt * KT - (lane & 32), P.S, lane);
    // every LDS operand of this tile is requested up front: the reads drain while the matrix cores and the softmax run
    bf8 ka[4], va0[2], va1[2];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) ka[sl] = lds_bf8(Ks + (lane & 31) * RS + sl * 16 + h8);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const u16 *lo = Vs + (16 * j + h4) * RS + (lane & 31), *hi = lo + 8 * RS;  // V^T: lane = channel, keys 16j+4h.. and 16j+8+4h..
        va0[j] = cat8(lds_col4(lo), lds_col4(hi));
        va1[j] = cat8(lds_col4(lo + 32), lds_col4(hi + 32));
    }
    f16v s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) s = PCM_MFMA16(ka[sl], qf[sl], s);
    const bool edge = (kt + 1) * KT > P.S || mask != nullptr;
    const float scale2 = P.scale * 1.44269504088896f;  // scores in the log2 domain: the exponentials are bare v_exp_f32
    float tmax = -INFINITY;
    const lanemask visb = edge ? tile_visible(mbyte, kt * KT - (lane & 32), P.S, lane) >> h4 : ~0ull;  // bit crow(r, 0) <-> this lane's row r
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s[r] *= scale2;
        if (edge) s[r] = ((visb >> ((r & 3) + 8 * (r >> 2))) & 1ull) ? s[r] : -INFINITY;
        tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m, tmax);
    const bool dead = m_new == -INFINITY;  // nothing visible yet for this query
    float p[16], psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        p[r] = dead ? 0.f : __builtin_amdgcn_exp2f(s[r] - m_new);
        psum += p[r];
    }
    psum += __shfl_xor(psum, 32);
    if (__any(m_new != m)) {  // the running maximum moves in the first tiles only: skip the 32 rescales afterwards
        const float alpha = (dead || m == -INFINITY) ? (dead ? 1.f : 0.f) : __builtin_amdgcn_exp2f(m - m_new);
        lsum *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) o0[r] *= alpha, o1[r] *= alpha;
    }
    lsum += psum;
    m = m_new;
    if (DROP) {
#pragma unroll
        for (int gh = 0; gh < 8; ++gh) {  // registers 2gh, 2gh+1 hold adjacent keys: one hash for the pair
            const uint32_t bits = attn_pair_bits(rb, (uint32_t)(kt * (KT / 2) + 4 * (gh >> 1) + 2 * (lane >> 5) + (gh & 1)));
            lanemask k0, k1;
            keep_masks(bits, dc.thr, k0, k1);
            p[2 * gh] = keep_if(k0, p[2 * gh] * dc.inv_keep), p[2 * gh + 1] = keep_if(k1, p[2 * gh + 1] * dc.inv_keep);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bf8 pf = cat8(pack4(p[8 * j], p[8 * j + 1], p[8 * j + 2], p[8 * j + 3]),
                            pack4(p[8 * j + 4], p[8 * j + 5], p[8 * j + 6], p[8 * j + 7]));
        o0 = PCM_MFMA16(va0[j], pf, o0);
        o1 = PCM_MFMA16(va1[j], pf, o1);
    }
}

// ------------------------------------------------------------------------------------------------ forward
// grid (B*H, ceil(L/32)), 4 waves: all own the SAME 32 query columns; wave j takes key tiles j, j+4, ... with its own
// online-softmax state and LDS tiles (no barrier in the loop); the four partial (m, l, O) meet once at the end.
template <bool DROP>  // dropout on / off is compiled in (a run-time flag leaves branches around the masked elements)
__global__ __launch_bounds__(WG) void pcm_attn_small_fwd_kernel(AttnParams P, u16 *__restrict__ out, float *__restrict__ lse)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[NW * 2 * TILE_U16 * 2 > NW * (64 * 32 + 64) * 4 ? NW * 2 * TILE_U16 * 2
                                                                                                         : NW * (64 * 32 + 64) * 4];
    asm volatile("" ::"s"(P.q), "s"(P.k), "s"(P.v), "s"(P.q_bs), "s"(P.q_ls), "s"(P.k_bs), "s"(P.k_ls), "s"(P.v_bs), "s"(P.v_ls), "s"(P.kpm), "s"(P.B), "s"(P.H), "s"(P.L), "s"(P.S), "s"(P.scale), "s"(P.p_drop), "s"(P.seed), "s"(P.site), "s"(out), "s"(lse));  // "Kernel heads", pcm_common.hpp
    const int bh = blockIdx.x, b = bh / P.H, h = bh % P.H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u16 *Ks = reinterpret_cast<u16 *>(smem) + w * 2 * TILE_U16, *Vs = Ks + TILE_U16;
    const int qi = blockIdx.y * 32 + (lane & 31);
    const bool qok = qi < P.L;
    bf8 qf[4];
    {
        const u16 *qp = P.q + (long)b * P.q_bs + (long)qi * P.q_ls + h * HD + 8 * (lane >> 5);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) qf[sl] = as_bf8(qok ? *reinterpret_cast<const uint4 *>(qp + sl * 16) : make_uint4(0, 0, 0, 0));
    }
    const DropCfg dc(P);
    const uint32_t rb = DROP ? attn_rowbase(dc.seed, P.site, (uint32_t)(bh * P.L + qi)) : 0u;
    f16v o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = 0.f, o1[r] = 0.f;
    float m = -INFINITY, lsum = 0.f;
    const u16 *kb = P.k + (long)b * P.k_bs + h * HD;
    const u16 *vb = P.v + (long)b * P.v_bs + h * HD;
    const unsigned char *mask = P.kpm ? P.kpm + (long)b * P.S : nullptr;
    const int ntiles = (P.S + KT - 1) / KT;
    TileRegs kr, vr;
    if (w < ntiles) {
        tile_fetch(kr, kb, P.k_ls, w * KT, P.S, lane);
        tile_fetch(vr, vb, P.v_ls, w * KT, P.S, lane);
    }
    for (int kt = w; kt < ntiles; kt += NW) {
        tile_store(kr, Ks, lane);  // one wave owns these tiles: its LDS operations execute in order, no barrier needed
        tile_store(vr, Vs, lane);
        __builtin_amdgcn_wave_barrier();  // the wave reads this tile back below: one wave's LDS operations execute in order on the hardware; the statement keeps the compiler (and the host model, tests/wavesim) to it -- emits no instruction
        if (kt + NW < ntiles) {
            tile_fetch(kr, kb, P.k_ls, (kt + NW) * KT, P.S, lane);
            tile_fetch(vr, vb, P.v_ls, (kt + NW) * KT, P.S, lane);
        }
        fwd_tile<DROP>(Ks, Vs, qf, o0, o1, m, lsum, kt, lane, P, mask, dc, rb);
    }
    // ---- the four partial results meet: part[w] = O_w (64 x 32), ml[w] = (m_w, l_w) per query
    __syncthreads();
    float *part = reinterpret_cast<float *>(smem), *ml = part + NW * 64 * 32;
    spill_acc(part + w * 64 * 32, o0, o1, lane);
    if (lane < 32) ml[w * 64 + lane] = m, ml[w * 64 + 32 + lane] = lsum;
    __syncthreads();
    const int q = threadIdx.x & 31, dg = threadIdx.x >> 5;  // 8 channels per thread
    const int qq = blockIdx.y * 32 + q;
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < NW; ++j) mt = fmaxf(mt, ml[j * 64 + q]);
    float sc[NW], lt = 0.f;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const float mj = ml[j * 64 + q];
        sc[j] = (mj == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mj - mt);
        lt += ml[j * 64 + 32 + q] * sc[j];
    }
    if (qq >= P.L) return;
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NW; ++j) acc += part[(j * 64 + dg * 8 + i) * 32 + q] * sc[j];
        o[i] = acc * inv;
    }
    const s4 lo = pack4(o[0], o[1], o[2], o[3]), hi = pack4(o[4], o[5], o[6], o[7]);
    uint4 pk;
    pk.x = reinterpret_cast<const uint32_t *>(&lo)[0], pk.y = reinterpret_cast<const uint32_t *>(&lo)[1];
    pk.z = reinterpret_cast<const uint32_t *>(&hi)[0], pk.w = reinterpret_cast<const uint32_t *>(&hi)[1];
    *reinterpret_cast<uint4 *>(out + ((long)b * P.L + qq) * (P.H * HD) + h * HD + dg * 8) = pk;
    if (dg == 0) lse[(long)bh * P.L + qq] = lt > 0.f ? mt * 0.693147180559945f + logf(lt) : INFINITY;  // m is in the log2 domain
}

// ------------------------------------------------------------------------------------------------ backward
// ONE launch, grid (B*H, nqt + nkt), 4 waves:
//   blockIdx.y <  nqt : role A -- dQ of query tile blockIdx.y; the waves split the KEY tiles, partial dQ summed at the end
//   blockIdx.y >= nqt : role B -- dK, dV of key tile blockIdx.y - nqt; Q / dO staged in LDS once, the waves split the QUERY
//                       tiles, partial dK / dV summed at the end
constexpr int BWD_LDS_A = NW * 2 * TILE_U16 * 2;
constexpr int BWD_LDS_B = 2 * LMAX * RS * 2 + 3 * LMAX * 4;
constexpr int BWD_LDS_C = NW * 64 * 32 * 4;  // combine buffer (aliases the tiles)
constexpr int BWD_LDS = (BWD_LDS_A > BWD_LDS_B ? (BWD_LDS_A > BWD_LDS_C ? BWD_LDS_A : BWD_LDS_C) : (BWD_LDS_B > BWD_LDS_C ? BWD_LDS_B : BWD_LDS_C));

__device__ __forceinline__ void store_rows_bf16(u16 *dst, const float *part, int q, int dg)
{
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NW; ++j) acc += part[(j * 64 + dg * 8 + i) * 32 + q];
        o[i] = acc;
    }
    const s4 lo = pack4(o[0], o[1], o[2], o[3]), hi = pack4(o[4], o[5], o[6], o[7]);
    uint2 a = *reinterpret_cast<const uint2 *>(&lo), c = *reinterpret_cast<const uint2 *>(&hi);
    reinterpret_cast<uint2 *>(dst + dg * 8)[0] = a;
    reinterpret_cast<uint2 *>(dst + dg * 8)[1] = c;
}

template <bool DROP>
__global__ __launch_bounds__(WG, 2) void pcm_attn_small_bwd_kernel(AttnParams P, const u16 *__restrict__ out,
                                                                const u16 *__restrict__ dout, const float *__restrict__ lse,
                                                                u16 *__restrict__ dq, long dq_bs, long dq_ls, u16 *__restrict__ dk,
                                                                long dk_bs, long dk_ls, u16 *__restrict__ dv, long dv_bs, long dv_ls)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[BWD_LDS];
    // the arguments the kernel's first loads need, in registers at the entry: one batch of kernarg loads ("Kernel heads", pcm_common.hpp;
    // the output pointers and strides are left to the compiler: naming all 31 arguments made it spill 26 scalar registers to lanes)
    asm volatile("" ::"s"(P.q), "s"(P.k), "s"(P.v), "s"(P.q_bs), "s"(P.q_ls), "s"(P.k_bs), "s"(P.k_ls), "s"(P.v_bs), "s"(P.v_ls), "s"(P.kpm), "s"(P.H), "s"(P.L), "s"(P.S), "s"(P.p_drop), "s"(P.seed), "s"(out), "s"(dout), "s"(lse));
    const int bh = blockIdx.x, b = bh / P.H, h = bh % P.H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int E = P.H * HD;
    const DropCfg dc(P);
    const u16 *qb = P.q + (long)b * P.q_bs + h * HD;
    const u16 *kb = P.k + (long)b * P.k_bs + h * HD;
    const u16 *vb = P.v + (long)b * P.v_bs + h * HD;
    const u16 *ob = out + (long)b * P.L * E + h * HD;
    const u16 *gb = dout + (long)b * P.L * E + h * HD;
    const unsigned char *mask = P.kpm ? P.kpm + (long)b * P.S : nullptr;
    const int nqt = (P.L + 31) / 32, nkt = (P.S + KT - 1) / KT;
    const int cq = threadIdx.x & 31, cdg = threadIdx.x >> 5;  // combine mapping: column, group of 8 channels
    float *part = reinterpret_cast<float *>(smem);
    if ((int)blockIdx.y < nqt) {
        // ---------------------------------------------------------------- role A: dQ
        u16 *Ks = reinterpret_cast<u16 *>(smem) + w * 2 * TILE_U16, *Vs = Ks + TILE_U16;
        const int qi = blockIdx.y * 32 + (lane & 31);
        const bool qok = qi < P.L;
        s4 qf[8], gf[8];
        float Dq = 0.f;
        const long c0 = 4 * (lane >> 5);
        // a zero upstream gradient gives a zero dQ: skip everything else (layers whose output never reaches the loss -- the
        // decoder's intermediate outputs 1..6, transformer.py:190-206 + act.py:270 -- arrive here with dO == 0)
        bool nz = false;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            gf[sl] = qok ? *reinterpret_cast<const s4 *>(gb + (long)qi * E + sl * 8 + c0) : zero_s4();
            nz |= ((gf[sl][0] | gf[sl][1] | gf[sl][2] | gf[sl][3]) & 0x7FFF) != 0;
        }
        const bool live = __any(nz);
        if (!live) {
            if (w == 0 && qok) {
                u16 *dz = dq + (long)b * dq_bs + (long)qi * dq_ls + h * HD + (lane >> 5) * 32;
#pragma unroll
                for (int i = 0; i < 4; ++i) reinterpret_cast<uint4 *>(dz)[i] = make_uint4(0, 0, 0, 0);
            }
            return;
        }
        {
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                qf[sl] = qok ? *reinterpret_cast<const s4 *>(qb + (long)qi * P.q_ls + sl * 8 + c0) : zero_s4();
                const s4 of = qok ? *reinterpret_cast<const s4 *>(ob + (long)qi * E + sl * 8 + c0) : zero_s4();
#pragma unroll
                for (int i = 0; i < 4; ++i) Dq += bf2f((u16)gf[sl][i]) * bf2f((u16)of[i]);
            }
            Dq += __shfl_xor(Dq, 32);  // the two half-waves hold complementary channels of the same query
        }
        const float scale2 = P.scale * 1.44269504088896f;
        const float lq2 = qok ? lse[(long)bh * P.L + qi] * 1.44269504088896f : INFINITY;  // log2 domain
        const uint32_t rb = DROP ? attn_rowbase(dc.seed, P.site, (uint32_t)(bh * P.L + qi)) : 0u;
        f16v a0, a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] = 0.f, a1[r] = 0.f;
        TileRegs kr, vr;
        if (w < nkt) {
            tile_fetch(kr, kb, P.k_ls, w * KT, P.S, lane);
            tile_fetch(vr, vb, P.v_ls, w * KT, P.S, lane);
        }
        for (int kt = w; kt < nkt; kt += NW) {
            tile_store(kr, Ks, lane);
            tile_store(vr, Vs, lane);
            __builtin_amdgcn_wave_barrier();  // the wave reads this tile back below: one wave's LDS operations execute in order on the hardware; the statement keeps the compiler (and the host model, tests/wavesim) to it -- emits no instruction
            const unsigned mbyte = tile_mask_byte(mask, kt * KT - (lane & 32), P.S, lane);  // one load per tile, before the prefetch (tile_visible, pcm_attn.hpp)
            if (kt + NW < nkt) {
                tile_fetch(kr, kb, P.k_ls, (kt + NW) * KT, P.S, lane);
                tile_fetch(vr, vb, P.v_ls, (kt + NW) * KT, P.S, lane);
            }
            f16v s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int off = (lane & 31) * RS + sl * 8 + 4 * (lane >> 5);
                s = PCM_MFMA(lds_s4(Ks + off), qf[sl], s);
                dp = PCM_MFMA(lds_s4(Vs + off), gf[sl], dp);
            }
            const bool edge = (kt + 1) * KT > P.S || mask != nullptr;
            const lanemask visb = edge ? tile_visible(mbyte, kt * KT - (lane & 32), P.S, lane) >> (4 * (lane >> 5)) : ~0ull;
            float ds[16];
            if (DROP) {
#pragma unroll
                for (int gh = 0; gh < 8; ++gh) {
                    const uint32_t bits = attn_pair_bits(rb, (uint32_t)(kt * (KT / 2) + 4 * (gh >> 1) + 2 * (lane >> 5) + (gh & 1)));
                    lanemask k0, k1;
                    keep_masks(bits, dc.thr, k0, k1);
                    dp[2 * gh] = keep_if(k0, dp[2 * gh] * dc.inv_keep), dp[2 * gh + 1] = keep_if(k1, dp[2 * gh + 1] * dc.inv_keep);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = __builtin_amdgcn_exp2f(s[r] * scale2 - lq2);
                if (edge) pr = ((visb >> ((r & 3) + 8 * (r >> 2))) & 1ull) ? pr : 0.f;
                ds[r] = pr * (dp[r] - Dq) * P.scale;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const s4 df = pack4(ds[4 * g], ds[4 * g + 1], ds[4 * g + 2], ds[4 * g + 3]);
                const u16 *kcol = Ks + (8 * g + 4 * (lane >> 5)) * RS + (lane & 31);  // K^T fragment
                a0 = PCM_MFMA(lds_col4(kcol), df, a0);
                a1 = PCM_MFMA(lds_col4(kcol + 32), df, a1);
            }
        }
        __syncthreads();
        spill_acc(part + w * 64 * 32, a0, a1, lane);
        __syncthreads();
        const int qq = blockIdx.y * 32 + cq;
        if (qq < P.L) store_rows_bf16(dq + (long)b * dq_bs + (long)qq * dq_ls + h * HD, part, cq, cdg);
        return;
    }
    // -------------------------------------------------------------------- role B: dK, dV of one key tile
    u16 *Qs = reinterpret_cast<u16 *>(smem), *dOs = Qs + LMAX * RS;
    float *lse_s = reinterpret_cast<float *>(dOs + LMAX * RS), *D_s = lse_s + LMAX;
    uint32_t *rb_s = reinterpret_cast<uint32_t *>(D_s + LMAX);
    const int kt = blockIdx.y - nqt;
    const int nchunk = (P.L + LMAX - 1) / LMAX;  // queries are staged LMAX rows at a time
    uint4 gpre[LMAX * 8 / WG];  // this thread's dO chunks of the first (for L <= 128: the only) query chunk
    if (nchunk == 1) {
        bool any_dout = false;
#pragma unroll
        for (int it = 0; it < LMAX * 8 / WG; ++it) {
            const int idx = threadIdx.x + it * WG, row = idx >> 3, d0 = (idx & 7) * 8;
            gpre[it] = (row < P.L) ? *reinterpret_cast<const uint4 *>(gb + (long)row * E + d0) : make_uint4(0, 0, 0, 0);
            any_dout |= ((gpre[it].x | gpre[it].y | gpre[it].z | gpre[it].w) & 0x7FFF7FFFu) != 0;
        }
        const bool live = __syncthreads_or(any_dout) != 0;  // zero upstream gradient -> dK = dV = 0 (see role A)
        if (!live) {
            const int kz = kt * KT + (threadIdx.x >> 3), dz = (threadIdx.x & 7) * 8;
            if (kz < P.S) {
                *reinterpret_cast<uint4 *>(dk + (long)b * dk_bs + (long)kz * dk_ls + h * HD + dz) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(dv + (long)b * dv_bs + (long)kz * dv_ls + h * HD + dz) = make_uint4(0, 0, 0, 0);
            }
            return;
        }
    }
    const float scale2 = P.scale * 1.44269504088896f;
    const int key = kt * KT + (lane & 31);
    const bool kin = key < P.S;
    const bool kok = kin && !(mask != nullptr && mask[key] != 0);
    s4 kf[8], vf[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
        kf[sl] = kin ? *reinterpret_cast<const s4 *>(kb + (long)key * P.k_ls + sl * 8 + 4 * (lane >> 5)) : zero_s4();
        vf[sl] = kin ? *reinterpret_cast<const s4 *>(vb + (long)key * P.v_ls + sl * 8 + 4 * (lane >> 5)) : zero_s4();
    }
    f16v dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) dk0[r] = 0.f, dk1[r] = 0.f, dv0[r] = 0.f, dv1[r] = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int row0 = ch * LMAX;
        const int tiles_here = ((P.L - row0 < LMAX ? P.L - row0 : LMAX) + 31) / 32;
        if (ch > 0) __syncthreads();  // the previous chunk's tiles are no longer read
#pragma unroll
        for (int it = 0; it < LMAX * 8 / WG; ++it) {  // stage Q and dO rows (zero beyond L), D = rowsum(dO * O)
            const int idx = threadIdx.x + it * WG;
            if (idx >= tiles_here * 32 * 8) break;
            const int lrow = idx >> 3, row = row0 + lrow, d0 = (idx & 7) * 8;  // 8 consecutive lanes share a row
            uint4 qv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0), gv = make_uint4(0, 0, 0, 0);
            if (row < P.L) {
                qv = *reinterpret_cast<const uint4 *>(qb + (long)row * P.q_ls + d0);
                ov = *reinterpret_cast<const uint4 *>(ob + (long)row * E + d0);
                gv = (nchunk == 1) ? gpre[it] : *reinterpret_cast<const uint4 *>(gb + (long)row * E + d0);
            }
            uint2 *a = reinterpret_cast<uint2 *>(Qs + lrow * RS + d0);
            a[0] = make_uint2(qv.x, qv.y), a[1] = make_uint2(qv.z, qv.w);
            uint2 *c = reinterpret_cast<uint2 *>(dOs + lrow * RS + d0);
            c[0] = make_uint2(gv.x, gv.y), c[1] = make_uint2(gv.z, gv.w);
            const u16 *oe = reinterpret_cast<const u16 *>(&ov), *ge = reinterpret_cast<const u16 *>(&gv);
            float dsum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) dsum += bf2f(oe[i]) * bf2f(ge[i]);
            dsum += __shfl_xor(dsum, 1);
            dsum += __shfl_xor(dsum, 2);
            dsum += __shfl_xor(dsum, 4);
            if ((idx & 7) == 0) {
                D_s[lrow] = dsum;
                lse_s[lrow] = row < P.L ? lse[(long)bh * P.L + row] * 1.44269504088896f : INFINITY;  // log2 domain; +inf silences padded queries
                rb_s[lrow] = DROP ? attn_rowbase(dc.seed, P.site, (uint32_t)(bh * P.L + row)) : 0u;
            }
        }
        __syncthreads();
        for (int qt = w; qt < tiles_here; qt += NW) {
            f16v s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int off = (qt * 32 + (lane & 31)) * RS + sl * 8 + 4 * (lane >> 5);
                s = PCM_MFMA(lds_s4(Qs + off), kf[sl], s);  // rows = queries, columns = keys
                dp = PCM_MFMA(lds_s4(dOs + off), vf[sl], dp);
            }
            float pd[16], ds[16];
            // dropout bits: a lane is a key here, so the two keys of a hash word sit in adjacent lanes -- the even lane hashes
            // query 2j, the odd lane query 2j+1, and the 2 x 2 keep bits are routed as wave masks (see attn_flash.hip, dK/dV)
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 = four consecutive queries
                lanemask km[4] = {0, 0, 0, 0};
                if (DROP) {
                    constexpr lanemask EV = 0x5555555555555555ull, OD = 0xAAAAAAAAAAAAAAAAull;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const uint32_t bits = attn_pair_bits(rb_s[qt * 32 + crow(4 * g + 2 * jj, lane) + (lane & 1)], (uint32_t)key >> 1);
                        lanemask clo, chi;
                        keep_masks(bits, dc.thr, clo, chi);
                        km[2 * jj] = (clo & EV) | ((chi & EV) << 1);
                        km[2 * jj + 1] = (chi & OD) | ((clo & OD) >> 1);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g + i;
                    const int qr = qt * 32 + crow(r, lane);  // row within the staged chunk
                    const float pr = kok ? __builtin_amdgcn_exp2f(s[r] * scale2 - lse_s[qr]) : 0.f;  // lse_s holds lse * log2(e)
                    float dpv = dp[r];
                    pd[r] = pr;
                    if (DROP) {
                        pd[r] = keep_if(km[i], pr * dc.inv_keep);
                        dpv = keep_if(km[i], dpv * dc.inv_keep);
                    }
                    ds[r] = pr * (dpv - D_s[qr]) * P.scale;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const s4 pf = pack4(pd[4 * g], pd[4 * g + 1], pd[4 * g + 2], pd[4 * g + 3]);
                const s4 df = pack4(ds[4 * g], ds[4 * g + 1], ds[4 * g + 2], ds[4 * g + 3]);
                // A operands = dO^T / Q^T: lane = channel row, 4 consecutive queries -> a column walk through the row-major tiles
                const int qrow = qt * 32 + 8 * g + 4 * (lane >> 5);
                const u16 *gcol = dOs + qrow * RS + (lane & 31), *qcol = Qs + qrow * RS + (lane & 31);
                dv0 = PCM_MFMA(lds_col4(gcol), pf, dv0);
                dv1 = PCM_MFMA(lds_col4(gcol + 32), pf, dv1);
                dk0 = PCM_MFMA(lds_col4(qcol), df, dk0);
                dk1 = PCM_MFMA(lds_col4(qcol + 32), df, dk1);
            }
        }
    }
    const int kk = kt * KT + cq;
    __syncthreads();  // everybody is done with Qs / dOs: the combine buffer aliases them
    spill_acc(part + w * 64 * 32, dk0, dk1, lane);
    __syncthreads();
    if (kk < P.S) store_rows_bf16(dk + (long)b * dk_bs + (long)kk * dk_ls + h * HD, part, cq, cdg);
    __syncthreads();
    spill_acc(part + w * 64 * 32, dv0, dv1, lane);
    __syncthreads();
    if (kk < P.S) store_rows_bf16(dv + (long)b * dv_bs + (long)kk * dv_ls + h * HD, part, cq, cdg);
}

inline bool strides_ok(long bs, long ls)
{
    return bs % 8 == 0 && ls % 8 == 0;  // 16-byte row loads
}

}  // namespace

extern "C" int pcm_attn_small_supported(int L, int S, int head_dim)
{
    return (head_dim == HD && L >= 1 && S >= 1) ? 1 : 0;
}

extern "C" int pcm_attn_small_forward_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k,
                                          long k_bs, long k_ls, const void *v, long v_bs, long v_ls,
                                          const unsigned char *key_padding_mask, float scale, float p_drop, const long *seed,
                                          unsigned site, void *out, float *lse, void *stream)
{
    if (B <= 0 || H <= 0) return B == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (!pcm_attn_small_supported(L, S, HD)) return PCM_ERR_UNSUPPORTED;
    if (!strides_ok(q_bs, q_ls) || !strides_ok(k_bs, k_ls) || !strides_ok(v_bs, v_ls)) return PCM_ERR_BAD_ARG;
    if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && seed == nullptr)) return PCM_ERR_BAD_ARG;
    AttnParams P{(const u16 *)q, (const u16 *)k, (const u16 *)v, q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, key_padding_mask,
                 B, H, L, S, scale, p_drop, seed, site};
    hipLaunchKernelGGL(p_drop > 0.f ? pcm_attn_small_fwd_kernel<true> : pcm_attn_small_fwd_kernel<false>, dim3(B * H, (L + 31) / 32), dim3(WG), 0,
                       (hipStream_t)stream, P, (u16 *)out, lse);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_attn_small_backward_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k,
                                           long k_bs, long k_ls, const void *v, long v_bs, long v_ls,
                                           const unsigned char *key_padding_mask, float scale, float p_drop, const long *seed,
                                           unsigned site, const void *out, const void *dout, const float *lse, void *dq,
                                           long dq_bs, long dq_ls, void *dk, long dk_bs, long dk_ls, void *dv, long dv_bs,
                                           long dv_ls, void *stream)
{
    if (B <= 0 || H <= 0) return B == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (!pcm_attn_small_supported(L, S, HD)) return PCM_ERR_UNSUPPORTED;
    if (!strides_ok(q_bs, q_ls) || !strides_ok(k_bs, k_ls) || !strides_ok(v_bs, v_ls)) return PCM_ERR_BAD_ARG;
    if (dq_ls % 8 || dk_ls % 8 || dv_ls % 8 || dq_bs % 8 || dk_bs % 8 || dv_bs % 8) return PCM_ERR_BAD_ARG;
    AttnParams P{(const u16 *)q, (const u16 *)k, (const u16 *)v, q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, key_padding_mask,
                 B, H, L, S, scale, p_drop, seed, site};
    hipLaunchKernelGGL(p_drop > 0.f ? pcm_attn_small_bwd_kernel<true> : pcm_attn_small_bwd_kernel<false>, dim3(B * H, (L + 31) / 32 + (S + KT - 1) / KT),
                       dim3(WG), 0, (hipStream_t)stream, P, (const u16 *)out,
                       (const u16 *)dout, lse, (u16 *)dq, dq_bs, dq_ls, (u16 *)dk, dk_bs, dk_ls, (u16 *)dv, dv_bs, dv_ls);
    return PCM_LAUNCH_STATUS();
}
