// attn_flash.hip -- multi-head attention for LONG query sets (head_dim 64), forward and backward, on the gfx950 matrix
// cores.  Replaces the framework's flash kernels on the encoder's self-attention over the point tokens
//   (/root/reference/src/models/components/act/transformer.py:221,244-262,297-298: nn.MultiheadAttention over
//    S = M + 2 or 3 tokens -- 515 at BASELINE configs[1], 1027 at configs[3], 2051 at the shipped ACT config).
// csrc/attn_small.hip keeps the short query sets (<= 128 queries: CVAE encoder, decoder).
//
// Shape of the work.  A workgroup owns 128 rows of the NON-streamed side (4 waves x 32 rows, the rows live in registers
// as MFMA B operands for the whole kernel) and streams the other side in tiles of 64 rows that all four waves share
// through LDS (double-buffered, one barrier per tile: a K/V tile is fetched from L2 once per 128 queries, 4x less often
// than with attn_small's 32-query workgroups -- at 2051 x 2051 that is the difference between 2.1 GB and 0.5 GB of
// L2 -> LDS traffic per call).
//
// Layout trick (as in attn_small.hip): scores are computed TRANSPOSED, S^T (keys x queries) = K_tile . Q^T, so the
// accumulator layout (lane = query column, registers = key rows) IS the B-operand layout of the second GEMM of the
// tile: P never leaves registers, and a lane owns one query's online-softmax state.  The A operand of that second GEMM
// is V^T (lane = channel row, k = keys): a column walk through the row-major V tile.  Here it is ONE
// ds_read_b64_tr_b16 per 4 keys (the LDS transpose read of gfx950; semantics pinned by tools/mb/tr_probe.hip) instead
// of four 2-byte reads.  Tiles are stored unpadded with an XOR swizzle of the 16-byte chunk index,
//   chunk' = chunk ^ f(row),  f(row) = ((row >> 1) & 1) << 2 | ((row >> 2) & 3),
// which makes BOTH access patterns bank-conflict free: ds_read_b128 of 16 different rows at one chunk (row-major A
// operands) and the transpose read's 4 rows x 32 bytes per 16-lane group.
//
// Backward = three launches, no atomics (the sums are fixed-order, so two runs give the same bits):
//   prep   delta[b,h,q] = sum_d dO . O
//   dq     one workgroup per 128 queries, streams K / V:   S^T, dP^T = V dO^T, dQ^T += K^T dS^T
//   dkv    one workgroup per 128 keys,    streams Q / dO:  S, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
// Remainder rows.  The token sets of this path are M + 3 long (515, 1027, 2051): after the full 128-row blocks, 3 rows are
// left per (batch, head), and a workgroup for them costs as much as a full one -- at 2051 tokens that is 64 x 17 = 1088
// workgroups on 512 resident slots, i.e. a THIRD pass over the chip for 0.15 % of the rows (measured: SQ_WAVE_CYCLES says
// every wave lives 1/3 of the kernel).  When the remainder is <= 32 rows its workgroup therefore runs in "rem" mode: all
// four waves stage the streamed tiles as usual, but waves 0 and 1 each take one 32-row HALF of every streamed tile for the
// same resident rows and their partial results are merged through LDS at the end (online-softmax merge in the forward,
// plain sums in the backward kernels): half the time of a full workgroup instead of all of it.
// Dropout on the attention weights: the counter hash of attn_small.hip (same bits for the same (seed, site, b, h, q, key),
// recomputed in both backward kernels).  P and dS are rounded to bf16 for the second GEMMs like every flash kernel.
// q, k, v: bf16, arbitrary batch / row strides (multiples of 8), unit stride over the 64 head channels (head h at column
// h*64); out, dout: (B, L, H*64) contiguous; lse, delta: (B, H, L) fp32.
#include "pcm_attn.hpp"

namespace {

constexpr int HD = 64;         // head dim
constexpr int NW = 4;          // waves per workgroup
constexpr int WG = 64 * NW;
constexpr int RW = 32;         // resident rows per wave
constexpr int RWG = RW * NW;   // resident rows per workgroup
constexpr int TR = 64;         // rows per streamed tile
constexpr int TILE = TR * HD;  // u16 per tile: 8 KiB, swizzled, unpadded
constexpr int OS = 72;         // row stride (u16) of the epilogue staging area

__device__ __forceinline__ int swz(int r)
{
    return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
}

typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;
__device__ __forceinline__ s4 lds_tr(const u16 *p)  // ds_read_b64_tr_b16
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)p);
}

// per-lane element offsets into a swizzled tile, fixed for the whole kernel
struct LaneOffsets {
    int row[4];    // row-major A operand (lane = tile row lane&31): k-slice sl = channels 16 sl + 8 (lane>>5) ..+7; +32 rows: +2048
    int tr[2][2];  // transposed A operand (lane = channel): [second 4-row group (+8 rows)][channel block (+32)]; +16 rows: +1024
};

__device__ __forceinline__ LaneOffsets lane_offsets(int lane)
{
    LaneOffsets o;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) o.row[sl] = r * HD + (((2 * sl + h) ^ swz(r)) << 3);
    const int p = lane & 15, g2 = (lane >> 4) & 1;
#pragma unroll
    for (int half2 = 0; half2 < 2; ++half2)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int row = 8 * half2 + 4 * h + (p >> 2);
            const int chunk = 4 * db + 2 * g2 + ((p & 3) >> 1);
            o.tr[half2][db] = row * HD + ((chunk ^ swz(row)) << 3) + 4 * (p & 1);
        }
    return o;
}

// 64-row tile: 512 chunks of 16 B, 2 per thread
__device__ __forceinline__ void stage_fetch(uint4 (&r)[2], const u16 *__restrict__ base, long row_stride, int row0, int nrows, int tid)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + WG * i, row = id >> 3, ch = id & 7;
        r[i] = (row0 + row < nrows) ? *reinterpret_cast<const uint4 *>(base + (long)(row0 + row) * row_stride + ch * 8) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void stage_store(const uint4 (&r)[2], u16 *tile, int tid)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + WG * i, row = id >> 3, ch = id & 7;
        *reinterpret_cast<uint4 *>(tile + row * HD + ((ch ^ swz(row)) << 3)) = r[i];
    }
}

// the two 32 x 32 accumulators of a wave (rows = channels, columns = this wave's 32 resident rows) -> bf16 rows in global
// memory through a wave-private LDS area [32][OS]; `scale_col` multiplies every column (1 / softmax sum in the forward).
__device__ __forceinline__ void store_acc_rows(u16 *stage, const f16v &a0, const f16v &a1, float scale_col, u16 *__restrict__ dst,
                                               long row_stride, int row0, int nrows, int lane)
{
    const int h4 = 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const s4 lo = pack4(a0[4 * g] * scale_col, a0[4 * g + 1] * scale_col, a0[4 * g + 2] * scale_col, a0[4 * g + 3] * scale_col);
        const s4 hi = pack4(a1[4 * g] * scale_col, a1[4 * g + 1] * scale_col, a1[4 * g + 2] * scale_col, a1[4 * g + 3] * scale_col);
        *reinterpret_cast<s4 *>(stage + (lane & 31) * OS + 8 * g + h4) = lo;
        *reinterpret_cast<s4 *>(stage + (lane & 31) * OS + 32 + 8 * g + h4) = hi;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = lane + 64 * i, row = id >> 3, ch = id & 7;
        if (row0 + row < nrows) {
            const uint2 x = *reinterpret_cast<const uint2 *>(stage + row * OS + ch * 8), y = *reinterpret_cast<const uint2 *>(stage + row * OS + ch * 8 + 4);
            *reinterpret_cast<uint4 *>(dst + (long)(row0 + row) * row_stride + ch * 8) = make_uint4(x.x, x.y, y.x, y.y);
        }
    }
}

__device__ __forceinline__ bf8 p_frag(const float (&p)[16], int j8)  // registers j8 .. j8+7 -> one B operand
{
    return cat8(pack4(p[j8], p[j8 + 1], p[j8 + 2], p[j8 + 3]), pack4(p[j8 + 4], p[j8 + 5], p[j8 + 6], p[j8 + 7]));
}

// ------------------------------------------------------------------------------------------------ forward
// grid (B*H, ceil(L / 128))
template <bool DROP>  // dropout on / off is compiled in: as a run-time flag it left a branch around every masked element
__global__ __launch_bounds__(WG, 2) void pcm_attn_flash_fwd_kernel(AttnParams P, u16 *__restrict__ out, float *__restrict__ lse)
{
    __shared__ __attribute__((aligned(16))) u16 smem[4 * TILE];  // K[2] | V[2]; the epilogue stages O in it
    const int bh = blockIdx.x, b = bh / P.H, h = bh % P.H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nfull = P.L / RWG, remr = P.L - nfull * RWG;
    const bool rem = remr > 0 && remr <= RW && (int)blockIdx.y == nfull;  // "rem" mode, see the header
    const int q0 = rem ? nfull * RWG : blockIdx.y * RWG + w * RW, qi = q0 + (lane & 31);
    const bool qok = qi < P.L;
    const LaneOffsets lo = lane_offsets(lane);
    bf8 qf[4];
    {
        const u16 *qp = P.q + (long)b * P.q_bs + (long)qi * P.q_ls + h * HD + 8 * (lane >> 5);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) qf[sl] = as_bf8(qok ? *reinterpret_cast<const uint4 *>(qp + sl * 16) : make_uint4(0, 0, 0, 0));
    }
    const DropCfg dc(P);
    const uint32_t rb = DROP ? attn_rowbase(dc.seed, P.site, (uint32_t)(bh * P.L + qi)) : 0u;
    const int hl = lane >> 5;
    f16v o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = 0.f, o1[r] = 0.f;
    float m = -INFINITY, lsum = 0.f;
    const u16 *kb = P.k + (long)b * P.k_bs + h * HD;
    const u16 *vb = P.v + (long)b * P.v_bs + h * HD;
    const unsigned char *mask = P.kpm ? P.kpm + (long)b * P.S : nullptr;
    const int ntiles = (P.S + TR - 1) / TR;
    const float scale2 = P.scale * 1.44269504088896f;  // scores in the log2 domain
    uint4 kr[2], vr[2];
    stage_fetch(kr, kb, P.k_ls, 0, P.S, tid);
    stage_fetch(vr, vb, P.v_ls, 0, P.S, tid);
    stage_store(kr, smem, tid);
    stage_store(vr, smem + 2 * TILE, tid);
    __syncthreads();
    if (rem) {
        // waves 0 / 1: keys 0..31 / 32..63 of every tile against the same (remainder) queries
        for (int kt = 0; kt < ntiles; ++kt) {
            const u16 *Kt = smem + (kt & 1) * TILE, *Vt = smem + (2 + (kt & 1)) * TILE;
            const unsigned mbyte = tile_mask_byte(mask, kt * TR, P.S, lane);  // before the prefetch: see tile_visible
            if (kt + 1 < ntiles) {
                stage_fetch(kr, kb, P.k_ls, (kt + 1) * TR, P.S, tid);
                stage_fetch(vr, vb, P.v_ls, (kt + 1) * TR, P.S, tid);
            }
            const int key0 = kt * TR + 32 * w;
            if (w < 2 && key0 < P.S) {  // wave-uniform
                f16v s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) s = PCM_MFMA16(lds_bf8(Kt + lo.row[sl] + w * 32 * HD), qf[sl], s);
                bf8 va0[2], va1[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const u16 *vs = Vt + (2 * w + jj) * 16 * HD;
                    va0[jj] = cat8(lds_tr(vs + lo.tr[0][0]), lds_tr(vs + lo.tr[1][0]));
                    va1[jj] = cat8(lds_tr(vs + lo.tr[0][1]), lds_tr(vs + lo.tr[1][1]));
                }
                if (key0 + 32 > P.S || mask != nullptr) {
                    const lanemask vb = tile_visible(mbyte, kt * TR, P.S, lane) >> (32 * w + 4 * hl);  // bit crow(r, 0) <-> this lane's row r
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = ((vb >> ((r & 3) + 8 * (r >> 2))) & 1ull) ? s[r] : -INFINITY;
                }
                float tmax = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; r += 2) tmax = max3f(tmax, s[r], s[r + 1]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float m_new = fmaxf(m, tmax);
                const bool dead = m_new == -INFINITY;
                const float mneg = dead ? 0.f : -m_new * scale2;
                float p[16], psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale2, mneg));
                    psum += p[r];
                }
                psum += __shfl_xor(psum, 32);
                if (__any(m_new != m)) {
                    const float alpha = (dead || m == -INFINITY) ? (dead ? 1.f : 0.f) : __builtin_amdgcn_exp2f((m - m_new) * scale2);
                    lsum *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o0[r] *= alpha, o1[r] *= alpha;
                }
                lsum += psum;
                m = m_new;
                if (DROP) {
#pragma unroll
                    for (int gh = 0; gh < 8; ++gh) {
                        const uint32_t bits = attn_pair_bits(rb, (uint32_t)(kt * (TR / 2) + 16 * w + 4 * (gh >> 1) + 2 * hl + (gh & 1)));
                        lanemask k0, k1;
                        keep_masks(bits, dc.thr, k0, k1);
                        p[2 * gh] = keep_if(k0, p[2 * gh]), p[2 * gh + 1] = keep_if(k1, p[2 * gh + 1]);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const bf8 pf = p_frag(p, 8 * jj);
                    o0 = PCM_MFMA16(va0[jj], pf, o0);
                    o1 = PCM_MFMA16(va1[jj], pf, o1);
                }
            }
            if (kt + 1 < ntiles) {
                stage_store(kr, smem + ((kt + 1) & 1) * TILE, tid);
                stage_store(vr, smem + (2 + ((kt + 1) & 1)) * TILE, tid);
            }
            __syncthreads();
        }
        // wave 1's partial (m, sum, O^T) joins wave 0's: the usual online-softmax merge
        float *part = reinterpret_cast<float *>(smem) + 2048;  // 8 KiB in: clear of wave 0's epilogue staging rows
        if (w == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r * 64 + lane] = o0[r], part[(16 + r) * 64 + lane] = o1[r];
            part[32 * 64 + lane] = m, part[33 * 64 + lane] = lsum;
        }
        __syncthreads();
        if (w != 0) return;
        const float m1 = part[32 * 64 + lane], l1 = part[33 * 64 + lane];
        const float mm = fmaxf(m, m1);
        const float f0 = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m - mm) * scale2);
        const float f1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m1 - mm) * scale2);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] = o0[r] * f0 + part[r * 64 + lane] * f1;
            o1[r] = o1[r] * f0 + part[(16 + r) * 64 + lane] * f1;
        }
        lsum = lsum * f0 + l1 * f1;
        m = mm;
    } else {
    // Software pipeline (round 6): the score GEMM of tile t + 1 is ISSUED under the softmax of tile t.  The K ring runs
    // one tile ahead of the V ring: at the top of iteration t the LDS holds K(t + 1) -- staged an iteration earlier -- and V(t); registers
    // fetch K(t + 2) and V(t + 1), stored at the END of the iteration into the buffers of K(t) (last read in iteration t - 1) and V(t - 1)
    // (last read in iteration t - 1): still one barrier per tile.  The matrix instructions of S(t + 1) have no consumer before the next
    // iteration, so they drain while the VALU works through the softmax, the dropout hash and the bf16 packing of tile t -- per tile and
    // wave that VALU work is ~4x the MFMA time (static count), and until now the wave could not start it before ITS OWN score GEMM had
    // returned.  Same arithmetic on the same operands in the same order: the results are the same bits.
    auto scores = [&](const u16 *Kt, f16v &a0, f16v &a1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] = 0.f, a1[r] = 0.f;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            a0 = PCM_MFMA16(lds_bf8(Kt + lo.row[sl]), qf[sl], a0);
            a1 = PCM_MFMA16(lds_bf8(Kt + lo.row[sl] + 32 * HD), qf[sl], a1);
        }
    };
    if (ntiles > 1) {  // K(1) joins K(0) | V(0) in LDS before the first iteration
        stage_fetch(kr, kb, P.k_ls, TR, P.S, tid);
        stage_store(kr, smem + TILE, tid);
        __syncthreads();
    }
    f16v s0, s1;
    scores(smem, s0, s1);
    for (int kt = 0; kt < ntiles; ++kt) {
        const u16 *Vt = smem + (2 + (kt & 1)) * TILE;
        const unsigned mbyte = tile_mask_byte(mask, kt * TR, P.S, lane);  // before the prefetch: see tile_visible
        if (kt + 2 < ntiles) stage_fetch(kr, kb, P.k_ls, (kt + 2) * TR, P.S, tid);
        if (kt + 1 < ntiles) stage_fetch(vr, vb, P.v_ls, (kt + 1) * TR, P.S, tid);
        f16v n0, n1;  // S(t + 1): issued now, first read in the next iteration
        if (kt + 1 < ntiles) scores(smem + ((kt + 1) & 1) * TILE, n0, n1);
        const bool edge = (kt + 1) * TR > P.S || mask != nullptr;
        // the running maximum is kept on the RAW scores (scale > 0 commutes with max): the scale is folded into the
        // exponent's FMA, exp2(s * scale2 - m * scale2), instead of costing one multiply per score
        if (edge) {
            const lanemask vis = tile_visible(mbyte, kt * TR, P.S, lane) >> (4 * hl);  // bit crow(r, 0) (+ 32) <-> this lane's row r
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sh = (r & 3) + 8 * (r >> 2);
                s0[r] = ((vis >> sh) & 1ull) ? s0[r] : -INFINITY;
                s1[r] = ((vis >> (sh + 32)) & 1ull) ? s1[r] : -INFINITY;
            }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = max3f(tmax, s0[r], s1[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m, tmax);
        const bool dead = m_new == -INFINITY;  // nothing visible yet for this query: every score is -inf
        const float mneg = dead ? 0.f : -m_new * scale2;
        float p0[16], p1[16], psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], scale2, mneg));  // dead: exp2(-inf * scale2 + 0) = 0, no select needed
            p1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], scale2, mneg));
            psum += p0[r] + p1[r];
        }
        psum += __shfl_xor(psum, 32);
        if (__any(m_new != m)) {  // the running maximum moves in the first tiles only
            const float alpha = (dead || m == -INFINITY) ? (dead ? 1.f : 0.f) : __builtin_amdgcn_exp2f((m - m_new) * scale2);
            lsum *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) o0[r] *= alpha, o1[r] *= alpha;
        }
        lsum += psum;
        m = m_new;
        // the transposed V fragments of the whole tile are requested HERE -- behind the exponentials, in front of the dropout hash and the
        // bf16 packing, under which the LDS reads drain (at the top of the tile they held 32 registers through the whole softmax, which the
        // pipelined S(t + 1) accumulators now need: 256 VGPRs and 6 spilled otherwise)
        bf8 va0[2], va1[2];  // two 16-key slabs at a time: the second pair is requested behind the first pair's matrix instructions
        auto vfrag = [&](int j0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u16 *vs = Vt + (j0 + j) * 16 * HD;
                va0[j] = cat8(lds_tr(vs + lo.tr[0][0]), lds_tr(vs + lo.tr[1][0]));
                va1[j] = cat8(lds_tr(vs + lo.tr[0][1]), lds_tr(vs + lo.tr[1][1]));
            }
        };
        vfrag(0);
        if (DROP) {
#pragma unroll
            for (int gh = 0; gh < 8; ++gh) {  // registers 2gh, 2gh+1 hold adjacent keys: one hash for the pair
                const uint32_t pair = (uint32_t)(kt * (TR / 2) + 4 * (gh >> 1) + 2 * hl + (gh & 1));
                const uint32_t b0 = attn_pair_bits(rb, pair), b1 = attn_pair_bits(rb, pair + 16);
                // the 1 / (1 - p) rescale of the kept weights is a constant: it is folded into the final normalisation of O
                lanemask k0, k1, k2, k3;
                keep_masks(b0, dc.thr, k0, k1);
                keep_masks(b1, dc.thr, k2, k3);
                p0[2 * gh] = keep_if(k0, p0[2 * gh]), p0[2 * gh + 1] = keep_if(k1, p0[2 * gh + 1]);
                p1[2 * gh] = keep_if(k2, p1[2 * gh]), p1[2 * gh + 1] = keep_if(k3, p1[2 * gh + 1]);
            }
        }
        // O^T += V^T P^T, 16 keys per MFMA: slab j = keys 16j .. 16j+15 of the tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf8 pf = p_frag(p0, 8 * j);
            o0 = PCM_MFMA16(va0[j], pf, o0);
            o1 = PCM_MFMA16(va1[j], pf, o1);
        }
        vfrag(2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf8 pf = p_frag(p1, 8 * j);
            o0 = PCM_MFMA16(va0[j], pf, o0);
            o1 = PCM_MFMA16(va1[j], pf, o1);
        }
        if (kt + 2 < ntiles) stage_store(kr, smem + (kt & 1) * TILE, tid);               // K(t + 2) over K(t)
        if (kt + 1 < ntiles) stage_store(vr, smem + (2 + ((kt + 1) & 1)) * TILE, tid);   // V(t + 1) over V(t - 1)
        __syncthreads();
        if (kt + 1 < ntiles) s0 = n0, s1 = n1;
    }
    }
    const float inv = lsum > 0.f ? dc.inv_keep / lsum : 0.f;
    store_acc_rows(smem + w * RW * OS, o0, o1, inv, out + (long)b * P.L * (P.H * HD) + h * HD, (long)P.H * HD, q0, P.L, lane);
    if (lane < 32 && qok) lse[(long)bh * P.L + qi] = lsum > 0.f ? m * P.scale + logf(lsum) : INFINITY;  // m: raw-score maximum
}

// ------------------------------------------------------------------------------------------------ backward
// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]: 8 lanes per (row, head), 16 bytes each
__global__ __launch_bounds__(WG) void pcm_attn_flash_prep_kernel(int B, int H, int L, const u16 *__restrict__ out,
                                                                 const u16 *__restrict__ dout, float *__restrict__ delta)
{
    const long total = (long)B * L * H * 8;
    for (long t = (long)blockIdx.x * WG + threadIdx.x; t < total; t += (long)gridDim.x * WG) {
        const uint4 o = *reinterpret_cast<const uint4 *>(out + t * 8), g = *reinterpret_cast<const uint4 *>(dout + t * 8);
        const u16 *oe = reinterpret_cast<const u16 *>(&o), *ge = reinterpret_cast<const u16 *>(&g);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += bf2f(oe[i]) * bf2f(ge[i]);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if ((t & 7) == 0) {
            const long rh = t >> 3;  // (b*L + q) * H + h
            const long row = rh / H;
            const int hh = (int)(rh - row * H);
            const long bb = row / L, q = row - bb * L;
            delta[(bb * H + hh) * L + q] = s;
        }
    }
}

// dQ: grid (B*H, ceil(L / 128)); streams K / V tiles
template <bool DROP>  // dropout on / off is compiled in: as a run-time flag it left a branch around every masked element
__global__ __launch_bounds__(WG, 2) void pcm_attn_flash_bwd_dq_kernel(AttnParams P, const u16 *__restrict__ dout, const float *__restrict__ lse,
                                                                   const float *__restrict__ delta, u16 *__restrict__ dq, long dq_bs,
                                                                   long dq_ls)
{
    __shared__ __attribute__((aligned(16))) u16 smem[4 * TILE];
    const int bh = blockIdx.x, b = bh / P.H, h = bh % P.H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hl = lane >> 5;
    const int E = P.H * HD;
    const int nfull = P.L / RWG, remr = P.L - nfull * RWG;
    const bool rem = remr > 0 && remr <= RW && (int)blockIdx.y == nfull;  // "rem" mode, see the header
    const int q0 = rem ? nfull * RWG : blockIdx.y * RWG + w * RW, qi = q0 + (lane & 31);
    const bool qok = qi < P.L;
    const LaneOffsets lo = lane_offsets(lane);
    bf8 qf[4], gf[4];
    {
        const u16 *qp = P.q + (long)b * P.q_bs + (long)qi * P.q_ls + h * HD + 8 * hl;
        const u16 *gp = dout + ((long)b * P.L + qi) * E + h * HD + 8 * hl;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            qf[sl] = as_bf8(qok ? *reinterpret_cast<const uint4 *>(qp + sl * 16) : make_uint4(0, 0, 0, 0));
            gf[sl] = as_bf8(qok ? *reinterpret_cast<const uint4 *>(gp + sl * 16) : make_uint4(0, 0, 0, 0));
        }
    }
    const DropCfg dc(P);
    const uint32_t rb = DROP ? attn_rowbase(dc.seed, P.site, (uint32_t)(bh * P.L + qi)) : 0u;
    const float scale2 = P.scale * 1.44269504088896f;
    const float lq2 = qok ? lse[(long)bh * P.L + qi] * 1.44269504088896f : INFINITY;  // +inf silences padded queries
    // dS = p * (keep * dP / (1 - p_drop) - D) * scale  =  p * (keep * dP - D * (1 - p_drop)) * (scale / (1 - p_drop))
    const float Dq = (qok ? delta[(long)bh * P.L + qi] : 0.f) / dc.inv_keep;
    const float sk = P.scale * dc.inv_keep;
    f16v a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = 0.f, a1[r] = 0.f;
    const u16 *kb = P.k + (long)b * P.k_bs + h * HD;
    const u16 *vb = P.v + (long)b * P.v_bs + h * HD;
    const unsigned char *mask = P.kpm ? P.kpm + (long)b * P.S : nullptr;
    const int ntiles = (P.S + TR - 1) / TR;
    uint4 kr[2], vr[2];
    stage_fetch(kr, kb, P.k_ls, 0, P.S, tid);
    stage_fetch(vr, vb, P.v_ls, 0, P.S, tid);
    stage_store(kr, smem, tid);
    stage_store(vr, smem + 2 * TILE, tid);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const u16 *Kt = smem + (kt & 1) * TILE, *Vt = smem + (2 + (kt & 1)) * TILE;
        const unsigned mbyte = tile_mask_byte(mask, kt * TR, P.S, lane);  // before the prefetch: see tile_visible (pcm_attn.hpp)
        if (kt + 1 < ntiles) {
            stage_fetch(kr, kb, P.k_ls, (kt + 1) * TR, P.S, tid);
            stage_fetch(vr, vb, P.v_ls, (kt + 1) * TR, P.S, tid);
        }
        const bool edge = (kt + 1) * TR > P.S || mask != nullptr;
        const lanemask vis = edge ? tile_visible(mbyte, kt * TR, P.S, lane) >> (4 * hl) : ~0ull;  // bit crow(r, 0) + 32 kh <-> row r of half kh
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {  // 32 keys at a time
            if (rem && kh != w) continue;  // rem mode: wave 0 / 1 owns key half 0 / 1, waves 2 and 3 only stage
            f16v s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                s = PCM_MFMA16(lds_bf8(Kt + lo.row[sl] + kh * 32 * HD), qf[sl], s);
                dp = PCM_MFMA16(lds_bf8(Vt + lo.row[sl] + kh * 32 * HD), gf[sl], dp);
            }
            // the transposed K fragments of this 32-key half are requested before the softmax arithmetic below
            bf8 kt0[2], kt1[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const u16 *ks = Kt + (2 * kh + jj) * 16 * HD;
                kt0[jj] = cat8(lds_tr(ks + lo.tr[0][0]), lds_tr(ks + lo.tr[1][0]));
                kt1[jj] = cat8(lds_tr(ks + lo.tr[0][1]), lds_tr(ks + lo.tr[1][1]));
            }
            if (DROP) {
#pragma unroll
                for (int gh = 0; gh < 8; ++gh) {
                    const uint32_t bits = attn_pair_bits(rb, (uint32_t)(kt * (TR / 2) + 16 * kh + 4 * (gh >> 1) + 2 * hl + (gh & 1)));
                    lanemask k0, k1;
                    keep_masks(bits, dc.thr, k0, k1);
                    dp[2 * gh] = keep_if(k0, dp[2 * gh]), dp[2 * gh + 1] = keep_if(k1, dp[2 * gh + 1]);  // 1 / (1 - p): in Dq / sk
                }
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale2, -lq2));
                if (edge) pr = ((vis >> ((r & 3) + 8 * (r >> 2) + 32 * kh)) & 1ull) ? pr : 0.f;
                ds[r] = pr * (dp[r] - Dq);  // the constant factor scale / (1 - p_drop) is applied once, to the finished dQ
            }
            // dQ^T += K^T dS^T: slab jj = keys 16 jj .. of this 32-key half
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const bf8 df = p_frag(ds, 8 * jj);
                a0 = PCM_MFMA16(kt0[jj], df, a0);
                a1 = PCM_MFMA16(kt1[jj], df, a1);
            }
        }
        if (kt + 1 < ntiles) {
            stage_store(kr, smem + ((kt + 1) & 1) * TILE, tid);
            stage_store(vr, smem + (2 + ((kt + 1) & 1)) * TILE, tid);
        }
        __syncthreads();
    }
    if (rem) {  // wave 1's half of the key sum joins wave 0's
        float *part = reinterpret_cast<float *>(smem) + 2048;  // 8 KiB in: clear of wave 0's epilogue staging rows
        if (w == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r * 64 + lane] = a0[r], part[(16 + r) * 64 + lane] = a1[r];
        }
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] += part[r * 64 + lane], a1[r] += part[(16 + r) * 64 + lane];
    }
    store_acc_rows(smem + w * RW * OS, a0, a1, sk, dq + (long)b * dq_bs + h * HD, dq_ls, q0, P.L, lane);
}

// dK, dV: grid (B*H, ceil(S / 128)); streams Q / dO tiles (+ lse, delta, dropout row keys of the tile's queries)
template <bool DROP>  // dropout on / off is compiled in: as a run-time flag it left a branch around every masked element
__global__ __launch_bounds__(WG, 2) void pcm_attn_flash_bwd_dkv_kernel(AttnParams P, const u16 *__restrict__ dout, const float *__restrict__ lse,
                                                                    const float *__restrict__ delta, u16 *__restrict__ dk, long dk_bs,
                                                                    long dk_ls, u16 *__restrict__ dv, long dv_bs, long dv_ls)
{
    __shared__ __attribute__((aligned(16))) u16 smem[4 * TILE];  // Q[2] | dO[2]
    __shared__ __attribute__((aligned(16))) float rows_f[2][2][TR];  // [buffer][lse * log2 e | delta][query]
    __shared__ uint32_t rows_rb[2][TR];
    const int bh = blockIdx.x, b = bh / P.H, h = bh % P.H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hl = lane >> 5;
    const int E = P.H * HD;
    const int nfull = P.S / RWG, remr = P.S - nfull * RWG;
    const bool rem = remr > 0 && remr <= RW && (int)blockIdx.y == nfull;  // "rem" mode, see the header
    const int k0 = rem ? nfull * RWG : blockIdx.y * RWG + w * RW, key = k0 + (lane & 31);
    const unsigned char *mask = P.kpm ? P.kpm + (long)b * P.S : nullptr;
    const bool kin = key < P.S;
    const bool kok = kin && !(mask != nullptr && mask[key] != 0);
    const bool all_ok = __all(kok);  // wave-uniform: interior key blocks without padding skip the per-element select
    const LaneOffsets lo = lane_offsets(lane);
    bf8 kf[4], vf[4];
    {
        const u16 *kp = P.k + (long)b * P.k_bs + (long)key * P.k_ls + h * HD + 8 * hl;
        const u16 *vp = P.v + (long)b * P.v_bs + (long)key * P.v_ls + h * HD + 8 * hl;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            kf[sl] = as_bf8(kin ? *reinterpret_cast<const uint4 *>(kp + sl * 16) : make_uint4(0, 0, 0, 0));
            vf[sl] = as_bf8(kin ? *reinterpret_cast<const uint4 *>(vp + sl * 16) : make_uint4(0, 0, 0, 0));
        }
    }
    const DropCfg dc(P);
    const float scale2 = P.scale * 1.44269504088896f;
    const float sk = P.scale * dc.inv_keep;
    f16v dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) dk0[r] = 0.f, dk1[r] = 0.f, dv0[r] = 0.f, dv1[r] = 0.f;
    const u16 *qb = P.q + (long)b * P.q_bs + h * HD;
    const u16 *gb = dout + (long)b * P.L * E + h * HD;
    const int ntiles = (P.L + TR - 1) / TR;
    uint4 qr[2], gr[2];
    float rl = 0.f, rd = 0.f;
    uint32_t rrb = 0u;
    auto fetch_rows = [&](int t) {
        if (tid < TR) {
            const int row = t * TR + tid;
            rl = row < P.L ? lse[(long)bh * P.L + row] * 1.44269504088896f : INFINITY;
            rd = row < P.L ? delta[(long)bh * P.L + row] / dc.inv_keep : 0.f;  // D * (1 - p_drop), see the dQ kernel
            rrb = DROP ? attn_rowbase(dc.seed, P.site, (uint32_t)(bh * P.L + row)) : 0u;
        }
    };
    auto store_rows = [&](int buf) {
        if (tid < TR) rows_f[buf][0][tid] = rl, rows_f[buf][1][tid] = rd, rows_rb[buf][tid] = rrb;
    };
    stage_fetch(qr, qb, P.q_ls, 0, P.L, tid);
    stage_fetch(gr, gb, E, 0, P.L, tid);
    fetch_rows(0);
    stage_store(qr, smem, tid);
    stage_store(gr, smem + 2 * TILE, tid);
    store_rows(0);
    __syncthreads();
    for (int qt = 0; qt < ntiles; ++qt) {
        const int cur = qt & 1;
        const u16 *Qt = smem + cur * TILE, *Gt = smem + (2 + cur) * TILE;
        if (qt + 1 < ntiles) {
            stage_fetch(qr, qb, P.q_ls, (qt + 1) * TR, P.L, tid);
            stage_fetch(gr, gb, E, (qt + 1) * TR, P.L, tid);
            fetch_rows(qt + 1);
        }
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {  // 32 queries at a time
            if (rem && qh != w) continue;  // rem mode: wave 0 / 1 owns query half 0 / 1, waves 2 and 3 only stage
            f16v s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                s = PCM_MFMA16(lds_bf8(Qt + lo.row[sl] + qh * 32 * HD), kf[sl], s);  // rows = queries, columns = keys
                dp = PCM_MFMA16(lds_bf8(Gt + lo.row[sl] + qh * 32 * HD), vf[sl], dp);
            }
            float pd[16], ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qr0 = 32 * qh + 8 * g + 4 * hl;  // registers 4g .. 4g+3 = queries qr0 .. qr0+3 of the tile
                const float4 l4 = *reinterpret_cast<const float4 *>(&rows_f[cur][0][qr0]);
                const float4 d4 = *reinterpret_cast<const float4 *>(&rows_f[cur][1][qr0]);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
                // Dropout bits.  Here a lane is a KEY and a register a query, so the two keys that share a hash word sit in
                // adjacent lanes: the even lane hashes query 2jj, the odd lane query 2jj+1 (half the hashes of one per element),
                // and the four keep bits of the 2 x 2 block are routed to their (register, lane) as wave masks on the scalar unit.
                lanemask km[4] = {0, 0, 0, 0};
                if (DROP) {
                    constexpr lanemask EV = 0x5555555555555555ull, OD = 0xAAAAAAAAAAAAAAAAull;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const uint32_t bits = attn_pair_bits(rows_rb[cur][qr0 + 2 * jj + (lane & 1)], (uint32_t)key >> 1);
                        lanemask clo, chi;  // even lanes: (query 2jj, own key | next key); odd lanes: (query 2jj+1, previous key | own key)
                        keep_masks(bits, dc.thr, clo, chi);
                        km[2 * jj] = (clo & EV) | ((chi & EV) << 1);
                        km[2 * jj + 1] = (chi & OD) | ((clo & OD) >> 1);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g + i;
                    float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale2, -lv[i]));
                    if (!all_ok) pr = kok ? pr : 0.f;
                    float dpv = dp[r];
                    pd[r] = pr;
                    if (DROP) {
                        pd[r] = keep_if(km[i], pr);  // dV is rescaled by 1 / (1 - p_drop) once, when it is stored
                        dpv = keep_if(km[i], dpv);
                    }
                    ds[r] = pr * (dpv - dv4[i]);  // scale / (1 - p_drop): applied once, to the finished dK
                }
            }
            // dV^T += dO^T P, dK^T += Q^T dS: slab jq = queries 16 jq .. of this 32-query half
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                const bf8 pf = p_frag(pd, 8 * jq), df = p_frag(ds, 8 * jq);
                const int so = (2 * qh + jq) * 16 * HD;
                dv0 = PCM_MFMA16(cat8(lds_tr(Gt + so + lo.tr[0][0]), lds_tr(Gt + so + lo.tr[1][0])), pf, dv0);
                dv1 = PCM_MFMA16(cat8(lds_tr(Gt + so + lo.tr[0][1]), lds_tr(Gt + so + lo.tr[1][1])), pf, dv1);
                dk0 = PCM_MFMA16(cat8(lds_tr(Qt + so + lo.tr[0][0]), lds_tr(Qt + so + lo.tr[1][0])), df, dk0);
                dk1 = PCM_MFMA16(cat8(lds_tr(Qt + so + lo.tr[0][1]), lds_tr(Qt + so + lo.tr[1][1])), df, dk1);
            }
        }
        if (qt + 1 < ntiles) {
            stage_store(qr, smem + (cur ^ 1) * TILE, tid);
            stage_store(gr, smem + (2 + (cur ^ 1)) * TILE, tid);
            store_rows(cur ^ 1);
        }
        __syncthreads();
    }
    if (rem) {  // wave 1's half of the query sums joins wave 0's
        float *part = reinterpret_cast<float *>(smem) + 2048;  // 8 KiB in: clear of wave 0's epilogue staging rows
        if (w == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                part[r * 64 + lane] = dk0[r], part[(16 + r) * 64 + lane] = dk1[r];
                part[(32 + r) * 64 + lane] = dv0[r], part[(48 + r) * 64 + lane] = dv1[r];
            }
        }
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk0[r] += part[r * 64 + lane], dk1[r] += part[(16 + r) * 64 + lane];
            dv0[r] += part[(32 + r) * 64 + lane], dv1[r] += part[(48 + r) * 64 + lane];
        }
    }
    u16 *stage = smem + w * RW * OS;
    store_acc_rows(stage, dk0, dk1, sk, dk + (long)b * dk_bs + h * HD, dk_ls, k0, P.S, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    store_acc_rows(stage, dv0, dv1, dc.inv_keep, dv + (long)b * dv_bs + h * HD, dv_ls, k0, P.S, lane);
}

inline bool strides_ok(long bs, long ls)
{
    return bs % 8 == 0 && ls % 8 == 0;  // 16-byte row loads
}

}  // namespace

extern "C" int pcm_attn_flash_supported(int L, int S, int head_dim)
{
    return (head_dim == HD && L >= 1 && S >= 1) ? 1 : 0;
}

extern "C" int pcm_attn_flash_forward_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k,
                                          long k_bs, long k_ls, const void *v, long v_bs, long v_ls,
                                          const unsigned char *key_padding_mask, float scale, float p_drop, const long *seed,
                                          unsigned site, void *out, float *lse, void *stream)
{
    if (B <= 0 || H <= 0) return B == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (!pcm_attn_flash_supported(L, S, HD)) return PCM_ERR_UNSUPPORTED;
    if (!strides_ok(q_bs, q_ls) || !strides_ok(k_bs, k_ls) || !strides_ok(v_bs, v_ls)) return PCM_ERR_BAD_ARG;
    if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && seed == nullptr)) return PCM_ERR_BAD_ARG;
    AttnParams P{(const u16 *)q, (const u16 *)k, (const u16 *)v, q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, key_padding_mask,
                 B, H, L, S, scale, p_drop, seed, site};
    if (p_drop > 0.f)
        hipLaunchKernelGGL(pcm_attn_flash_fwd_kernel<true>, dim3(B * H, (L + RWG - 1) / RWG), dim3(WG), 0, (hipStream_t)stream, P, (u16 *)out, lse);
    else
        hipLaunchKernelGGL(pcm_attn_flash_fwd_kernel<false>, dim3(B * H, (L + RWG - 1) / RWG), dim3(WG), 0, (hipStream_t)stream, P, (u16 *)out, lse);
    return PCM_LAUNCH_STATUS();
}

// delta: (B, H, L) fp32 workspace (written here).  dq / dk / dv: bf16 with the given batch / row strides.
// stage_mask: 1 delta | 2 dK, dV | 4 dQ (bench.py times the kernels one at a time; <= 0: all)
extern "C" int pcm_attn_flash_backward_stages_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k,
                                                  long k_bs, long k_ls, const void *v, long v_bs, long v_ls,
                                                  const unsigned char *key_padding_mask, float scale, float p_drop, const long *seed,
                                                  unsigned site, const void *out, const void *dout, const float *lse, float *delta,
                                                  void *dq, long dq_bs, long dq_ls, void *dk, long dk_bs, long dk_ls, void *dv,
                                                  long dv_bs, long dv_ls, int stage_mask, void *stream)
{
    if (B <= 0 || H <= 0) return B == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    if (!pcm_attn_flash_supported(L, S, HD)) return PCM_ERR_UNSUPPORTED;
    if (!strides_ok(q_bs, q_ls) || !strides_ok(k_bs, k_ls) || !strides_ok(v_bs, v_ls)) return PCM_ERR_BAD_ARG;
    if (dq_ls % 8 || dk_ls % 8 || dv_ls % 8 || dq_bs % 8 || dk_bs % 8 || dv_bs % 8) return PCM_ERR_BAD_ARG;
    if (stage_mask <= 0) stage_mask = 7;
    AttnParams P{(const u16 *)q, (const u16 *)k, (const u16 *)v, q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, key_padding_mask,
                 B, H, L, S, scale, p_drop, seed, site};
    hipStream_t st = (hipStream_t)stream;
    const long groups = (long)B * L * H * 8;
    long pblocks = (groups + WG - 1) / WG;
    if (pblocks > 4096) pblocks = 4096;
    if (stage_mask & 1)
        hipLaunchKernelGGL(pcm_attn_flash_prep_kernel, dim3((int)pblocks), dim3(WG), 0, st, B, H, L, (const u16 *)out, (const u16 *)dout, delta);
    if (stage_mask & 2)
        hipLaunchKernelGGL(p_drop > 0.f ? pcm_attn_flash_bwd_dkv_kernel<true> : pcm_attn_flash_bwd_dkv_kernel<false>, dim3(B * H, (S + RWG - 1) / RWG),
                           dim3(WG), 0, st, P, (const u16 *)dout, lse, delta, (u16 *)dk, dk_bs, dk_ls, (u16 *)dv, dv_bs, dv_ls);
    if (stage_mask & 4)
        hipLaunchKernelGGL(p_drop > 0.f ? pcm_attn_flash_bwd_dq_kernel<true> : pcm_attn_flash_bwd_dq_kernel<false>, dim3(B * H, (L + RWG - 1) / RWG),
                           dim3(WG), 0, st, P, (const u16 *)dout, lse, delta, (u16 *)dq, dq_bs, dq_ls);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_attn_flash_backward_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k,
                                           long k_bs, long k_ls, const void *v, long v_bs, long v_ls,
                                           const unsigned char *key_padding_mask, float scale, float p_drop, const long *seed,
                                           unsigned site, const void *out, const void *dout, const float *lse, float *delta,
                                           void *dq, long dq_bs, long dq_ls, void *dk, long dk_bs, long dk_ls, void *dv,
                                           long dv_bs, long dv_ls, void *stream)
{
    return pcm_attn_flash_backward_stages_hip(B, H, L, S, q, q_bs, q_ls, k, k_bs, k_ls, v, v_bs, v_ls, key_padding_mask, scale, p_drop, seed,
                                              site, out, dout, lse, delta, dq, dq_bs, dq_ls, dk, dk_bs, dk_ls, dv, dv_bs, dv_ls, 0, stream);
}
