// proj_ln.hip -- out = LayerNorm(x + dropout(a W^T + b)) as ONE kernel on the matrix cores (gfx950, v_mfma_f32_16x16x32_bf16).
//
// The tail of every attention sub-layer of the ACT transformer
//   (/root/reference/src/models/components/act/transformer.py:244-256 encoder, :296-346 decoder: `src2 = self.self_attn(...)[0]` -- whose
//    last step is the output projection of nn.MultiheadAttention -- `src = src + self.dropout1(src2); src = self.norm1(src)`)
// ran as a library GEMM (800 x 512 x 512 in the decoder: 6-9 us, 2 % of the MFMA peak: a launch-floor-bound product) followed by
// csrc/drln.hip (7.5 us, reads the GEMM's output back from HBM / L2).  22 such pairs per ACT step.
//
// Design, for the ~800-row case ("a row tile's whole K = 512 A panel resident in LDS and B streamed once per workgroup"):
//  * one workgroup = 16 rows x ALL E columns (LayerNorm needs whole rows), 8 waves; wave w owns columns [w E/8, (w+1) E/8);
//  * the A panel (16 x K bf16, <= 32 KiB) is loaded ONCE into LDS (rows padded by 8 elements: the 16-byte operand reads of the 16
//    rows fall into different banks); the weight is never staged: the B operand of the 16x16x32 instruction is 8 consecutive k of one
//    output column = 16 contiguous bytes of W's row (W is (E, K) row-major, the nn.Linear layout) -- each lane fetches its operand
//    straight from L2 into registers, two k-steps ahead of the matrix instruction that consumes it (W = 512 KiB stays L2 resident;
//    per workgroup it is streamed exactly once: 16 k-steps x 4 column tiles x 16 bytes per lane);
//  * the accumulators (+ bias, rounded to bf16 like the library GEMM's output under autocast) go to an LDS tile, ONE barrier, then
//    each wave finishes two rows exactly like csrc/drln.hip: s = x + keep * y / (1 - p) with the same counter-based mask
//    (so pcm_drln_backward2_hip is this kernel's backward), row mean / variance by wave reductions, the affine map, and the
//    consumer's bf16 operands (out + pos, out) in the same pass.  All global traffic of the epilogue is whole-row, 16 bytes per lane.
// Bound: streaming W through one CU's L2 port (E K 2 bytes / 64 B per clock = 3.4 us at E = K = 512) -- about the time of the GEMM
// alone today, with the second kernel and the round trip of y gone.
// Algorithmic bytes per row: 2 K (a) + 4 E (x) + 8 E (s, out) [+ 4 E (sum16, out16)]; the weight E K 2 once per 16 rows from L2.
#include "pcm_attn.hpp"

namespace {

constexpr int kTM = 16;          // rows per workgroup
constexpr int kWavesP = 8;       // waves per workgroup
constexpr int kThreadsP = 64 * kWavesP;
constexpr int kAPad = 8;         // bf16 elements of padding per A row
constexpr int kYPad = 4;         // floats of padding per y row

typedef float f4v __attribute__((ext_vector_type(4)));
#define PCM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ bf8 ldg_bf8(const u16 *p)  // 16 bytes, global
{
    return as_bf8(*reinterpret_cast<const uint4 *>(p));
}

// Row offset of activation row r inside a position table of pos_rows rows (pos_rows >= 1; r0 % pos_rows is taken once per workgroup,
// a row of the tile adds at most a few wraps: no 64-bit division per element).
__device__ __forceinline__ long pos_row(long base, int i, long pos_rows)
{
    long pr = base + i;
    while (pr >= pos_rows) pr -= pos_rows;
    return pr;
}

// A panel (TM rows of K bf16, rows past R zero) -> LDS with ALL of a thread's 16-byte loads in flight before the first LDS store.
// The first version's `for (c = tid; c < TM * chunks; c += T) { load; store }` compiled to one load, `s_waitcnt vmcnt(0)`, one store
// per iteration: 2 (projection, 16 rows) to 16 (linear, 64 rows) L2 round trips in series before the first matrix instruction.
// Fast path: the threads tile the panel as (T / cpr rows) x (cpr pieces), cpr = K / 8 pieces per row, when cpr divides T (K = 256,
// 512, 1024 ...); thread (i0, kc) then owns rows i0, i0 + T / cpr, ... of piece kc: no division inside the loop.
template <int TM, int T, int U>
__device__ __forceinline__ void panel_bf16(const u16 *__restrict__ a, long a_ls, long r0, long R, int K, u16 *__restrict__ As, int AS, int tid)
{
    const int cpr = K / 8;
    if (T % cpr == 0) {
        const int rpp = T / cpr, i0 = tid / cpr, kc = tid - i0 * cpr;
        for (int ib = i0; ib < TM; ib += U * rpp) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = ib + u * rpp;
                v[u] = make_uint4(0, 0, 0, 0);
                if (i < TM && r0 + i < R) v[u] = *reinterpret_cast<const uint4 *>(a + (r0 + i) * a_ls + kc * 8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = ib + u * rpp;
                if (i < TM) *reinterpret_cast<uint4 *>(As + i * AS + kc * 8) = v[u];
            }
        }
    } else {  // any K % 32 == 0
        for (int c = tid; c < TM * cpr; c += T) {
            const int i = c / cpr, kc = c % cpr;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r0 + i < R) v = *reinterpret_cast<const uint4 *>(a + (r0 + i) * a_ls + kc * 8);
            *reinterpret_cast<uint4 *>(As + i * AS + kc * 8) = v;
        }
    }
}

// MT = 16-row tiles per workgroup: 1 for the ~800-row sites (50 workgroups, the weight streamed 50 times), 4 for the long ones (the
// encoder's 4120 / 8216 / 16408 rows: every B operand fetched from L2 feeds FOUR matrix instructions, the weight streams R / 64 times;
// the product tile then waits in LDS as bf16 -- the value it is rounded to anyway -- to fit 64 rows into the CU's 160 KiB).
template <int E, int MT>
__global__ __launch_bounds__(kThreadsP) void pcm_proj_drln_fwd_kernel(
    long R, int K, const u16 *__restrict__ a, long a_ls, const u16 *__restrict__ W, const void *__restrict__ bias, int bias_is_bf16,
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float p_drop,
    const long *__restrict__ seed_ptr, unsigned site, float *__restrict__ s_out, float *__restrict__ out, float *__restrict__ mean_out,
    float *__restrict__ rstd_out, const float *__restrict__ pos, long pos_rows, __hip_bfloat16 *__restrict__ sum16,
    __hip_bfloat16 *__restrict__ x16)
{
    // every argument in registers at the entry: one batch of kernarg loads ("Kernel heads", pcm_common.hpp)
    asm volatile("" ::"s"(R), "s"(K), "s"(a), "s"(a_ls), "s"(W), "s"(bias), "s"(bias_is_bf16), "s"(x), "s"(gamma), "s"(beta), "s"(eps), "s"(p_drop), "s"(seed_ptr), "s"(site), "s"(s_out), "s"(out), "s"(mean_out), "s"(rstd_out), "s"(pos), "s"(pos_rows), "s"(sum16), "s"(x16));
    constexpr int TM = kTM * MT;            // rows per workgroup
    constexpr int RPW = TM / kWavesP;       // rows a wave finishes in the row phase (2 or 8)
    constexpr bool EARLY = MT == 1;         // the rows' x values are requested BEFORE the products (16 registers); MT = 4: behind them
    constexpr int NT = E / (16 * kWavesP);  // 16-column tiles per wave
    constexpr int NCH = E / 256;            // float4 chunks per lane in the row phase
    constexpr bool YB = MT > 1;             // product tile kept as bf16
    constexpr int YS = E + (YB ? 8 : kYPad);  // elements per row of the product tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int AS = K + kAPad;
    u16 *As = reinterpret_cast<u16 *>(smem);                                   // [TM][AS] bf16
    unsigned char *ybase = smem + (size_t)TM * AS * 2;                         // 16-byte aligned: AS % 8 == 0
    float *Ys = reinterpret_cast<float *>(ybase);                              // [TM][YS] fp32   (MT == 1)
    u16 *Yb = reinterpret_cast<u16 *>(ybase);                                  // [TM][YS] bf16   (MT > 1)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long r0 = (long)blockIdx.x * TM;

    // ---- products: wave w, column tiles t = 0 .. NT-1 at n0 + 16 t; operands of k-step kt: k = 32 kt + 8 (lane / 16) .. + 7
    const int n0 = w * (E / kWavesP), li = lane & 15, lk = 8 * (lane >> 4);
    const u16 *wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wrow[t] = W + (long)(n0 + 16 * t + li) * K + lk;
    f4v acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f4v{0.f, 0.f, 0.f, 0.f};
    const int ksteps = K / 32;
    // Three operand buffers in rotation, the loop unrolled by three so that every buffer index is a compile-time constant: the
    // operands of k-step kt + 2 are requested before step kt is multiplied and nothing is ever moved between registers (a rotation by
    // moves forces a full `s_waitcnt vmcnt(0)` per step: seen in the first version's ISA).  Loads past the last step re-read it.
    bf8 bq[3][NT];
    auto fetch = [&](bf8(&dst)[NT], int kt) {
        const int kk = kt < ksteps ? kt : ksteps - 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[t] = ldg_bf8(wrow[t] + 32 * kk);
    };
    auto multiply = [&](const bf8(&cur)[NT], int kt) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bf8 af = lds_bf8(As + (16 * m + li) * AS + 32 * kt + lk);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = PCM_MFMA_16x16x32(af, cur[t], acc[m][t]);
        }
    };
    fetch(bq[0], 0);
    fetch(bq[1], 1);
    // ---- everything the epilogue reads from global memory is requested HERE, under the product phase (the first version asked for it
    // row by row after the products: one exposed L2 / HBM round trip per row and wave).  x of this wave's rows (EARLY), the bias of its
    // columns, gamma / beta of the row phase's columns.
    uint32_t braw[NT];  // raw bits; converted where they are used (a conversion here would wait for each load in turn)
    if (bias == nullptr) {
#pragma unroll
        for (int t = 0; t < NT; ++t) braw[t] = 0u;
    } else if (bias_is_bf16) {
#pragma unroll
        for (int t = 0; t < NT; ++t) braw[t] = reinterpret_cast<const u16 *>(bias)[n0 + 16 * t + li];
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) braw[t] = reinterpret_cast<const uint32_t *>(bias)[n0 + 16 * t + li];
    }
    float xs[RPW][NCH][4], gs[NCH][4], bs[NCH][4];
    auto fetch_rows = [&]() {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const long r = r0 + w + kWavesP * j;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (r < R) load4<float>(x + r * E + c * 256 + lane * 4, xs[j][c]);
                else xs[j][c][0] = xs[j][c][1] = xs[j][c][2] = xs[j][c][3] = 0.f;
            }
        }
    };
    if (EARLY) fetch_rows();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        load4<float>(gamma + c * 256 + lane * 4, gs[c]);
        load4<float>(beta + c * 256 + lane * 4, bs[c]);
    }
    // ---- A panel -> LDS (rows past R are zero), every load of a thread in flight at once -- and BEHIND the first weight operands and
    // the epilogue's inputs in program order, so that the panel's round trip is theirs too (one exposed latency before the products)
    panel_bf16<TM, kThreadsP, (MT == 1 ? 2 : 8)>(a, a_ls, r0, R, K, As, AS, tid);
    __syncthreads();  // A panel complete
    int kt = 0;
    for (; kt + 2 < ksteps; kt += 3) {
        fetch(bq[2], kt + 2);
        multiply(bq[0], kt);
        fetch(bq[0], kt + 3);
        multiply(bq[1], kt + 1);
        fetch(bq[1], kt + 4);
        multiply(bq[2], kt + 2);
    }
    if (kt < ksteps) {
        multiply(bq[0], kt);
        if (kt + 1 < ksteps) multiply(bq[1], kt + 1);
    }

    // ---- y = bf16(acc + bias) -> LDS tile.  Accumulator register r of `lane`, row tile m: row 16 m + 4 (lane / 16) + r,
    // column n0 + 16 t + lane % 16
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = n0 + 16 * t + li;
        const float bv = __uint_as_float(bias != nullptr && bias_is_bf16 ? braw[t] << 16 : braw[t]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t yb = pcm_cvt_pk_bf16(acc[m][t][r] + bv, 0.f) & 0xFFFFu;  // rounded to bf16
                const int row = 16 * m + 4 * (lane >> 4) + r;
                if (YB) Yb[row * YS + col] = (u16)yb;
                else Ys[row * YS + col] = __uint_as_float(yb << 16);
            }
    }
    if (!EARLY) fetch_rows();  // the accumulators' registers are free now: all RPW rows' x in flight across the barrier
    __syncthreads();

    // ---- rows: wave w finishes rows w, w + 8, ... of the tile (csrc/drln.hip's row code on the LDS-resident y).  Straight-line code
    // over the wave's RPW rows -- rows past R compute on zeros and store nothing -- so that the rows' dependent chains (hash, two
    // wave reductions each) interleave instead of running one after another.
    const bool drop = p_drop > 0.f;
    const uint64_t seed = drop ? (uint64_t)seed_ptr[0] : 0ull;
    const uint32_t thr = drop ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
    const float scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    const long pbase = sum16 != nullptr ? r0 % pos_rows : 0;  // wave-uniform, once
    float mus[RPW], rstds[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = w + kWavesP * j;
        const long r = r0 + i;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 256 + lane * 4;
            const long e0 = r * E + col;
            float yv[4];
            if (YB) load4<__hip_bfloat16>(reinterpret_cast<const __hip_bfloat16 *>(Yb + i * YS + col), yv);
            else load4<float>(Ys + i * YS + col, yv);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float yy = keep_elem(seed, site, (uint64_t)(e0 + v), thr) ? yv[v] * scale : 0.f;
                xs[j][c][v] = xs[j][c][v] + yy;  // s = x + dropout(y)
                sum += xs[j][c][v];
            }
        }
        mus[j] = wave_sum(sum) * (1.f / E);
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float d = xs[j][c][v] - mus[j];
                sq += d * d;
            }
        rstds[j] = rsqrtf(wave_sum(sq) * (1.f / E) + eps);
    }
    // the emitted operand's position rows: requested for a GROUP of the wave's rows before the group's first store (inside the per-row
    // `if` each load was waited for on the spot: 2 x RPW round trips in series); groups of 4 rows for the 64-row tile (all 8 at once
    // spilled); a row past R reads the position row of the tile's first row
    constexpr int G = RPW < 4 ? RPW : 4;
#pragma unroll
    for (int g0 = 0; g0 < RPW; g0 += G) {
        float ps[G][NCH][4];
        if (sum16 != nullptr) {
#pragma unroll
            for (int jj = 0; jj < G; ++jj) {
                const int i = w + kWavesP * (g0 + jj);
                const long prow = pos_row(pbase, r0 + i < R ? i : 0, pos_rows);
#pragma unroll
                for (int c = 0; c < NCH; ++c) load4<float>(pos + prow * E + c * 256 + lane * 4, ps[jj][c]);
            }
        }
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const int j = g0 + jj;
            const int i = w + kWavesP * j;
            const long r = r0 + i;
            if (r < R) {  // wave-uniform
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int col = c * 256 + lane * 4;
                    const long e0 = r * E + col;
                    float o[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) o[v] = (xs[j][c][v] - mus[j]) * rstds[j] * gs[c][v] + bs[c][v];
                    store4<float>(s_out + e0, xs[j][c]);
                    store4<float>(out + e0, o);
                    if (sum16 != nullptr) {
                        float q[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) q[v] = o[v] + ps[jj][c][v];
                        store4<__hip_bfloat16>(sum16 + e0, q);
                    }
                    if (x16 != nullptr) store4<__hip_bfloat16>(x16 + e0, o);
                }
                if (lane == 0) mean_out[r] = mus[j], rstd_out[r] = rstds[j];
            }
        }
    }
}

inline size_t proj_smem_bytes(int E, int K, int MT)
{
    const size_t rows = (size_t)kTM * MT;
    return rows * (K + kAPad) * 2 + (MT > 1 ? rows * (E + 8) * 2 : rows * (E + kYPad) * 4);
}

// rows from which the 64-row tile is taken (below: 16-row tiles, more workgroups for the short activations)
constexpr long kLongRows = 2048;

template <int E, int MT>
int launch_proj(long R, int K, const void *a, long a_ls, const void *W, const void *bias, int bias_is_bf16, const float *x,
                const float *gamma, const float *beta, float eps, float p_drop, const long *seed, unsigned site, float *s, float *out,
                float *mean, float *rstd, const float *pos, long pos_n, void *sum16, void *x16, hipStream_t st)
{
    const size_t smem = proj_smem_bytes(E, K, MT);
    const long rows = (long)kTM * MT, blocks = (R + rows - 1) / rows;
    if (smem > 64 * 1024) {  // E = K = 1024, or the 64-row tile: up to 133 KiB (E = K = 512) of the CU's 160
        const int rc = pcm_status(hipFuncSetAttribute(reinterpret_cast<const void *>(pcm_proj_drln_fwd_kernel<E, MT>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (rc) return rc;
    }
    hipLaunchKernelGGL((pcm_proj_drln_fwd_kernel<E, MT>), dim3((unsigned)blocks), dim3(kThreadsP), smem, st, R, K, (const u16 *)a, a_ls,
                       (const u16 *)W, bias, bias_is_bf16, x, gamma, beta, eps, p_drop, seed, site, s, out, mean, rstd, pos,
                       pos != nullptr && pos_n >= E ? pos_n / E : 1, (__hip_bfloat16 *)sum16, (__hip_bfloat16 *)x16);
    return PCM_LAUNCH_STATUS();
}

}  // namespace

// E: multiples of 256 up to 1024 (the row phase's float4 chunks; E / 128 column tiles per wave); K: multiples of 32 up to 1024
extern "C" int pcm_proj_drln_mfma_supported(int E, int K)
{
    return (E == 256 || E == 512 || E == 768 || E == 1024) && K >= 32 && K <= 1024 && K % 32 == 0;
}

extern "C" int pcm_proj_drln_mfma_forward_hip(long R, int E, int K, const void *a_bf16, long a_ls, const void *w_bf16, const void *bias,
                                              int bias_is_bf16, const float *x, const float *gamma, const float *beta, float eps,
                                              float p_drop, const long *seed, unsigned site, float *s, float *out, float *mean,
                                              float *rstd, const float *pos, long pos_n, void *sum_bf16, void *out_bf16, void *stream)
{
    if (R < 0 || E <= 0 || K <= 0 || a_ls < K || !(p_drop >= 0.f && p_drop < 1.f)) return PCM_ERR_BAD_ARG;
    if (!pcm_proj_drln_mfma_supported(E, K)) return PCM_ERR_UNSUPPORTED;
    if (R == 0) return PCM_OK;
    if (!a_bf16 || !w_bf16 || !x || !gamma || !beta || !s || !out || !mean || !rstd || (p_drop > 0.f && !seed)) return PCM_ERR_BAD_ARG;
    if (sum_bf16 != nullptr && (pos == nullptr || pos_n <= 0 || pos_n % E != 0)) return PCM_ERR_BAD_ARG;
    if ((a_ls % 8) != 0 || (((uintptr_t)a_bf16 | (uintptr_t)w_bf16) % 16) != 0) return PCM_ERR_BAD_ARG;  // 16-byte operand loads
    hipStream_t st = (hipStream_t)stream;
    // the 64-row tile needs rows * (K + 8) * 2 + rows * (E + 8) * 2 bytes of LDS: E + K <= 1232 (E = K = 512: 133 KiB); E = 768 / 1024 keep the
    // 16-row tile (eight rows of 768 values per wave in flight for the row phase, or four row tiles of 8 column tiles, need more than
    // 256 registers: the 768-wide 64-row variant spilled 44 bytes)
    const bool wide = R >= kLongRows && E <= 512 && (size_t)proj_smem_bytes(E, K, 4) <= 160 * 1024;
#define PCM_PROJ(EE, MM)                                                                                                             \
    return launch_proj<EE, MM>(R, K, a_bf16, a_ls, w_bf16, bias, bias_is_bf16, x, gamma, beta, eps, p_drop, seed, site, s, out, mean, \
                               rstd, pos, pos_n, sum_bf16, out_bf16, st)
    switch (E) {
    case 256:
        if (wide) PCM_PROJ(256, 4);
        PCM_PROJ(256, 1);
    case 512:
        if (wide) PCM_PROJ(512, 4);
        PCM_PROJ(512, 1);
    case 768:
        PCM_PROJ(768, 1);
    default:
        PCM_PROJ(1024, 1);
    }
#undef PCM_PROJ
}

// ================================================================================================================================
// pcm_linear_mfma: out = A W^T + bias for SHORT activations (R up to ~1000 rows: the decoder's and the CVAE encoder's in-projections
// and the cross-attention query projection, transformer.py:244-262, 296-346), with the operand preparation fused in:
//   * A is either bf16 (one matrix, or two: `a` for the output columns below `pos_cols`, `a_alt` for the others -- the operands a
//     producer kernel already emitted), or fp32 x with the position embedding added on the way into LDS -- bf16(x + pos) for the
//     columns below `pos_cols` (q and k of nn.MultiheadAttention's packed in-projection), bf16(x) for the others (v) -- which is what
//     csrc/tokens.hip's add + cast kernels and the doubled-row product `[x + pos ; x] W^T` did in two launches and twice the FLOPs;
//     the two bf16 operands can be written out as well (emit_*: the backward's weight-gradient products need them);
//   * same streaming scheme as the kernel above: a 16-row A panel in LDS, the weight's rows straight from L2 into the 16x16x32 B
//     operands two k-steps ahead; no LayerNorm here, so the columns are split too: one workgroup = 16 rows x 256 columns, 4 waves;
//   * the accumulators (+ bias) are rounded once to the output type and leave through LDS as whole 16-byte pieces of a row.
// ================================================================================================================================
namespace {

constexpr int kLW = 4;             // waves per workgroup
constexpr int kLThreads = 64 * kLW;
constexpr int kLN = 256;           // columns per workgroup
constexpr int kLT = kLN / (16 * kLW);  // 16-column tiles per wave (4)

// MT = 16-row tiles per workgroup (1: short activations; 4: from 2048 rows on, each weight operand feeds four matrix instructions)
template <bool OUT_BF16, int MT>
__global__ __launch_bounds__(kLThreads) void pcm_linear_mfma_kernel(long R, int N, int K, const void *__restrict__ a, int a_is_f32,
                                                                    long a_ls, const u16 *__restrict__ a_alt, const float *__restrict__ pos,
                                                                    long pos_rows, int pos_cols, const u16 *__restrict__ W,
                                                                    const void *__restrict__ bias, int bias_is_bf16, void *__restrict__ out,
                                                                    long out_ls, u16 *__restrict__ emit_pos16, u16 *__restrict__ emit_x16)
{
    asm volatile("" ::"s"(R), "s"(N), "s"(K), "s"(a), "s"(a_is_f32), "s"(a_ls), "s"(a_alt), "s"(pos), "s"(pos_rows), "s"(pos_cols), "s"(W), "s"(bias), "s"(bias_is_bf16), "s"(out), "s"(out_ls), "s"(emit_pos16), "s"(emit_x16));  // "Kernel heads", pcm_common.hpp
    constexpr int TM = kTM * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    const int AS = K + kAPad;
    u16 *As = reinterpret_cast<u16 *>(smem2);  // [TM][AS] bf16; reused for the output tile after the products
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long r0 = (long)blockIdx.x * TM;
    const int c0 = blockIdx.y * kLN;           // first output column of this workgroup
    const bool below = c0 < pos_cols;          // this workgroup's columns see the FIRST operand (x + pos, or `a`)
    const bool with_pos = a_is_f32 && pos != nullptr && below;
    // the bf16 operands are the backward's weight-gradient operands: written out once, by the first column block of either kind
    u16 *emit = nullptr;
    if (a_is_f32) emit = with_pos ? (blockIdx.y == 0 ? emit_pos16 : nullptr) : ((c0 == (pos != nullptr ? pos_cols : 0)) ? emit_x16 : nullptr);

    // ---- products (columns past N: the weight row index is clamped, the results are not stored)
    const int n0 = c0 + w * (kLN / kLW), li = lane & 15, lk = 8 * (lane >> 4);
    const u16 *wrow[kLT];
#pragma unroll
    for (int t = 0; t < kLT; ++t) {
        int col = n0 + 16 * t + li;
        col = col < N ? col : N - 1;
        wrow[t] = W + (long)col * K + lk;
    }
    f4v acc[MT][kLT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < kLT; ++t) acc[m][t] = f4v{0.f, 0.f, 0.f, 0.f};
    const int ksteps = K / 32;
    bf8 bq[3][kLT];  // three operand buffers in rotation, loop unrolled by three: see pcm_proj_drln_fwd_kernel
    auto fetch = [&](bf8(&dst)[kLT], int kt) {
        const int kk = kt < ksteps ? kt : ksteps - 1;
#pragma unroll
        for (int t = 0; t < kLT; ++t) dst[t] = ldg_bf8(wrow[t] + 32 * kk);
    };
    auto multiply = [&](const bf8(&cur)[kLT], int kt) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bf8 af = lds_bf8(As + (16 * m + li) * AS + 32 * kt + lk);
#pragma unroll
            for (int t = 0; t < kLT; ++t) acc[m][t] = PCM_MFMA_16x16x32(af, cur[t], acc[m][t]);
        }
    };
    fetch(bq[0], 0);
    fetch(bq[1], 1);
    // the bias of this lane's columns, requested under the products (was: round trips behind them); raw bits, converted at the use
    uint32_t braw[kLT];
    if (bias == nullptr) {
#pragma unroll
        for (int t = 0; t < kLT; ++t) braw[t] = 0u;
    } else if (bias_is_bf16) {
#pragma unroll
        for (int t = 0; t < kLT; ++t) {
            const int col = n0 + 16 * t + li;
            braw[t] = reinterpret_cast<const u16 *>(bias)[col < N ? col : N - 1];
        }
    } else {
#pragma unroll
        for (int t = 0; t < kLT; ++t) {
            const int col = n0 + 16 * t + li;
            braw[t] = reinterpret_cast<const uint32_t *>(bias)[col < N ? col : N - 1];
        }
    }
    // ---- A panel -> LDS as bf16 (rows past R are zero); every load of a thread in flight before the first conversion / LDS store, and
    // behind the first weight operands in program order (one exposed round trip before the products, not two)
    constexpr int U = 8;
    if (a_is_f32) {
        const float *x = reinterpret_cast<const float *>(a);
        const int cpr = K / 4;  // float4 pieces per row
        const long pbase = with_pos ? r0 % pos_rows : 0;  // once per workgroup (the first version: a 64-bit modulo per piece)
        auto convert = [&](int i, int kc, float (&v)[4], const float (&p)[4]) {
            if (with_pos) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] += p[u];
            }
            const uint2 pk = make_uint2(pcm_cvt_pk_bf16(v[0], v[1]), pcm_cvt_pk_bf16(v[2], v[3]));
            *reinterpret_cast<uint2 *>(As + i * AS + kc * 4) = pk;
            if (emit != nullptr && r0 + i < R) *reinterpret_cast<uint2 *>(emit + (r0 + i) * (long)K + kc * 4) = pk;
        };
        if (kLThreads % cpr == 0) {  // K = 256, 512, 1024: thread (i0, kc) owns rows i0, i0 + T / cpr, ... of piece kc
            const int rpp = kLThreads / cpr, i0 = tid / cpr, kc = tid - i0 * cpr;
            for (int ib = i0; ib < TM; ib += U * rpp) {
                float v[U][4], p[U][4];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = ib + u * rpp;
                    v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
                    p[u][0] = p[u][1] = p[u][2] = p[u][3] = 0.f;
                    if (i < TM && r0 + i < R) {
                        load4<float>(x + (r0 + i) * a_ls + kc * 4, v[u]);
                        if (with_pos) load4<float>(pos + pos_row(pbase, i, pos_rows) * K + kc * 4, p[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = ib + u * rpp;
                    if (i < TM) convert(i, kc, v[u], p[u]);
                }
            }
        } else {
            for (int c = tid; c < TM * cpr; c += kLThreads) {
                const int i = c / cpr, kc = c % cpr;
                float v[4] = {0.f, 0.f, 0.f, 0.f}, p[4] = {0.f, 0.f, 0.f, 0.f};
                if (r0 + i < R) {
                    load4<float>(x + (r0 + i) * a_ls + kc * 4, v);
                    if (with_pos) load4<float>(pos + pos_row(pbase, i, pos_rows) * K + kc * 4, p);
                }
                convert(i, kc, v, p);
            }
        }
    } else {
        const u16 *ab = (!below && a_alt != nullptr) ? a_alt : reinterpret_cast<const u16 *>(a);
        panel_bf16<TM, kLThreads, U>(ab, a_ls, r0, R, K, As, AS, tid);
    }
    __syncthreads();  // A panel complete
    int kt = 0;
    for (; kt + 2 < ksteps; kt += 3) {
        fetch(bq[2], kt + 2);
        multiply(bq[0], kt);
        fetch(bq[0], kt + 3);
        multiply(bq[1], kt + 1);
        fetch(bq[1], kt + 4);
        multiply(bq[2], kt + 2);
    }
    if (kt < ksteps) {
        multiply(bq[0], kt);
        if (kt + 1 < ksteps) multiply(bq[1], kt + 1);
    }
    __syncthreads();  // every wave is done with the A panel: the same LDS now takes the output tile

    // ---- + bias -> output tile in LDS [TM][kLN] (fp32, or bf16 packed two per word), then whole 16-byte pieces to global
    constexpr int OS = OUT_BF16 ? kLN + 8 : kLN + 4;  // elements per LDS row (padding keeps the scattered accumulator stores apart)
    u16 *Ob = reinterpret_cast<u16 *>(smem2);
    float *Of = reinterpret_cast<float *>(smem2);
#pragma unroll
    for (int t = 0; t < kLT; ++t) {
        const int lc = w * (kLN / kLW) + 16 * t + li;
        const float bv = __uint_as_float(bias != nullptr && bias_is_bf16 ? braw[t] << 16 : braw[t]);  // columns past N are not stored
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = acc[m][t][r] + bv;
                const int row = 16 * m + 4 * (lane >> 4) + r;
                if (OUT_BF16) Ob[row * OS + lc] = (u16)(pcm_cvt_pk_bf16(y, 0.f) & 0xFFFFu);
                else Of[row * OS + lc] = y;
            }
    }
    __syncthreads();
    constexpr int EPB = OUT_BF16 ? 8 : 4;           // elements per 16-byte piece
    constexpr int pieces = kLN / EPB;               // pieces per row
    for (int c = tid; c < TM * pieces; c += kLThreads) {
        const int i = c / pieces, pc = c % pieces, col = c0 + pc * EPB;
        if (r0 + i >= R || col >= N) continue;      // N % 8 == 0: a piece is inside or outside as a whole
        if (OUT_BF16)
            *reinterpret_cast<uint4 *>(reinterpret_cast<u16 *>(out) + (r0 + i) * out_ls + col) = *reinterpret_cast<const uint4 *>(Ob + i * OS + pc * EPB);
        else
            *reinterpret_cast<uint4 *>(reinterpret_cast<float *>(out) + (r0 + i) * out_ls + col) = *reinterpret_cast<const uint4 *>(Of + i * OS + pc * EPB);
    }
}

inline size_t linear_smem_bytes(int K, int out_bf16, int MT)
{
    const size_t rows = (size_t)kTM * MT;
    const size_t a = rows * (K + kAPad) * 2, o = out_bf16 ? rows * (kLN + 8) * 2 : rows * (kLN + 4) * 4;
    return a > o ? a : o;
}

template <bool OUT_BF16, int MT>
int launch_linear(long R, int N, int K, const void *a, int a_is_f32, long a_ls, const void *a_alt, const float *pos, long pos_n, int pos_cols,
                  const void *W, const void *bias, int bias_is_bf16, void *out, long out_ls, void *e_pos, void *e_x, hipStream_t st)
{
    const long rows = (long)kTM * MT;
    const dim3 grid((unsigned)((R + rows - 1) / rows), (unsigned)((N + kLN - 1) / kLN));
    const size_t smem = linear_smem_bytes(K, OUT_BF16, MT);
    if (smem > 64 * 1024) {
        const int rc = pcm_status(hipFuncSetAttribute(reinterpret_cast<const void *>(pcm_linear_mfma_kernel<OUT_BF16, MT>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (rc) return rc;
    }
    hipLaunchKernelGGL((pcm_linear_mfma_kernel<OUT_BF16, MT>), grid, dim3(kLThreads), smem, st, R, N, K, a, a_is_f32, a_ls, (const u16 *)a_alt,
                       pos, pos != nullptr && pos_n >= K ? pos_n / K : 1, pos_cols, (const u16 *)W, bias, bias_is_bf16, out, out_ls, (u16 *)e_pos,
                       (u16 *)e_x);
    return PCM_LAUNCH_STATUS();
}

}  // namespace

// K: multiples of 32 up to 1024; N: a multiple of 8; pos_cols: 0, N or a multiple of 256 (a workgroup's 256 columns see ONE operand)
extern "C" int pcm_linear_mfma_supported(int N, int K, int pos_cols)
{
    return N > 0 && N % 8 == 0 && K >= 32 && K <= 1024 && K % 32 == 0 && pos_cols >= 0 && (pos_cols % kLN == 0 || pos_cols >= N);
}

extern "C" int pcm_linear_mfma_forward_hip(long R, int N, int K, const void *a, int a_is_f32, long a_ls, const void *a_alt_bf16,
                                           const float *pos, long pos_n, int pos_cols, const void *w_bf16, const void *bias,
                                           int bias_is_bf16, void *out, int out_is_bf16, long out_ls, void *emit_pos_bf16,
                                           void *emit_x_bf16, void *stream)
{
    if (R < 0 || N <= 0 || K <= 0 || a_ls < K || out_ls < N) return PCM_ERR_BAD_ARG;
    if (!pcm_linear_mfma_supported(N, K, pos_cols)) return PCM_ERR_UNSUPPORTED;
    if (R == 0) return PCM_OK;
    if (!a || !w_bf16 || !out) return PCM_ERR_BAD_ARG;
    if (pos != nullptr && (!a_is_f32 || pos_n <= 0 || pos_n % K != 0)) return PCM_ERR_BAD_ARG;
    if (a_alt_bf16 != nullptr && (a_is_f32 || ((uintptr_t)a_alt_bf16 % 16) != 0)) return PCM_ERR_BAD_ARG;  // second bf16 operand only
    if ((emit_pos_bf16 != nullptr || emit_x_bf16 != nullptr) && !a_is_f32) return PCM_ERR_BAD_ARG;
    if (emit_pos_bf16 != nullptr && (pos == nullptr || pos_cols <= 0)) return PCM_ERR_BAD_ARG;
    if (emit_x_bf16 != nullptr && pos != nullptr && pos_cols >= N) return PCM_ERR_BAD_ARG;  // no column block sees plain x
    const long a_align = a_is_f32 ? 4 : 8, o_align = out_is_bf16 ? 8 : 4;  // 16-byte loads / stores
    if (a_ls % a_align || out_ls % o_align || (((uintptr_t)a | (uintptr_t)w_bf16 | (uintptr_t)out) % 16) != 0) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = R >= kLongRows && linear_smem_bytes(K, out_is_bf16, 4) <= 160 * 1024;  // K <= 1024: at most 129 KiB
#define PCM_LIN(OB, MT)                                                                                                               \
    return launch_linear<OB, MT>(R, N, K, a, a_is_f32, a_ls, a_alt_bf16, pos, pos_n, pos_cols, w_bf16, bias, bias_is_bf16, out, out_ls, \
                                 emit_pos_bf16, emit_x_bf16, st)
    if (out_is_bf16) {
        if (wide) PCM_LIN(true, 4);
        PCM_LIN(true, 1);
    }
    if (wide) PCM_LIN(false, 4);
    PCM_LIN(false, 1);
#undef PCM_LIN
}

// ================================================================================================================================
// pcm_proj_drln_mfma_backward: the BACKWARD of the chain above for the short sites -- csrc/drln.hip's backward row code and the input
// gradient of the projection,  da = dy W,  as ONE kernel (round 6; built while the GPU pool was closed: verified on the host wave64
// model only, opt-in through PCM_PROJ_MFMA_BWD, never timed).  The library served `dy @ W` as an 800 x 512 x 512 product at the launch
// floor behind pcm_drln_bwd (which had just written dy): one launch and one round trip of dy less per site.
//
//  * one workgroup = 16 rows, K / 128 waves (K = 256 / 512 / 1024: 2 / 4 / 8 waves).  Row phase: wave w finishes rows w, w + NW, ... of
//    the tile exactly like pcm_drln_bwd_kernel (same per-row arithmetic in the same order: dx and dy are bit-identical to it), stores
//    dx (fp32) and dy (bf16, the weight-gradient product still needs it) and leaves the bf16 dy row in an LDS panel [16][E + 8];
//    dgamma / dbeta / column sums of dy: per-workgroup partial rows [block][3][E], closed by pcm_reduce_batch_hip like the row kernel's.
//  * product phase: da (16 x K) = dy (16 x E) W (E x K): the reduction runs over W's ROWS, so the 16x16x32 B operand -- 8 consecutive e of
//    one output column -- is strided in memory.  No transposed mirror and no LDS staging of W: lane (li, lg) loads the 8 x 8 block
//    W[32 kt + 8 lg .. + 7][n0 + 8 li .. + 7] as eight 16-byte row pieces (a wave instruction covers 4 rows x 256 contiguous bytes)
//    and transposes it IN ITS OWN REGISTERS (two operations per operand dword): column j of the block is the B operand of the lane for
//    output column n0 + 8 li + j.  The eight column tiles of a wave are therefore the INTERLEAVED sets { n0 + 8 li + j : li = 0..15 },
//    j = 0..7 -- which makes a lane's accumulators for one row eight CONSECUTIVE output columns: da leaves the registers as one 16-byte
//    store per row, no LDS tile for the output.
//  * the first two k-steps' weight blocks are requested at the kernel's entry, under the row phase.
// Algorithmic bytes per row: 8 E read (dout, s) [+ 4 E (dout2)] + 4 E (dx) + 2 E (dy) + 2 K (da); the weight E K 2 once per 16 rows from L2.
// ================================================================================================================================
namespace {

constexpr int kBW = 128;  // output columns per wave (16 lanes x 8 columns)

template <int NCH, int NW>  // E = 256 NCH; NW = K / 128 waves
__global__ __launch_bounds__(64 * NW) void pcm_proj_drln_bwd_kernel(long R, int K, const float *__restrict__ dout, const float *__restrict__ dout2,
                                                                const float *__restrict__ s, const float *__restrict__ mean,
                                                                const float *__restrict__ rstd, const float *__restrict__ gamma, float p_drop,
                                                                const long *__restrict__ seed_ptr, unsigned site, const u16 *__restrict__ W,
                                                                float *__restrict__ dx, __hip_bfloat16 *__restrict__ dy, u16 *__restrict__ da,
                                                                long da_ls, float *__restrict__ partial)
{
    constexpr int E = NCH * 256;
    constexpr int AS = E + kAPad;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    u16 *As = reinterpret_cast<u16 *>(smem3);                                  // [16][AS] bf16: the dy panel
    float *red = reinterpret_cast<float *>(smem3 + (size_t)kTM * AS * 2);      // [NW][3][E]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long r0 = (long)blockIdx.x * kTM;
    const int n0 = w * kBW, li = lane & 15, lg = lane >> 4;

    // ---- the weight blocks of the first two k-steps: in flight under the whole row phase
    const u16 *wp = W + (long)(8 * lg) * K + n0 + 8 * li;
    constexpr int ksteps = E / 32;
    uint4 wq[2][8];
    auto fetch = [&](uint4(&dst)[8], int kt) {
        const int kk = kt < ksteps ? kt : ksteps - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = *reinterpret_cast<const uint4 *>(wp + (long)(32 * kk + j) * K);
    };
    fetch(wq[0], 0);
    fetch(wq[1], 1);

    // ---- rows (csrc/drln.hip pcm_drln_bwd_kernel, row for row)
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
    // The wave's rows w, w + NW, ... in groups of G: every load of a group (dout [, dout2], s, mean, rstd) is requested before the first row of
    // the group is worked on -- the row-at-a-time loop of pcm_drln_bwd_kernel exposes one L2 / HBM round trip per row, which that kernel
    // hides behind the other waves of a full CU and this one (one workgroup of NW waves per CU at 800 rows) cannot.  Rows are still FINISHED
    // in order w, w + NW, ...: the column sums accumulate in the same order, every bit as before.  A row past R reads row R - 1 and is
    // discarded (zero dy row in the panel, nothing stored, nothing summed).
    constexpr int RPW = kTM / NW;          // rows per wave: 8, 4, 2
    constexpr int G = RPW < 2 ? RPW : 2;   // rows in flight
#pragma unroll
    for (int j0 = 0; j0 < RPW; j0 += G) {
        float dv[G][NCH][4], d2[G][NCH][4], sv[G][NCH][4], mus[G], rss[G];
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const long r = r0 + w + NW * (j0 + jj), rc = r < R ? r : R - 1;
            mus[jj] = mean[rc], rss[jj] = rstd[rc];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const long e0 = rc * E + c * 256 + lane * 4;
                load4<float>(dout + e0, dv[jj][c]);
                if (dout2 != nullptr) load4<float>(dout2 + e0, d2[jj][c]);
                else d2[jj][c][0] = d2[jj][c][1] = d2[jj][c][2] = d2[jj][c][3] = 0.f;
                load4<float>(s + e0, sv[jj][c]);
            }
        }
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const int i = w + NW * (j0 + jj);
            const long r = r0 + i;
            if (r >= R) {  // wave-uniform: a row past the end contributes a zero dy row to the product and nothing else
#pragma unroll
                for (int c = 0; c < NCH; ++c) *reinterpret_cast<uint2 *>(As + i * AS + c * 256 + lane * 4) = make_uint2(0u, 0u);
                continue;
            }
            const float mu = mus[jj], rs = rss[jj];
            float gd[NCH][4], xh[NCH][4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float dvv = dout2 != nullptr ? dv[jj][c][v] + d2[jj][c][v] : dv[jj][c][v];
                    xh[c][v] = (sv[jj][c][v] - mu) * rs;
                    gd[c][v] = dvv * g[c][v];
                    s1 += gd[c][v];
                    s2 += gd[c][v] * xh[c][v];
                    dg[c][v] += dvv * xh[c][v];
                    db[c][v] += dvv;
                }
            }
            const float m1 = wave_sum(s1) * (1.f / E), m2 = wave_sum(s2) * (1.f / E);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const long e0 = r * E + c * 256 + lane * 4;
                float o[4], oy[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    o[v] = rs * (gd[c][v] - m1 - xh[c][v] * m2);
                    oy[v] = (keep_elem(seed, site, (uint64_t)(e0 + v), thr)) ? o[v] * scale : 0.f;
                    dys[c][v] += oy[v];
                }
                store4<float>(dx + e0, o);
                store4<__hip_bfloat16>(dy + e0, oy);
                store4<__hip_bfloat16>(reinterpret_cast<__hip_bfloat16 *>(As + i * AS + c * 256 + lane * 4), oy);  // the product's A panel
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            red[(w * 3 + 0) * E + c * 256 + lane * 4 + v] = dg[c][v];
            red[(w * 3 + 1) * E + c * 256 + lane * 4 + v] = db[c][v];
            red[(w * 3 + 2) * E + c * 256 + lane * 4 + v] = dys[c][v];
        }
    __syncthreads();  // dy panel and the waves' column sums complete
    for (int e = tid; e < 3 * E; e += 64 * NW) {
        float acc = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) acc += red[ww * 3 * E + e];
        partial[(size_t)blockIdx.x * 3 * E + e] = acc;
    }

    // ---- da tile = dy panel x W: column tile j of this wave = output columns n0 + 8 li + j (interleaved, see above)
    f4v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f4v{0.f, 0.f, 0.f, 0.f};
    auto multiply = [&](const uint4(&q)[8], int kt) {
        const bf8 af = lds_bf8(As + li * AS + 32 * kt + 8 * lg);
        const uint32_t(&u)[8][4] = reinterpret_cast<const uint32_t(&)[8][4]>(q);  // u[row e][dword]: dword d holds columns 2 d, 2 d + 1
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint4 b;
            uint32_t *bd = reinterpret_cast<uint32_t *>(&b);
#pragma unroll
            for (int d = 0; d < 4; ++d) {  // operand dword d = (W[e 2 d][col j], W[e 2 d + 1][col j])
                const uint32_t lo = u[2 * d][j >> 1], hi = u[2 * d + 1][j >> 1];
                bd[d] = (j & 1) ? ((lo >> 16) | (hi & 0xFFFF0000u)) : ((lo & 0xFFFFu) | (hi << 16));
            }
            acc[j] = PCM_MFMA_16x16x32(af, as_bf8(b), acc[j]);
        }
    };
    for (int kt = 0; kt < ksteps; kt += 2) {  // E / 32 is even
        multiply(wq[0], kt);
        fetch(wq[0], kt + 2);
        multiply(wq[1], kt + 1);
        fetch(wq[1], kt + 3);
    }
    // accumulator register r of `lane`: row 4 lg + r, column n0 + 8 li + j for j = 0..7: one 16-byte store per row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = r0 + 4 * lg + r;
        if (row < R) {
            const uint4 o = make_uint4(pcm_cvt_pk_bf16(acc[0][r], acc[1][r]), pcm_cvt_pk_bf16(acc[2][r], acc[3][r]),
                                       pcm_cvt_pk_bf16(acc[4][r], acc[5][r]), pcm_cvt_pk_bf16(acc[6][r], acc[7][r]));
            *reinterpret_cast<uint4 *>(da + row * da_ls + n0 + 8 * li) = o;
        }
    }
}

inline size_t proj_bwd_smem_bytes(int E, int K)
{
    return (size_t)kTM * (E + kAPad) * 2 + (size_t)(K / kBW) * 3 * E * sizeof(float);
}

}  // namespace

// E in {256, 512, 768, 1024} (the row code's float4 chunks); K in {256, 512, 1024} (a wave = 128 output columns, 2 / 4 / 8 waves)
extern "C" int pcm_proj_drln_mfma_backward_supported(int E, int K)
{
    return (E == 256 || E == 512 || E == 768 || E == 1024) && (K == 256 || K == 512 || K == 1024);
}

extern "C" int pcm_proj_drln_mfma_backward_blocks(long R) { return R <= 0 ? 0 : (int)((R + kTM - 1) / kTM); }

extern "C" int pcm_proj_drln_mfma_backward_hip(long R, int E, int K, const float *dout, const float *dout2, const float *s, const float *mean,
                                               const float *rstd, const float *gamma, float p_drop, const long *seed, unsigned site,
                                               const void *w_bf16, float *dx, void *dy_bf16, void *da_bf16, long da_ls, float *partial,
                                               float *dgamma_dbeta, void *dysum_bf16, void *stream)
{
    if (R < 0 || E <= 0 || K <= 0 || da_ls < K || !(p_drop >= 0.f && p_drop < 1.f)) return PCM_ERR_BAD_ARG;
    if (!pcm_proj_drln_mfma_backward_supported(E, K)) return PCM_ERR_UNSUPPORTED;
    if (R == 0) return PCM_OK;
    if (!dout || !s || !mean || !rstd || !gamma || !w_bf16 || !dx || !dy_bf16 || !da_bf16 || !partial || (p_drop > 0.f && !seed)) return PCM_ERR_BAD_ARG;
    if ((da_ls % 8) != 0 || (((uintptr_t)w_bf16 | (uintptr_t)da_bf16) % 16) != 0) return PCM_ERR_BAD_ARG;  // 16-byte loads / stores
    if (R > 0x7FFFFFFFL * kTM) return PCM_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = pcm_proj_drln_mfma_backward_blocks(R);
    const size_t smem = proj_bwd_smem_bytes(E, K);
#define PCM_PB(NCH, NWV)                                                                                                               \
    do {                                                                                                                               \
        if (smem > 64 * 1024) {                                                                                                        \
            const int rc_ = pcm_status(hipFuncSetAttribute(reinterpret_cast<const void *>(pcm_proj_drln_bwd_kernel<NCH, NWV>),         \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                    \
            if (rc_) return rc_;                                                                                                       \
        }                                                                                                                              \
        hipLaunchKernelGGL((pcm_proj_drln_bwd_kernel<NCH, NWV>), dim3((unsigned)blocks), dim3(64 * NWV), smem, st, R, K, dout, dout2, s, \
                           mean, rstd, gamma, p_drop, seed, site, (const u16 *)w_bf16, dx, (__hip_bfloat16 *)dy_bf16, (u16 *)da_bf16,  \
                           da_ls, partial);                                                                                           \
    } while (0)
#define PCM_PBK(NCH)                                                                                                                   \
    do {                                                                                                                               \
        if (K == 256) PCM_PB(NCH, 2);                                                                                                  \
        else if (K == 512) PCM_PB(NCH, 4);                                                                                             \
        else PCM_PB(NCH, 8);                                                                                                           \
    } while (0)
    switch (E / 256) {
    case 1: PCM_PBK(1); break;
    case 2: PCM_PBK(2); break;
    case 3: PCM_PBK(3); break;
    default: PCM_PBK(4); break;
    }
#undef PCM_PBK
#undef PCM_PB
    int rc = PCM_LAUNCH_STATUS();
    if (rc || dgamma_dbeta == nullptr) return rc;  // partial rows only: closed later by pcm_reduce_batch_hip (policy/deferred.py)
    const void *parts[1] = {partial};
    const int nslots[1] = {blocks}, width[1] = {3 * E}, from[1] = {2 * E};
    void *o32[1] = {dgamma_dbeta}, *o16[1] = {dysum_bf16};
    return pcm_reduce_batch_hip(1, parts, nslots, width, o32, o16, from, stream);
}

// ================================================================================================================================
// pcm_linear_mfma_backward: the INPUT gradient of the short in-projections as one kernel (round 6, opt-in PCM_LINEAR_MFMA_BWD; host-model
// evidence only, like everything in this file):
//     dx   (R, K) fp32 = dy (R, N) W (N, K) [+ dres]          dpos (R, K) fp32 = dy[:, :pos_cols] W[:pos_cols]   (nullable)
// For nn.MultiheadAttention's packed self-attention in-projection (transformer.py:244-262, 296-346: q = k = x + pos, v = x) dy is
// dq | dk | dv side by side (N = 3 E, the layout csrc/attn_small.hip's backward writes), dres the residual branch's gradient of the same x,
// pos_cols = 2 E: the position embedding sees q's and k's share only.  The framework path was ONE batched product (3 x (R, E) @ (E, E)) plus
// pcm_add4_cast2 (d3[0] + d3[1] + d3[2] + dres -> dx, d3[0] + d3[1] -> dpos): two launches and 3 R E bf16 + R E fp32 of round trip; the
// cross-attention query projection (N = E, pos_cols = N: dx and dpos are the same tensor) was a library product at the launch floor.
// Same scheme as pcm_proj_drln_bwd_kernel's product phase: 16 rows x all K columns per workgroup, K / 128 waves, the A panel (16 x N bf16)
// in LDS, W's 8 x 8 blocks transposed in registers, eight consecutive output columns per lane and row -> two 16-byte fp32 stores per
// output.  The accumulators are copied once, at k = pos_cols, for dpos.
// ================================================================================================================================
namespace {

template <int NW>  // waves per workgroup = K / 128
__global__ __launch_bounds__(64 * NW) void pcm_linear_bwd_kernel(long R, int N, int K, const u16 *__restrict__ dy, long dy_ls,
                                                                 const u16 *__restrict__ W, const float *__restrict__ dres,
                                                                 float *__restrict__ dx, float *__restrict__ dpos, int pos_cols)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
    u16 *As = reinterpret_cast<u16 *>(smem4);  // [16][N + 8] bf16: the dy panel
    const int AS = N + kAPad;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long r0 = (long)blockIdx.x * kTM;
    const int n0 = w * kBW, li = lane & 15, lg = lane >> 4;
    const u16 *wp = W + (long)(8 * lg) * K + n0 + 8 * li;
    const int ksteps = N / 32;
    uint4 wq[2][8];
    auto fetch = [&](uint4(&dst)[8], int kt) {
        const int kk = kt < ksteps ? kt : ksteps - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = *reinterpret_cast<const uint4 *>(wp + (long)(32 * kk + j) * K);
    };
    fetch(wq[0], 0);
    fetch(wq[1], 1);
    // the residual gradient of this lane's outputs: requested here, consumed after the products
    float4 rs[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = r0 + 4 * lg + r;
        rs[r][0] = rs[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dres != nullptr && row < R) {
            rs[r][0] = *reinterpret_cast<const float4 *>(dres + row * K + n0 + 8 * li);
            rs[r][1] = *reinterpret_cast<const float4 *>(dres + row * K + n0 + 8 * li + 4);
        }
    }
    panel_bf16<kTM, 64 * NW, 4>(dy, dy_ls, r0, R, N, As, AS, tid);
    __syncthreads();
    f4v acc[8], accp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = accp[j] = f4v{0.f, 0.f, 0.f, 0.f};
    auto multiply = [&](const uint4(&q)[8], int kt) {
        const bf8 af = lds_bf8(As + li * AS + 32 * kt + 8 * lg);
        const uint32_t(&u)[8][4] = reinterpret_cast<const uint32_t(&)[8][4]>(q);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint4 b;
            uint32_t *bd = reinterpret_cast<uint32_t *>(&b);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t lo = u[2 * d][j >> 1], hi = u[2 * d + 1][j >> 1];
                bd[d] = (j & 1) ? ((lo >> 16) | (hi & 0xFFFF0000u)) : ((lo & 0xFFFFu) | (hi << 16));
            }
            acc[j] = PCM_MFMA_16x16x32(af, as_bf8(b), acc[j]);
        }
    };
    const int ksnap = (dpos != nullptr && pos_cols < N) ? pos_cols / 32 : -1;  // k-step in front of which the accumulators are dpos
    for (int kt = 0; kt < ksteps; kt += 2) {
        if (kt == ksnap) {
#pragma unroll
            for (int j = 0; j < 8; ++j) accp[j] = acc[j];
        }
        multiply(wq[0], kt);
        fetch(wq[0], kt + 2);
        if (kt + 1 < ksteps) {
            if (kt + 1 == ksnap) {
#pragma unroll
                for (int j = 0; j < 8; ++j) accp[j] = acc[j];
            }
            multiply(wq[1], kt + 1);
            fetch(wq[1], kt + 3);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = r0 + 4 * lg + r;
        if (row >= R) continue;
        float *o = dx + row * K + n0 + 8 * li;
        *reinterpret_cast<float4 *>(o) = make_float4(acc[0][r] + rs[r][0].x, acc[1][r] + rs[r][0].y, acc[2][r] + rs[r][0].z, acc[3][r] + rs[r][0].w);
        *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[4][r] + rs[r][1].x, acc[5][r] + rs[r][1].y, acc[6][r] + rs[r][1].z, acc[7][r] + rs[r][1].w);
        if (dpos != nullptr) {
            float *p = dpos + row * K + n0 + 8 * li;
            if (ksnap >= 0) {
                *reinterpret_cast<float4 *>(p) = make_float4(accp[0][r], accp[1][r], accp[2][r], accp[3][r]);
                *reinterpret_cast<float4 *>(p + 4) = make_float4(accp[4][r], accp[5][r], accp[6][r], accp[7][r]);
            } else {  // pos_cols >= N: the position embedding sees the whole product (without the residual's share)
                *reinterpret_cast<float4 *>(p) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                *reinterpret_cast<float4 *>(p + 4) = make_float4(acc[4][r], acc[5][r], acc[6][r], acc[7][r]);
            }
        }
    }
}

}  // namespace

// N: a multiple of 32 up to 3072 (the dy panel: 16 x (N + 8) bf16 of LDS); K in {256, 512, 1024}; pos_cols: a multiple of 32, or >= N
extern "C" int pcm_linear_mfma_backward_supported(int N, int K, int pos_cols)
{
    return N >= 32 && N <= 3072 && N % 32 == 0 && (K == 256 || K == 512 || K == 1024) && pos_cols >= 0 && (pos_cols >= N || pos_cols % 32 == 0);
}

extern "C" int pcm_linear_mfma_backward_hip(long R, int N, int K, const void *dy_bf16, long dy_ls, const void *w_bf16, const float *dres,
                                            float *dx, float *dpos, int pos_cols, void *stream)
{
    if (R < 0 || N <= 0 || K <= 0 || dy_ls < N) return PCM_ERR_BAD_ARG;
    if (!pcm_linear_mfma_backward_supported(N, K, pos_cols)) return PCM_ERR_UNSUPPORTED;
    if (R == 0) return PCM_OK;
    if (!dy_bf16 || !w_bf16 || !dx) return PCM_ERR_BAD_ARG;
    if ((dy_ls % 8) != 0 || (((uintptr_t)dy_bf16 | (uintptr_t)w_bf16 | (uintptr_t)dx | (uintptr_t)dpos | (uintptr_t)dres) % 16) != 0) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((R + kTM - 1) / kTM);
    const size_t smem = (size_t)kTM * (N + kAPad) * 2;
#define PCM_LB(NWV)                                                                                                                     \
    do {                                                                                                                               \
        if (smem > 64 * 1024) {                                                                                                        \
            const int rc_ = pcm_status(hipFuncSetAttribute(reinterpret_cast<const void *>(pcm_linear_bwd_kernel<NWV>),                 \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                    \
            if (rc_) return rc_;                                                                                                       \
        }                                                                                                                              \
        hipLaunchKernelGGL(pcm_linear_bwd_kernel<NWV>, dim3(blocks), dim3(64 * NWV), smem, st, R, N, K, (const u16 *)dy_bf16, dy_ls,     \
                           (const u16 *)w_bf16, dres, dx, dpos, pos_cols);                                                             \
    } while (0)
    if (K == 256) PCM_LB(2);
    else if (K == 512) PCM_LB(4);
    else PCM_LB(8);
#undef PCM_LB
    return PCM_LAUNCH_STATUS();
}
