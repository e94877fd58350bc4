// EXPERIMENT (not part of libpcm_pointops.so; results in DESIGN.md section 5: no gain over the library at 800 rows, see the end of this
// comment).  Build + run: python tools/mb/experiments/mb_gemm_small.py
//
// gemm_small.hip -- bf16 projection GEMMs of SHORT activations (the ACT decoder's and CVAE encoder's 800-816 rows) on the
// matrix cores of gfx950:      C[M, N] = A[M, K] . B[N, K]^T (+ bias[N])
//
// These are the `nn.Linear` / `nn.MultiheadAttention` projections of /root/reference/src/models/components/act/transformer.py
// (:228-256 encoder layer, :317-345 decoder layer) and act.py:137-188 at the row counts of one training batch: 0.4 GFLOP each,
// ~50 of them in a row per step.  The vendor library runs them as a software-pipelined K loop (16 iterations of 32 at K = 512;
// MT32x64x32) whose every iteration waits for a round trip to L2: 7.2 us per product inside a replayed hipGraph where a
// dependent kernel node costs 1.5 us (tools/mb/mb_launch_floor.py) -- latency, not arithmetic (0.4 GFLOP = 0.2 us of the chip).
//
// Here a product is ONE round trip:
//   * a workgroup (4 waves) owns a 32 x 32 tile of C; wave w owns the quarter [w K/4, (w+1) K/4) of the reduction;
//   * both operands are K-contiguous, and v_mfma_f32_32x32x16_bf16 wants exactly 8 consecutive K values per lane for a row
//     (A: lane & 31 = row of the tile) / column (B: lane & 31 = column) -- so every lane loads its operands STRAIGHT from
//     global memory into the registers the MFMA reads, 64 contiguous bytes per lane and 64-value chunk, all loads of the tile issued
//     before the first MFMA (K = 512: 8 + 8 global_load_dwordx4 per lane in flight at once).  The order of the reduction index
//     inside an MFMA is free as long as A and B agree: lane half h of chunk c feeds k = 64 c + 32 h + 8 j .. + 7 to MFMA j;
//   * the four partial tiles meet in LDS (16 KiB, lane-major float4 slots: conflict-free), wave w finishes rows 8 w .. 8 w + 7:
//     fixed-order fp32 sum, bias, ONE rounding to the output type, 64-byte row segments stored.
// No LDS staging of operands, no K loop, no barrier but the one of the reduction.  400 workgroups at 800 x 512: every CU gets one
// or two, the whole product is launch + one memory round trip + ~0.3 us of MFMA / LDS / stores.
//
// MEASURED (MI355X, 20 products per replayed graph): 3.4 us at M = 8 and 4.1 us at M = 100 (library 3.8 / 5.0), but 5.9 us at
// 800 x 512 x 512 (library 5.9) and 19 us at 4120 rows (library 8.9): without operand reuse through LDS the 400 workgroups pull
// 25.6 MB through the L2 -> CU fabric (strided 16-byte pieces, four instructions per 128-byte line), ~10 TB/s -- the tile
// traffic, not the round trip, sets the time.  An LDS-staged 64 x 64 variant would be the library's own design.
#include "../../../pointcloudmatters_amd/csrc/pcm_attn.hpp"

namespace {

constexpr int kW = 4;  // waves per workgroup = K quarters

template <int NCH, bool OUT_F32>
__global__ __launch_bounds__(64 * kW) void pcm_gemm_nt_kernel(int M, int N, int K, const u16 *__restrict__ A, long lda,
                                                              const u16 *__restrict__ B, long ldb, const void *__restrict__ bias,
                                                              int bias_f32, void *__restrict__ C, long ldc)
{
    __shared__ __attribute__((aligned(16))) float red[kW * 4 * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, n = lane & 31;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int arow = m0 + n < M ? m0 + n : M - 1;
    const int kw = K / kW;
    const u16 *ap = A + (long)arow * lda + wave * kw + 32 * h;
    const u16 *bp = B + (long)(n0 + n) * ldb + wave * kw + 32 * h;
    uint4 av[NCH][4], bv[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            av[c][j] = *reinterpret_cast<const uint4 *>(ap + 64 * c + 8 * j);
            bv[c][j] = *reinterpret_cast<const uint4 *>(bp + 64 * c + 8 * j);
        }
    float bb = 0.f;
    if (bias != nullptr) bb = bias_f32 ? static_cast<const float *>(bias)[n0 + n] : bf2f(static_cast<const u16 *>(bias)[n0 + n]);
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = PCM_MFMA16(as_bf8(av[c][j]), as_bf8(bv[c][j]), acc);
    // accumulator register r of lane (h, n): C[m0 + (r & 3) + 8 (r >> 2) + 4 h][n0 + n]
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(red + ((wave * 4 + q) * 64 + lane) * 4) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    __syncthreads();
    float4 t = *reinterpret_cast<const float4 *>(red + ((0 * 4 + wave) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < kW; ++w) {
        const float4 u = *reinterpret_cast<const float4 *>(red + ((w * 4 + wave) * 64 + lane) * 4);
        t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
    }
    const float o[4] = {t.x + bb, t.y + bb, t.z + bb, t.w + bb};
    const int row0 = m0 + 8 * wave + 4 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (row0 + i >= M) break;
        if (OUT_F32)
            static_cast<float *>(C)[(long)(row0 + i) * ldc + n0 + n] = o[i];
        else
            static_cast<u16 *>(C)[(long)(row0 + i) * ldc + n0 + n] = (u16)(cvt_pk_bf16(o[i], 0.f) & 0xFFFFu);
    }
}

template <int NCH>
int launch_nt(int M, int N, int K, const void *A, long lda, const void *B, long ldb, const void *bias, int bias_f32, void *C, long ldc,
              int out_f32, hipStream_t st)
{
    const dim3 grid(N / 32, (M + 31) / 32);
    if (out_f32)
        hipLaunchKernelGGL((pcm_gemm_nt_kernel<NCH, true>), grid, dim3(64 * kW), 0, st, M, N, K, (const u16 *)A, lda, (const u16 *)B, ldb,
                           bias, bias_f32, C, ldc);
    else
        hipLaunchKernelGGL((pcm_gemm_nt_kernel<NCH, false>), grid, dim3(64 * kW), 0, st, M, N, K, (const u16 *)A, lda, (const u16 *)B, ldb,
                           bias, bias_f32, C, ldc);
    return PCM_LAUNCH_STATUS();
}

}  // namespace

// 1 when pcm_gemm_bf16_nt_hip takes the shape: N a multiple of 32, K in {256, 512, 1024}, 1 <= M <= 65535 * 32.
extern "C" int pcm_gemm_bf16_nt_supported(long M, long N, long K)
{
    return (M >= 1 && M <= 65535L * 32 && N >= 32 && N % 32 == 0 && N / 32 <= 2147483647L && (K == 256 || K == 512 || K == 1024)) ? 1 : 0;
}

// C (M, N; row stride ldc; bf16, or fp32 when out_f32) = A (M, K; bf16; row stride lda) . B (N, K; bf16; row stride ldb)^T + bias (N;
// bf16, or fp32 when bias_f32; may be NULL).  fp32 accumulation in a fixed order, one rounding.  Row strides in elements, multiples
// of 8 (16-byte operand loads); A, B 16-byte aligned.
extern "C" int pcm_gemm_bf16_nt_hip(long M, long N, long K, const void *A, long lda, const void *B, long ldb, const void *bias, int bias_f32,
                                    void *C, long ldc, int out_f32, void *stream)
{
    if (M == 0 || N == 0) return PCM_OK;
    if (M < 0 || N < 0 || K <= 0 || !A || !B || !C) return PCM_ERR_BAD_ARG;
    if (!pcm_gemm_bf16_nt_supported(M, N, K)) return PCM_ERR_UNSUPPORTED;
    if (lda < K || ldb < K || ldc < N || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return PCM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (K) {
    case 256: return launch_nt<1>((int)M, (int)N, (int)K, A, lda, B, ldb, bias, bias_f32, C, ldc, out_f32, st);
    case 512: return launch_nt<2>((int)M, (int)N, (int)K, A, lda, B, ldb, bias, bias_f32, C, ldc, out_f32, st);
    default: return launch_nt<4>((int)M, (int)N, (int)K, A, lda, B, ldb, bias, bias_f32, C, ldc, out_f32, st);
    }
}
