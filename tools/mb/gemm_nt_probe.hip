// gemm_nt_probe.hip -- is a no-LDS MFMA kernel faster than hipBLASLt for the step's tiny NT GEMMs (C = A . W^T + b)?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/mb/libgemm_nt_probe.so tools/mb/gemm_nt_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bf8 ld8(const uint16_t *p) { return __builtin_bit_cast(bf8, *reinterpret_cast<const uint4 *>(p)); }

// one wave = one 32 x 32 tile of C; workgroup = TM x TN waves.  MFMA-A = A rows (m), MFMA-B = W rows (n): result lane <-> n,
// registers <-> m rows {8j + 4(lane>>5) + i}.
template <int WM, int WN, int UN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt(int M, int N, int K, const uint16_t *__restrict__ A, long lda,
                                                        const uint16_t *__restrict__ W, long ldw, const uint16_t *__restrict__ bias,
                                                        uint16_t *__restrict__ C, long ldc)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int r0 = (blockIdx.y * WM + wm) * 32, c0 = (blockIdx.x * WN + wn) * 32;
    if (r0 >= M || c0 >= N) return;
    const int i = lane & 31, kh = lane >> 5;
    int ar = r0 + i;
    if (ar >= M) ar = M - 1;
    const uint16_t *ap = A + (long)ar * lda + kh * 8;
    const uint16_t *wp = W + (long)(c0 + i) * ldw + kh * 8;
    f16v acc = {0};
    for (int k0 = 0; k0 < K; k0 += 16 * UN) {
        bf8 a[UN], b[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            a[u] = ld8(ap + k0 + 16 * u);
            b[u] = ld8(wp + k0 + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc, 0, 0, 0);
    }
    const float bv = bias ? __uint_as_float((uint32_t)bias[c0 + i] << 16) : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = r0 + 8 * j + 4 * kh + q;
            if (m < M) {
                const __hip_bfloat16 o = __float2bfloat16(acc[4 * j + q] + bv);
                C[(long)m * ldc + c0 + i] = *reinterpret_cast<const uint16_t *>(&o);
            }
        }
}

extern "C" int gemm_nt_launch(int variant, int M, int N, int K, const void *A, long lda, const void *W, long ldw, const void *bias,
                              void *C, long ldc, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
#define L(WM, WN, UN)                                                                                                          \
    hipLaunchKernelGGL((gemm_nt<WM, WN, UN>), dim3((N + 32 * WN - 1) / (32 * WN), (M + 32 * WM - 1) / (32 * WM)), dim3(64 * WM * WN), 0, \
                       st, M, N, K, (const uint16_t *)A, lda, (const uint16_t *)W, ldw, (const uint16_t *)bias, (uint16_t *)C, ldc)
    switch (variant) {
    case 0: L(2, 2, 8); break;
    case 1: L(1, 1, 8); break;
    case 2: L(1, 2, 8); break;
    case 3: L(2, 2, 16); break;
    case 4: L(1, 4, 8); break;
    case 5: L(4, 1, 8); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
