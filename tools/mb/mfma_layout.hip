// Checks the operand / result lane maps of v_mfma_f32_32x32x8_bf16_1k on gfx950 against the assumed ones:
//   A: lane l holds A[i = l&31][k = 4*(l>>5) + 0..3],  B: lane l holds B[k = 4*(l>>5) + 0..3][j = l&31],
//   C: reg r of lane l is C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C) {  // A 32x8, B 8x32 row-major
  const int l = threadIdx.x;
  s4 a, b;
  for (int t = 0; t < 4; ++t) {
    __hip_bfloat16 x = __float2bfloat16(A[(l & 31) * 8 + 4 * (l >> 5) + t]);
    __hip_bfloat16 y = __float2bfloat16(B[(4 * (l >> 5) + t) * 32 + (l & 31)]);
    a[t] = *reinterpret_cast<short*>(&x);
    b[t] = *reinterpret_cast<short*>(&y);
  }
  f16v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
int main() {
  float hA[256], hB[256], hC[1024], ref[1024];
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 8; ++kk) hA[i * 8 + kk] = (float)((i * 3 + kk * 5) % 7 - 3);
  for (int kk = 0; kk < 8; ++kk) for (int j = 0; j < 32; ++j) hB[kk * 32 + j] = (float)((kk * 11 + j * 2) % 5 - 2) + (j == 3 ? 1.f : 0.f);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 8; ++kk) s += hA[i * 8 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) bad += hC[i] != ref[i];
  printf("mfma 32x32x8 bf16_1k layout check: %d mismatches of 1024\n", bad);
  return bad != 0;
}
