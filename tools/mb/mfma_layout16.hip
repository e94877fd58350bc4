// Checks the operand lane maps of v_mfma_f32_32x32x16_bf16 on gfx950:
//   A: lane l holds A[i = l&31][k = 8*(l>>5) + 0..7],  B: lane l holds B[k = 8*(l>>5) + 0..7][j = l&31],  C as for 32x32x8.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C) {  // A 32x16, B 16x32 row-major
  const int l = threadIdx.x;
  bf8 a, b;
  for (int t = 0; t < 8; ++t) {
    a[t] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + t];
    b[t] = (__bf16)B[(8 * (l >> 5) + t) * 32 + (l & 31)];
  }
  f16v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
int main() {
  float hA[512], hB[512], hC[1024], ref[1024];
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) hA[i * 16 + kk] = (float)((i * 3 + kk * 5) % 7 - 3);
  for (int kk = 0; kk < 16; ++kk) for (int j = 0; j < 32; ++j) hB[kk * 32 + j] = (float)((kk * 11 + j * 2) % 5 - 2) + (j == 3 ? 1.f : 0.f);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dC;
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dC, sizeof hC);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) bad += hC[i] != ref[i];
  printf("mfma 32x32x16 bf16 layout check: %d mismatches of 1024\n", bad);
  return bad != 0;
}
