// tr_probe.hip -- what does ds_read_b64_tr_b16 return?  (hipcc --offload-arch=gfx950 tools/mb/tr_probe.hip -o tools/mb/tr_probe)
// Hypothesis (guide, LDS section): inside each 16-lane group, lane p supplies the address of 4 contiguous bf16 which become
// M[p/4][4*(p%4) .. +3] of a 4 x 16 matrix; the instruction returns to lane i (in-group index) the column M[0..3][i].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t *in, uint16_t *out, const int *addr)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main()
{
    uint16_t h_in[8192], h_out[256];
    int h_addr[64];
    for (int i = 0; i < 8192; ++i) h_in[i] = (uint16_t)i;
    // rows of 72 elements (the attention tiles' stride); lane p of group g reads row (4*g' + p/4), columns 16*c + 4*(p%4)
    const int RS = 72;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, p = l & 15;
        h_addr[l] = (10 * g + p / 4) * RS + 16 * (g & 1) + 4 * (p % 4);
    }
    uint16_t *d_in, *d_out;
    int *d_addr;
    hipMalloc(&d_in, sizeof(h_in)), hipMalloc(&d_out, sizeof(h_out)), hipMalloc(&d_addr, sizeof(h_addr));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice), hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, d_addr);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        for (int j = 0; j < 4; ++j) {
            const int src_lane = 16 * g + 4 * j + i / 4;  // the lane that supplied M[j][i]
            const int want = h_addr[src_lane] + (i % 4);
            if (h_out[l * 4 + j] != want) {
                if (bad < 16) printf("lane %d elem %d: got %d want %d\n", l, j, h_out[l * 4 + j], want);
                ++bad;
            }
        }
    }
    printf("tr_probe: %s (%d mismatches)\n", bad ? "HYPOTHESIS WRONG" : "hypothesis confirmed", bad);
    if (bad) {
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
    return 0;
}
