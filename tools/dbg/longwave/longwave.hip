// Does a long-running, register-resident workgroup compute the same thing when heavy kernels run beside it on another stream?
// mode 0: a dependent fp32 chain in VGPRs; mode 1: + a wave reduction (DPP / readlane) per iteration; mode 2: + an LDS exchange with a
// workgroup barrier per iteration (the structure of csrc/fps.hip's pick loop).
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(128) void longwave_kernel(int iters, int mode, const float *__restrict__ in, float *__restrict__ out)
{
    __shared__ float slot[2][4];
    const int u = threadIdx.x, lane = u & 63, wave = u >> 6;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[(blockIdx.x * 128 + u) * 8 + i];
    float o = in[0];
    for (int it = 0; it < iters; ++it) {
        float best = -1.f;
        for (int i = 0; i < 8; ++i) {
            const float d = (v[i] - o) * (v[i] - o);
            v[i] = v[i] * 0.999999f + d * 1e-3f;
            best = d > best ? d : best;
        }
        if (mode >= 1) {
            for (int s = 32; s >= 1; s >>= 1) best = fmaxf(best, __shfl_xor(best, s));
        }
        if (mode >= 2) {
            if (lane == 0) slot[it & 1][wave] = best;
            __syncthreads();
            best = fmaxf(slot[it & 1][0], slot[it & 1][1]);
        }
        o = best * 0.5f;
    }
    float acc = o;
    for (int i = 0; i < 8; ++i) acc += v[i];
    out[blockIdx.x * 128 + u] = acc;
}
extern "C" int longwave_launch(int blocks, int iters, int mode, const float *in, float *out, void *stream)
{
    hipLaunchKernelGGL(longwave_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, iters, mode, in, out);
    return (int)hipGetLastError();
}
