// Where do the ~1200 clocks of an FPS pick go?  Strip the loop down piece by piece (results are NOT valid FPS output).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t wave_max_fast(uint32_t v)
{
    asm volatile(
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// MODE bits: 1 distance update, 2 wave reduce, 4 LDS publish + barrier + read, 8 idx store
template <int MODE>
__global__ __launch_bounds__(256) void probe(const float *xyz, int N, int M, int *idx, long long *cyc)
{
    constexpr int PPT = 4, W = 4;
    __shared__ uint32_t keys[2][4];
    __shared__ float4 cand[2][4];
    const int u = threadIdx.x, lane = u & 63, wave = u >> 6;
    const float *cloud = xyz + (size_t)blockIdx.x * N * 3;
    f2 px[2], py[2], pz[2];
    float md[4];
    for (int s = 0; s < PPT; ++s) {
        int j = u + s * 256;
        px[s / 2][s % 2] = cloud[j * 3], py[s / 2][s % 2] = cloud[j * 3 + 1], pz[s / 2][s % 2] = cloud[j * 3 + 2];
        md[s] = 1e10f;
    }
    float ox = cloud[0], oy = cloud[1], oz = cloud[2];
    if (u < 8) ((uint32_t *)keys)[u] = 0;
    __syncthreads();
    long long t0 = wall_clock64();
    long long c0 = clock64();
    uint32_t acc = 0;
    for (int it = 1; it < M; ++it) {
        float best = -1.f; int bs = 0;
        if (MODE & 1) {
            f2 o2x = (f2)(ox), o2y = (f2)(oy), o2z = (f2)(oz);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f2 dx = px[k] - o2x, dy = py[k] - o2y, dz = pz[k] - o2z;
                f2 d = dx * dx; d = d + dy * dy; d = d + dz * dz;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int s = 2 * k + h;
                    float d2 = __builtin_fminf(d[h], md[s]); md[s] = d2;
                    bool g = d2 > best; best = g ? d2 : best; bs = g ? s : bs;
                }
            }
        } else { best = md[0] + (float)it; }
        uint32_t key = __float_as_uint(best) + 1u;
        uint32_t j = u + bs * 256;
        uint32_t m1 = key;
        unsigned long long own = 1;
        if (MODE & 2) { m1 = wave_max_fast(key); own = __ballot(key == m1); }
        int buf = it & 1;
        float4 o = make_float4(__uint_as_float(j), ox + 1e-3f, oy, oz);
        if (MODE & 4) {
            if (lane == (int)__builtin_ctzll(own)) {
                float bx = 0, by = 0, bz = 0;
#pragma unroll
                for (int s = 0; s < PPT; ++s) if (s == bs) bx = px[s / 2][s % 2], by = py[s / 2][s % 2], bz = pz[s / 2][s % 2];
                keys[buf][wave] = m1; cand[buf][wave] = make_float4(__uint_as_float(j), bx, by, bz);
            }
            __syncthreads();
            uint4 q = *reinterpret_cast<const uint4 *>(&keys[buf][0]);
            uint32_t kw[4] = {q.x, q.y, q.z, q.w};
            uint32_t gmax = max(max(kw[0], kw[1]), max(kw[2], kw[3]));
            int wsel = 0;
            for (int w = W - 1; w >= 0; --w) wsel = kw[w] == gmax ? w : wsel;
            o = cand[buf][wsel];
        } else { acc += m1; }
        ox = o.y, oy = o.z, oz = o.w;
        if ((MODE & 8) && u == 0) idx[blockIdx.x * M + it] = (int)__float_as_uint(o.x);
    }
    long long c1 = clock64(), t1 = wall_clock64();
    if (u == 0) { cyc[blockIdx.x * 2] = c1 - c0; cyc[blockIdx.x * 2 + 1] = t1 - t0; idx[blockIdx.x * M] = (int)acc + (int)md[0]; }
}
template <int MODE> void run(const float *x, int N, int M, int *idx, long long *cyc, const char *what)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(8), dim3(256), 0, 0, x, N, M, idx, cyc);
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(8), dim3(256), 0, 0, x, N, M, idx, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
    printf("%-44s %7.1f ns/pick   clock64 ticks/pick %.1f   wall ticks/pick %.1f (wall clock rate %d kHz)\n", what, ms / 10 * 1e6 / (M - 1),
           (double)h[0] / (M - 1), (double)h[1] / (M - 1), wcr);
}
int main()
{
    const int N = 1024, M = 512, B = 8;
    std::vector<float> h(B * N * 3);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 10007) / 10007.f;
    float *x; int *idx; long long *cyc;
    hipMalloc(&x, h.size() * 4); hipMalloc(&idx, B * M * 4); hipMalloc(&cyc, 256);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("device clock rate attribute: %d kHz\n", clk);
    run<15>(x, N, M, idx, cyc, "full (distance + wave + LDS/barrier + store)");
    run<7>(x, N, M, idx, cyc, "no idx store");
    run<3>(x, N, M, idx, cyc, "distance + wave reduce (no LDS / barrier)");
    run<1>(x, N, M, idx, cyc, "distance only");
    run<4>(x, N, M, idx, cyc, "LDS publish + barrier + read only");
    run<6>(x, N, M, idx, cyc, "wave reduce + LDS/barrier");
    run<2>(x, N, M, idx, cyc, "wave reduce only");
    return 0;
}
