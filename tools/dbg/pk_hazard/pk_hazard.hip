// Minimal reproducer of the packed-fp32 hazard of DESIGN.md section 2: one packed instruction form per variant, written in inline assembly,
// evaluated ITERS times per lane against the same arithmetic done with scalar fp32 instructions; the kernel counts disagreements.
// Variant: 0  v_pk_add_f32 d, a, b                                        (no operand select)
//          1  v_pk_add_f32 d, a, b op_sel_hi:[1,0]                        (high lane takes b.lo: broadcast of b.lo)
//          2  v_pk_add_f32 d, a, b op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]   (a - broadcast(b.lo): what the compiler emits for vector - scalar)
//          3  v_pk_add_f32 d, a, b op_sel:[0,1]  neg_lo:[0,1] neg_hi:[0,1]      (a - broadcast(b.hi))
//          4  v_pk_mul_f32 d, a, b op_sel_hi:[1,0]
//          5  v_pk_fma_f32 d, a, b, c op_sel_hi:[1,0,1]
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(128) void pk_kernel(int iters, const float *__restrict__ in, unsigned *__restrict__ bad)
{
    const int t = blockIdx.x * 128 + threadIdx.x;
    f2 a = {in[2 * t], in[2 * t + 1]};
    f2 b = {in[2 * t + 1] * 0.5f + 0.25f, in[2 * t] * 0.75f - 0.125f};
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f2 d, want;
        const f2 c = {0.5f, 0.25f};
        if (V == 0) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x + b.x, a.y + b.y};
        } else if (V == 1) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x + b.x, a.y + b.x};
        } else if (V == 2) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x - b.x, a.y - b.x};
        } else if (V == 3) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x - b.y, a.y - b.y};
        } else if (V == 4) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x * b.x, a.y * b.x};
        } else if (V == 5) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            want = (f2){__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.x, c.y)};
        } else if (V == 6) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x + b.y, a.y + b.y};
        } else if (V == 7) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.y + b.x, a.y + b.y};
        } else if (V == 8) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x * b.y, a.y * b.y};
        } else if (V == 9) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x + b.y, a.y + b.x};
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            want = (f2){a.x - b.x, a.y - b.y};
        }
        nbad += (__float_as_uint(d.x) != __float_as_uint(want.x)) || (__float_as_uint(d.y) != __float_as_uint(want.y));
        // new operands every iteration (scalar arithmetic only), bounded
        a = (f2){want.x * 0.5f + 0.125f, want.y * 0.25f - 0.0625f};
        b = (f2){b.y * 0.5f + 0.3f, b.x * 0.5f - 0.2f};
    }
    bad[t] = nbad;
}

extern "C" int pk_launch(int variant, int blocks, int iters, const float *in, unsigned *bad, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
    case 0: hipLaunchKernelGGL(pk_kernel<0>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 1: hipLaunchKernelGGL(pk_kernel<1>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 2: hipLaunchKernelGGL(pk_kernel<2>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 3: hipLaunchKernelGGL(pk_kernel<3>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 4: hipLaunchKernelGGL(pk_kernel<4>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 5: hipLaunchKernelGGL(pk_kernel<5>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 6: hipLaunchKernelGGL(pk_kernel<6>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 7: hipLaunchKernelGGL(pk_kernel<7>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 8: hipLaunchKernelGGL(pk_kernel<8>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    case 9: hipLaunchKernelGGL(pk_kernel<9>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    default: hipLaunchKernelGGL(pk_kernel<10>, dim3(blocks), dim3(128), 0, st, iters, in, bad); break;
    }
    return (int)hipGetLastError();
}
