// Self-test of the host wave64 model (tests/test_wavesim_parity.py::test_model_cross_lane_and_matrix_primitives): every cross-lane
// primitive the kernel sources use, against a direct formula.  Compiled with the same stand-in headers as the kernel sources.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

static int g_bad = 0;
#define CHECK(cond, what)                                                        \
    do {                                                                         \
        if (!(cond)) { if (g_bad++ < 10) std::printf("FAILED: %s\n", what); }    \
    } while (0)

__global__ void cross_lane(int *out)  // 128 threads = 2 waves; out[tid * 16 + k]
{
    const int tid = threadIdx.x, lane = tid & 63;
    int *o = out + tid * 16;
    const int v = 1000 * (tid >> 6) + lane;
    o[0] = __shfl_xor(v, 5);
    o[1] = __shfl(v, 7, 16);       // lane 7 of each 16-lane segment
    o[2] = __shfl_up(v, 3);
    o[3] = __shfl_down(v, 2, 32);
    o[4] = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    o[5] = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    o[6] = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    o[7] = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false);  // row_mirror
    o[8] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xF, 0xF, false); // row_shr:1, old = -1 where the source is outside the row
    o[9] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, true);  // wave_shr:1 with bound_ctrl: 0 in lane 0
    o[10] = __builtin_amdgcn_readlane(v, 42);
    o[11] = __builtin_amdgcn_ds_bpermute(4 * ((lane * 3) & 63), v);
    o[12] = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // EXEC-mask semantics: only the odd lanes of the first wave take part in this ballot
    unsigned long long m = 0;
    if ((lane & 1) && tid < 64) m = __ballot(lane < 10);
    o[13] = (int)m;
    o[14] = __builtin_amdgcn_readfirstlane(v);
    if (tid >= 64 + 32) return;  // half of the second wave ends before the barrier: s_barrier counts live waves / lanes
    __shared__ int sh[128];
    sh[tid] = v;
    __syncthreads();
    o[15] = sh[(tid + 1) % 96];
}

__global__ void mfma(const float *A, const float *B, float *D, int shape)
{
    const int l = threadIdx.x;
    if (shape == 0) {  // 32x32x8: A 32 x 8, B 8 x 32
        s4 a, b;
        for (int t = 0; t < 4; ++t) {
            a[t] = (short)(__float_as_uint(A[(l % 32) * 8 + 4 * (l / 32) + t]) >> 16);
            b[t] = (short)(__float_as_uint(B[(4 * (l / 32) + t) * 32 + l % 32]) >> 16);
        }
        f16v c;
        for (int r = 0; r < 16; ++r) c[r] = 1.f;
        c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
    } else if (shape == 1) {  // 32x32x16: A 32 x 16, B 16 x 32
        bf8 a, b;
        for (int t = 0; t < 8; ++t) a[t] = (__bf16)A[(l % 32) * 16 + 8 * (l / 32) + t], b[t] = (__bf16)B[(8 * (l / 32) + t) * 32 + l % 32];
        f16v c;
        for (int r = 0; r < 16; ++r) c[r] = 1.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
    } else {  // 16x16x32: A 16 x 32, B 32 x 16
        bf8 a, b;
        for (int t = 0; t < 8; ++t) a[t] = (__bf16)A[(l % 16) * 32 + 8 * (l / 16) + t], b[t] = (__bf16)B[(8 * (l / 16) + t) * 16 + l % 16];
        f4v c = {1.f, 1.f, 1.f, 1.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = c[r];
    }
}

__global__ void tr16(const unsigned short *in, unsigned short *out, const int *addr)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds + addr[threadIdx.x]);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

int main()
{
    static int out[128 * 16];
    hipLaunchKernelGGL(cross_lane, dim3(1), dim3(128), 0, 0, out);
    for (int tid = 0; tid < 128; ++tid) {
        const int lane = tid & 63, base = 1000 * (tid >> 6);
        const int *o = out + tid * 16;
        CHECK(o[0] == base + (lane ^ 5), "shfl_xor");
        CHECK(o[1] == base + (lane & ~15) + 7, "shfl width 16");
        CHECK(o[2] == base + (lane >= 3 ? lane - 3 : lane), "shfl_up");
        CHECK(o[3] == base + (((lane & 31) + 2 < 32) ? lane + 2 : lane), "shfl_down width 32");
        CHECK(o[4] == base + (lane ^ 1), "dpp quad_perm [1,0,3,2]");
        CHECK(o[5] == base + (lane ^ 2), "dpp quad_perm [2,3,0,1]");
        CHECK(o[6] == base + ((lane & ~7) | (7 - (lane & 7))), "dpp row_half_mirror");
        CHECK(o[7] == base + ((lane & ~15) | (15 - (lane & 15))), "dpp row_mirror");
        CHECK(o[8] == ((lane & 15) ? base + lane - 1 : -1), "dpp row_shr:1");
        CHECK(o[9] == (lane ? base + lane - 1 : 0), "dpp wave_shr:1 bound_ctrl");
        CHECK(o[10] == base + 42, "readlane");
        CHECK(o[11] == base + ((lane * 3) & 63), "ds_bpermute");
        CHECK(o[12] == lane, "mbcnt");
        CHECK(o[13] == ((tid < 64 && (lane & 1)) ? 0x2AA : 0), "ballot under a divergent branch");  // odd lanes below 10: 1,3,5,7,9
        CHECK(o[14] == base, "readfirstlane");
        if (tid < 96) CHECK(o[15] == 1000 * (((tid + 1) % 96) >> 6) + (((tid + 1) % 96) & 63), "barrier with ended lanes");
    }
    static float A[32 * 32], B[32 * 32], D[32 * 32];
    for (int i = 0; i < 1024; ++i) A[i] = (float)((i * 7) % 5 - 2), B[i] = (float)((i * 3) % 7 - 3);
    const int M[3] = {32, 32, 16}, K[3] = {8, 16, 32};
    for (int shape = 0; shape < 3; ++shape) {
        hipLaunchKernelGGL(mfma, dim3(1), dim3(64), 0, 0, (const float *)A, (const float *)B, D, shape);
        for (int i = 0; i < M[shape]; ++i)
            for (int j = 0; j < M[shape]; ++j) {
                float s = 1.f;
                for (int k = 0; k < K[shape]; ++k) s += A[i * K[shape] + k] * B[k * M[shape] + j];
                CHECK(s == D[i * M[shape] + j], "mfma");
            }
    }
    static unsigned short in[4096], tro[256];
    static int addr[64];
    for (int i = 0; i < 4096; ++i) in[i] = (unsigned short)i;
    for (int l = 0; l < 64; ++l) addr[l] = (10 * (l >> 4) + (l & 15) / 4) * 72 + 16 * ((l >> 4) & 1) + 4 * ((l & 15) % 4);
    hipLaunchKernelGGL(tr16, dim3(1), dim3(64), 0, 0, (const unsigned short *)in, tro, (const int *)addr);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) CHECK(tro[l * 4 + j] == addr[16 * (l >> 4) + 4 * j + (l & 15) / 4] + (l & 15) % 4, "ds_read_tr16_b64");
    if (g_bad) { std::printf("%d checks failed\n", g_bad); return 1; }
    std::printf("all ok\n");
    return 0;
}
