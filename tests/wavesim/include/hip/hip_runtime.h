// Stand-in for <hip/hip_runtime.h> when the product's kernel sources are compiled for the HOST wave64 model (tests/wavesim; tests
// only).  Declares exactly the device-language surface those sources use; anything else fails to compile, on purpose.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../wavesim.hpp"

#define PCM_WAVESIM 1

using wavesim::dim3;

// ---- language keywords ---------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static
#define HIP_SYMBOL(x) x

#define threadIdx (wavesim::g_cur->tid)
#define blockIdx (wavesim::g_blk->bid)
#define blockDim (wavesim::g_blk->bdim)
#define gridDim (wavesim::g_blk->gdim)
#define warpSize 64

// ---- runtime API (host pointers stand for device pointers; one stream, everything synchronous) -----------------------------------
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipDeviceptr_t;
enum { hipSuccess = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
template <class F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemsetD32Async(hipDeviceptr_t p, int v, size_t n, hipStream_t)
{
    for (size_t i = 0; i < n; ++i) ((int *)p)[i] = v;
    return hipSuccess;
}
template <class T> inline hipError_t hipMemcpyFromSymbol(void *dst, const T &sym, size_t n) { std::memcpy(dst, &sym, n); return hipSuccess; }
inline hipError_t hipRuntimeGetVersion(int *v) { *v = 0; return hipSuccess; }

template <class... KA, class... A>
inline void wavesim_launch_ggl(void (*kernel)(KA...), dim3 grid, dim3 block, size_t dyn_bytes, hipStream_t, A... args)
{
    std::function<void()> body = [=]() { kernel(args...); };
    wavesim::launch(body, grid, block, dyn_bytes);
}
// the launch census (wavesim_census_*: tools/dbg/launch_census.py) keys on the kernel expression as written at the call site
#define hipLaunchKernelGGL(kernel, ...) (wavesim::note_launch(#kernel), wavesim_launch_ggl(kernel, __VA_ARGS__))

// ---- vector types -----------------------------------------------------------------------------------------------------------------
struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) ushort4 { unsigned short x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline int2 make_int2(int x, int y) { return {x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

// A cross-lane call site is identified by the address of a static marker that every macro expansion creates for itself: lanes waiting at
// different sites are never combined into one operation (they are in different branches: different EXEC masks).
#define WS_SITE (wavesim::Site{[]() -> const void * { static const char marker = 0; return &marker; }(), \
                               (uint32_t)((__builtin_strcmp(__FILE__, __BASE_FILE__) == 0 ? 1u << 24 : 0u) | (uint32_t)__LINE__)})

// ---- synchronisation ---------------------------------------------------------------------------------------------------------------
inline void __syncthreads() { wavesim::syncthreads(); }
inline int __syncthreads_or(int p) { return wavesim::syncthreads_or(p); }
inline void __threadfence() {}
inline void __threadfence_block() {}
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_barrier() wavesim::syncthreads()
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, sync) ((void)0)  /* scheduling directive: no instruction, no value */
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)wavesim::cross(wavesim::OP_WAVE_BARRIER, WS_SITE, 0, 0))


// ---- cross-lane operations ---------------------------------------------------------------------------------------------------------
template <class T> inline __attribute__((always_inline)) T ws_shfl_from(T v, int src, wavesim::Site site)
{
    if constexpr (sizeof(T) <= 8) return wavesim::from_bits<T>(wavesim::cross(wavesim::OP_SHFL, site, wavesim::to_bits(v), 0, src));
}
#define __shfl(v, src, ...) ws_shfl_impl((v), (src), WS_SITE, ##__VA_ARGS__)
#define __shfl_xor(v, m, ...) ws_shfl_xor_impl((v), (m), WS_SITE, ##__VA_ARGS__)
#define __shfl_up(v, d, ...) ws_shfl_up_impl((v), (d), WS_SITE, ##__VA_ARGS__)
#define __shfl_down(v, d, ...) ws_shfl_down_impl((v), (d), WS_SITE, ##__VA_ARGS__)
template <class T> inline T ws_shfl_impl(T v, int src, wavesim::Site site, int width = 64)
{
    const int lane = wavesim::g_cur->lane;
    return ws_shfl_from(v, (lane & ~(width - 1)) + (src & (width - 1)), site);
}
template <class T> inline T ws_shfl_xor_impl(T v, int mask, wavesim::Site site, int width = 64)
{
    const int lane = wavesim::g_cur->lane;
    const int s = lane ^ mask;
    return ws_shfl_from(v, (s & ~(width - 1)) == (lane & ~(width - 1)) ? s : lane, site);
}
template <class T> inline T ws_shfl_up_impl(T v, unsigned d, wavesim::Site site, int width = 64)
{
    const int lane = wavesim::g_cur->lane;
    const int s = lane - (int)d;
    return ws_shfl_from(v, (s >= (lane & ~(width - 1))) ? s : lane, site);
}
template <class T> inline T ws_shfl_down_impl(T v, unsigned d, wavesim::Site site, int width = 64)
{
    const int lane = wavesim::g_cur->lane;
    const int s = lane + (int)d;
    return ws_shfl_from(v, (s <= (lane | (width - 1))) ? s : lane, site);
}
#define __ballot(p) ((unsigned long long)wavesim::cross(wavesim::OP_BALLOT, WS_SITE, (p) ? 1 : 0, 0))
#define __builtin_amdgcn_ballot_w64(p) ((unsigned long long)wavesim::cross(wavesim::OP_BALLOT, WS_SITE, (p) ? 1 : 0, 0))
#define __any(p) ((int)wavesim::cross(wavesim::OP_ANY_ALL, WS_SITE, (p) ? 1 : 0, 0, 0))
#define __all(p) ((int)wavesim::cross(wavesim::OP_ANY_ALL, WS_SITE, (p) ? 1 : 0, 0, 1))
#define __builtin_amdgcn_readlane(v, l) ((int)wavesim::cross(wavesim::OP_READLANE, WS_SITE, (uint32_t)(v), 0, (l)))
#define __builtin_amdgcn_readfirstlane(v) ((int)wavesim::cross(wavesim::OP_READFIRST, WS_SITE, (uint32_t)(v), 0))
#define __builtin_amdgcn_ds_bpermute(addr, v) ((int)wavesim::cross(wavesim::OP_BPERMUTE, WS_SITE, (uint32_t)(v), 0, (addr)))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) \
    ((int)wavesim::cross(wavesim::OP_DPP, WS_SITE, (uint32_t)(src), (uint32_t)(old), (ctrl), (rm), (bm), (bc) ? 1 : 0))
inline bool __builtin_amdgcn_inverse_ballot_w64(unsigned long long m) { return (m >> wavesim::g_cur->lane) & 1; }
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base)
{
    const int lane = wavesim::g_cur->lane;
    const unsigned below = lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u);
    return base + (unsigned)__builtin_popcount(mask & below);
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base)
{
    const int lane = wavesim::g_cur->lane;
    const unsigned below = lane <= 32 ? 0u : ((1u << (lane - 32)) - 1u);
    return base + (unsigned)__builtin_popcount(mask & below);
}

// ds_read_b64_tr_b16: inside each 16-lane group, lane p supplies the address of 4 contiguous 16-bit elements = M[p / 4][4 * (p % 4) .. + 3]
// of a 4 x 16 matrix; lane i receives the column M[0 .. 3][i] (measured on MI355X: tools/mb/tr_probe.hip)
typedef short ws_s4 __attribute__((ext_vector_type(4)));
template <class P> inline __attribute__((always_inline)) ws_s4 ws_ds_read_tr16(P p, wavesim::Site site)
{
    uint64_t mine;
    std::memcpy(&mine, (const void *)p, 8);
    return wavesim::from_bits<ws_s4>(wavesim::cross(wavesim::OP_TR16, site, mine, 0));
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) ws_ds_read_tr16((p), WS_SITE)

// matrix instructions: every lane deposits its operand registers, the resolver multiplies with the ISA's operand layouts (wavesim.cpp)
typedef float ws_f16v __attribute__((ext_vector_type(16)));
typedef float ws_f4v __attribute__((ext_vector_type(4)));
template <class A, class C> inline __attribute__((always_inline)) C ws_mfma(int shape, A a, A b, C c, wavesim::Site site)
{
    struct { char a[16], b[16]; float c[16]; } in;
    float out[16];
    std::memset(&in, 0, sizeof(in));
    std::memcpy(in.a, &a, sizeof(A)), std::memcpy(in.b, &b, sizeof(A)), std::memcpy(in.c, &c, sizeof(C));
    wavesim::cross(wavesim::OP_MFMA, site, 0, 0, shape, 0, 0, 0, &in, out);
    C r;
    std::memcpy(&r, out, sizeof(C));
    return r;
}
#define __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, x, y, z) ws_mfma(wavesim::MFMA_32x32x8_BF16_1K, (a), (b), (c), WS_SITE)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) ws_mfma(wavesim::MFMA_32x32x16_BF16, (a), (b), (c), WS_SITE)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) ws_mfma(wavesim::MFMA_16x16x32_BF16, (a), (b), (c), WS_SITE)

// ---- bit casts, integer intrinsics -------------------------------------------------------------------------------------------------
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline unsigned __brev(unsigned v) { return __builtin_bitreverse32(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFF) * (b & 0xFFFFFF); }
using std::max;
using std::min;
inline unsigned max(unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; }
inline long max(long a, int b) { return a > b ? a : (long)b; }
inline long min(long a, int b) { return a < b ? a : (long)b; }
inline long max(int a, long b) { return a > b ? (long)a : b; }
inline long min(int a, long b) { return a < b ? (long)a : b; }

// ---- math ----------------------------------------------------------------------------------------------------------------------------
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

// ---- atomics (lanes never run concurrently) -------------------------------------------------------------------------------------------
template <class T, class U> inline T atomicAdd(T *p, U v) { T o = *p; *p = o + (T)v; return o; }
template <class T, class U> inline T unsafeAtomicAdd(T *p, U v) { T o = *p; *p = o + (T)v; return o; }
template <class T, class U> inline T atomicSub(T *p, U v) { T o = *p; *p = o - (T)v; return o; }
template <class T, class U> inline T atomicMin(T *p, U v) { T o = *p; *p = o < (T)v ? o : (T)v; return o; }
template <class T, class U> inline T atomicMax(T *p, U v) { T o = *p; *p = o > (T)v ? o : (T)v; return o; }
template <class T, class U> inline T atomicOr(T *p, U v) { T o = *p; *p = o | (T)v; return o; }
template <class T, class U> inline T atomicAnd(T *p, U v) { T o = *p; *p = o & (T)v; return o; }
template <class T, class U> inline T atomicExch(T *p, U v) { T o = *p; *p = (T)v; return o; }
template <class T, class U, class V> inline T atomicCAS(T *p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
