// pcm_elem.hpp -- small device helpers shared by the fused elementwise / row kernels (drln.hip, ffn.hip):
// counter-based dropout mask, 16-byte typed loads / stores, wave sum.
#pragma once
#include "pcm_common.hpp"

#include <hip/hip_bf16.h>

namespace {

// 32-bit avalanche ("lowbias32"): two multiplies, good enough for Bernoulli masks and cheap in VGPRs
__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}

// keep decision for element `e` of call site `site` under `seed`; threshold = p * 2^32.
// The counter (seed, site, e) is folded to 32 bits first, then avalanched twice.
__device__ __forceinline__ bool keep_elem(uint64_t seed, uint32_t site, uint64_t e, uint32_t threshold)
{
    const uint32_t k = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ (site * 0x85EBCA6Bu) ^ ((uint32_t)(e >> 32) * 0xC2B2AE35u);
    return mix32(mix32((uint32_t)e ^ k) + k) >= threshold;
}

__device__ __forceinline__ float wave_sum(float v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <typename T>
__device__ __forceinline__ void load4(const T *p, float (&o)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float *p, float (&o)[4])
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
}
template <>
__device__ __forceinline__ void load4<__hip_bfloat16>(const __hip_bfloat16 *p, float (&o)[4])
{
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    o[0] = __uint_as_float(v.x << 16), o[1] = __uint_as_float(v.x & 0xFFFF0000u);
    o[2] = __uint_as_float(v.y << 16), o[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
template <typename T>
__device__ __forceinline__ void store4(T *p, const float (&o)[4]);
template <>
__device__ __forceinline__ void store4<float>(float *p, const float (&o)[4])
{
    *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
template <>
__device__ __forceinline__ void store4<__hip_bfloat16>(__hip_bfloat16 *p, const float (&o)[4])
{
    *reinterpret_cast<uint2 *>(p) = make_uint2(pcm_cvt_pk_bf16(o[0], o[1]), pcm_cvt_pk_bf16(o[2], o[3]));
}


}  // namespace
