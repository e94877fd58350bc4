// Stand-in for <hip/hip_bf16.h> (tests/wavesim, tests only): __hip_bfloat16 as a 16-bit storage type with round-to-nearest-even
// conversion, the subset the kernel sources use.
#pragma once
#include <cstdint>
#include <cstring>

struct __hip_bfloat16 {
    uint16_t data;
    __hip_bfloat16() = default;
    __hip_bfloat16(float f) { data = from_float(f); }
    operator float() const { uint32_t u = (uint32_t)data << 16; float f; std::memcpy(&f, &u, 4); return f; }
    static uint16_t from_float(float f)
    {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        if ((u & 0x7FFFFFFF) > 0x7F800000) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
        u += 0x7FFF + ((u >> 16) & 1);
        return (uint16_t)(u >> 16);
    }
};
struct __hip_bfloat162 { __hip_bfloat16 x, y; };
inline __hip_bfloat16 __float2bfloat16(float f) { return __hip_bfloat16(f); }
inline float __bfloat162float(__hip_bfloat16 b) { return (float)b; }
struct float2;
template <class F2> inline __hip_bfloat162 __float22bfloat162_rn(F2 v) { return {__hip_bfloat16(v.x), __hip_bfloat16(v.y)}; }
