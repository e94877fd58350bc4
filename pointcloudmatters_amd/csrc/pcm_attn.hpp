// pcm_attn.hpp -- types, MFMA operand helpers and the dropout hash shared by the attention kernels
// (csrc/attn_small.hip: short query sets; csrc/attn_flash.hip: long sequences).
#pragma once
#include "pcm_elem.hpp"

#include <math.h>

namespace {

typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

#define PCM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0)

struct AttnParams {
    const u16 *q, *k, *v;
    long q_bs, q_ls, k_bs, k_ls, v_bs, v_ls;  // element strides: batch, row
    const unsigned char *kpm;                 // (B, S) or NULL
    int B, H, L, S;
    float scale, p_drop;
    const long *seed;
    unsigned site;
};

__device__ __forceinline__ float bf2f(u16 v)
{
    return __uint_as_float((uint32_t)v << 16);
}
// two fp32 -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32.  The library form
// (__float22bfloat162_rn + reinterpretation) compiled to a conversion per element plus shifts and ors: 4 VALU operations
// per pair, a quarter of the softmax's instruction count in the attention kernels.
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi)
{
    return pcm_cvt_pk_bf16(lo, hi);  // compiler-visible (hazards between v_exp_f32 and its reader are the compiler's job)
}
__device__ __forceinline__ s4 pack4(float a, float b, float c, float d)
{
    const uint2 r = make_uint2(cvt_pk_bf16(a, b), cvt_pk_bf16(c, d));
    return __builtin_bit_cast(s4, r);
}
// max of three: the builtin chain compiles to ONE v_max3_f32 (no canonicalising v_max), and -- unlike an asm statement --
// lets the compiler insert the wait states an MFMA result needs before a VALU instruction may read it
__device__ __forceinline__ float max3f(float a, float b, float c)
{
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define PCM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
__device__ __forceinline__ bf8 as_bf8(const uint4 &v)
{
    return __builtin_bit_cast(bf8, v);
}
__device__ __forceinline__ bf8 cat8(const s4 &a, const s4 &b)  // (k 0..3 | k 4..7) of one lane's 32x32x16 operand
{
    const uint2 x = __builtin_bit_cast(uint2, a), y = __builtin_bit_cast(uint2, b);
    return as_bf8(make_uint4(x.x, x.y, y.x, y.y));
}
__device__ __forceinline__ bf8 lds_bf8(const u16 *p)
{
    return as_bf8(*reinterpret_cast<const uint4 *>(p));
}
__device__ __forceinline__ s4 lds_s4(const u16 *p)
{
    return *reinterpret_cast<const s4 *>(p);
}
__device__ __forceinline__ s4 zero_s4()
{
    s4 z = {0, 0, 0, 0};
    return z;
}
__device__ __forceinline__ int crow(int r, int lane)  // accumulator register r of `lane` -> tile row
{
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// Visibility of the 64 keys key0 .. key0 + 63 of one batch row: the byte each lane has loaded for ITS key (tile_mask_byte: one coalesced
// load per tile, requested before the tile's key / value prefetch so that waiting for it leaves those loads in flight) becomes a wave
// mask, bit t <-> key key0 + t.  The masked kernels used to read `mask[key]` inside the 32 short-circuit conditions of a tile: 32 times
// global_load_ubyte + s_waitcnt vmcnt(0) in series (each wait also drained the prefetch) -- read in the ISA in round 5.
__device__ __forceinline__ unsigned tile_mask_byte(const unsigned char *mask, int key0, int S, int lane)
{
    const int key = key0 + lane;
    return (mask != nullptr && key < S) ? (unsigned)mask[key] : 0u;
}
__device__ __forceinline__ unsigned long long tile_visible(unsigned mask_byte, int key0, int S, int lane)
{
    asm volatile("" : "+v"(mask_byte));  // the comparison stays HERE, behind the tile's matrix instructions (the compiler had moved it -- and the wait -- up to the load)
    return __ballot(key0 + lane < S && mask_byte == 0u);
}

// dropout on the attention weights.  One 32-bit hash serves the PAIR of adjacent keys (2j, 2j+1) of a query:
//   h = mixp(rowbase(b, h, q) + j * C);  keep(2j) = (h & 0xFFFF) >= thr16,  keep(2j+1) = (h >> 16) >= thr16,
// thr16 = round(p * 65536) (|p_eff - p| < 8e-6).  mixp uses a 24-bit multiply (full-rate v_mul_u32_u24; the 32-bit
// integer multiply is quarter rate) -- the mask generator used to cost as much VALU time as the softmax itself.
__device__ __forceinline__ uint32_t attn_rowbase(uint64_t seed, uint32_t site, uint32_t rowid)
{
    const uint32_t k = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ (site * 0x85EBCA6Bu);
    return mix32(k ^ rowid);
}
__device__ __forceinline__ uint32_t mixp(uint32_t x)
{
    // one xorshift-multiply-xorshift round on top of the Weyl sequence of attn_pair_bits (5 VALU operations; the mask
    // generator used to cost more VALU time than the softmax).  Keep rate, row / column means and the correlations between
    // adjacent keys, adjacent pairs and adjacent rows are at the noise level of i.i.d. bits (tests/test_small_attn_gpu.py).
    x ^= x >> 16;
    x = __umul24(x, 0xD35A2Du);  // 24-bit multiplicands: low 24 bits of x, 24-bit odd constant
    x ^= x >> 12;
    return x;
}
__device__ __forceinline__ uint32_t attn_pair_bits(uint32_t rowbase, uint32_t key_pair)
{
    return mixp(rowbase + key_pair * 0x9E3779B1u);
}

// keep masks of one hash word as WAVE masks (one bit per lane, in scalar registers): the ballots compile to two SDWA
// compares on the 16-bit halves, no extraction arithmetic.  `__builtin_amdgcn_inverse_ballot_w64(mask)` turns such a mask back
// into a per-lane condition (a v_cndmask on that scalar pair), and masks can be combined / shifted between lanes on the SCALAR
// unit for free.  (Builtins, not inline assembly: gfx950 needs two wait states between a VALU write of a scalar register and
// a VALU read of it, which only the compiler's hazard recognizer inserts.)
typedef unsigned long long lanemask;
__device__ __forceinline__ void keep_masks(uint32_t bits, uint32_t thr, lanemask &lo, lanemask &hi)
{
    lo = __builtin_amdgcn_ballot_w64((bits & 0xFFFFu) >= thr);
    hi = __builtin_amdgcn_ballot_w64((bits >> 16) >= thr);
}
__device__ __forceinline__ float keep_if(lanemask m, float x)
{
    return __builtin_amdgcn_inverse_ballot_w64(m) ? x : 0.f;
}

struct DropCfg {
    bool on;
    uint64_t seed;
    uint32_t thr;
    float inv_keep;
    __device__ DropCfg(const AttnParams &P)
    {
        on = P.p_drop > 0.f;
        seed = on ? (uint64_t)P.seed[0] : 0ull;
        thr = on ? (uint32_t)((double)P.p_drop * 65536.0 + 0.5) : 0u;  // 16-bit threshold, see attn_pair_bits
        inv_keep = on ? 1.f / (1.f - P.p_drop) : 1.f;
    }
};

}  // namespace
