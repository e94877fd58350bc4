// pcm_common.hpp -- shared device helpers for the gfx950 pointops kernels.
// CDNA4 only: wave = 64 lanes, DPP row ops, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/pcm_pointops.h"

#define PCM_WAVE 64

static inline int pcm_status(hipError_t e) { return e == hipSuccess ? PCM_OK : PCM_ERR_HIP_BASE + (int)e; }
#define PCM_LAUNCH_STATUS() pcm_status(hipGetLastError())

// A/B switches of the microbenchmarks (tools/mb): the environment is read only in a build made with `make MB=1`
// (-DPCM_MB_SWITCHES); the shipped library never reads the environment and always takes the measured default.
static inline int pcm_mb_switch(const char *name, int dflt)
{
#ifdef PCM_MB_SWITCHES
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// ---- cross-lane primitives ------------------------------------------------------------------
// DPP controls (cdna4 ISA): quad_perm[a,b,c,d] = a|b<<2|c<<4|d<<6, row_half_mirror 0x141,
// row_mirror 0x140, wave_shr:1 0x138, wave_shl:1 0x130.
template <int CTRL>
__device__ __forceinline__ uint32_t pcm_dpp(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

// After the four steps every lane of a 16-lane row holds the row's reduction; the four row
// results are combined on the scalar unit.  Result is wave-uniform.
__device__ __forceinline__ uint32_t pcm_wave_max_u32(uint32_t v)
{
    v = max(v, pcm_dpp<0xB1>(v));   // lane ^ 1
    v = max(v, pcm_dpp<0x4E>(v));   // lane ^ 2
    v = max(v, pcm_dpp<0x141>(v));  // mirror within 8
    v = max(v, pcm_dpp<0x140>(v));  // mirror within 16
    const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ uint32_t pcm_wave_min_u32(uint32_t v)
{
    v = min(v, pcm_dpp<0xB1>(v));
    v = min(v, pcm_dpp<0x4E>(v));
    v = min(v, pcm_dpp<0x141>(v));
    v = min(v, pcm_dpp<0x140>(v));
    const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

// two fp32 -> one dword holding two bf16 (round to nearest even), ONE v_cvt_pk_bf16_f32 (gfx950).  The library route
// (__float2bfloat16 per element + reinterpretation) costs a conversion per element plus a shift and an or per pair.
// Written as a vector conversion, NOT as inline assembly: the hazard recognizer does not look inside an asm statement, and
// gfx950 needs a wait state between a transcendental (v_exp_f32, v_rcp_f32 ...) and a VALU instruction that reads its result
// -- an asm v_cvt_pk placed right behind a v_exp read a stale register (found when a loop was restructured).
typedef float pcm_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 pcm_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pcm_cvt_pk_bf16(float lo, float hi)
{
    const pcm_f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pcm_bf2));
}

// ---- XCD-aware work split -----------------------------------------------------------------------------------------------------
// MI355X dispatches workgroup b to XCD b % 8, and every XCD has its own 4 MiB L2.  A gather kernel whose consecutive workgroups
// take consecutive rows therefore spreads every cloud over all eight L2s, and each source row / record line is fetched from
// HBM (or the Infinity Cache) up to eight times (PMC: 1.7-3.6x the algorithmic bytes, profiles/r03_bench_tables.json).  With
// this split XCD x owns the x-th contiguous eighth of the rows -- whole clouds, in the packed layout -- so a source line is
// fetched by ONE L2.  Launchers round the grid to a multiple of 8 (pcm_xcd_grid); results do not change (same per-row code).
struct PcmXcdSplit {
    long lo, hi;     // this workgroup's row range [lo, hi)
    long first, step;  // first row of this workgroup's first item and the row stride between its passes, for `per_block` rows per pass
};
__device__ __forceinline__ PcmXcdSplit pcm_xcd_split(long rows, long per_block)
{
    PcmXcdSplit s;
    if ((gridDim.x & 7) != 0 || gridDim.x < 8) {  // not split: one range, blocks strided over it
        s.lo = 0, s.hi = rows, s.first = (long)blockIdx.x * per_block, s.step = (long)gridDim.x * per_block;
        return s;
    }
    const long xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, per = gridDim.x >> 3;
    const long chunk = (rows + 7) >> 3;
    s.lo = xcd * chunk;
    s.hi = s.lo + chunk < rows ? s.lo + chunk : rows;
    s.first = s.lo + lb * per_block;
    s.step = per * per_block;
    return s;
}
static inline int pcm_xcd_grid(long blocks) { return (int)(blocks >= 8 ? (blocks + 7) / 8 * 8 : (blocks < 1 ? 1 : blocks)); }

// Kernel heads.  The compiler sinks every kernel-argument load (s_load from the kernarg segment) to the block that first uses the
// argument; a kernel that tests `p_drop > 0` before it reads `seed_ptr`, the seed, R and then the rest starts with a CHAIN of four or
// five dependent scalar round trips before its first vector load is even issued (read in the ISA of csrc/drln.hip; the phase clocks of
// csrc/ffn_mfma.hip measured "kernel arguments -> seed -> operand loads" at ~5 k clocks in round 4).  An empty asm statement that names
// the arguments as scalar inputs at the top of the kernel --   asm volatile("" ::"s"(R), "s"(x), ...);   -- forces all of them into
// registers there: the loads leave as one batch, one wait.  The dropout seed (a load THROUGH one of those arguments) is taken off the
// chain the same way from the other side: it is first touched behind the row's vector loads (`asm volatile("" : "+v"(seed_r))` keeps
// the hash constants from being hoisted in front of them again).  tests/wavesim strips both (build.py rewrite 3).

// Job lookup of the table-driven batch kernels (tokens.hip, optim.hip): the index of the job whose first workgroup `first(j)` is the
// last one <= blk, for n <= 128 jobs with ascending first(j), first(0) == 0.  `first` reads the table in the kernel-argument segment
// with a LANE-dependent index: one vector load per 64 jobs, one ballot -- the scalar loop `for (j = 1; j < n; ++j) if (blk >= first(j))
// i = j;` it replaces was n - 1 dependent scalar loads (s_load, s_waitcnt lgkmcnt(0), compare: up to 31 round trips in series at the head
// of a 6-25 us kernel; the bisection of pcm_xfer_batch_kernel: 7).
template <class F>
__device__ __forceinline__ int pcm_job_of(int blk, int n, F first)
{
    const int lane = (int)(threadIdx.x & 63);
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const bool le = j >= 1 && j < n && first(j) <= blk;
        cnt += __popcll(__ballot(le));
    }
    return cnt;
}

// The two offset tables of a batch of at most 64 clouds, one entry per LANE (two vector loads, in flight together), and the lookups the
// index kernels need from them in registers: cloud of a query (= number of clouds whose last query index + 1 is <= q: the same answer as
// pcm_cloud_of) by a ballot, table entries by v_readlane.  The kernels used to bisect `new_offset` with dependent scalar loads (2 x log2 b)
// and then read four more entries one after another: ~12 scalar round trips in series before the first point was requested.
struct PcmCloudTable {
    int off, noff;  // offset[lane], new_offset[lane] (INT_MAX past b)
    bool ok;        // b <= 64: the table is valid (otherwise the callers keep pcm_cloud_of and plain loads)
    __device__ __forceinline__ PcmCloudTable(const int *__restrict__ offset, const int *__restrict__ new_offset, int b, int lane)
    {
        ok = b > 0 && b <= 64;
        off = (ok && lane < b) ? offset[lane] : 0x7FFFFFFF;
        noff = (ok && lane < b) ? new_offset[lane] : 0x7FFFFFFF;
    }
    __device__ __forceinline__ int cloud_of(int q) const { return __popcll(__ballot(noff <= q)); }
    __device__ __forceinline__ int offset_at(int c) const { return __builtin_amdgcn_readlane(off, c); }
    __device__ __forceinline__ int new_offset_at(int c) const { return __builtin_amdgcn_readlane(noff, c); }
};

// Closing reduction of partial rows: sum over slots s0, s0 + step, ... (< nslots) of partial[s * VH + e] in fp64, IN THAT ORDER, with
// eight loads in flight.  The plain loop `acc += partial[s * VH + e]` compiles to load - s_waitcnt vmcnt(0) - add per slot: one exposed
// L2 round trip per slot and wave (tools/isa_load_chains.py), 30-130 in series for the 256-1030 partial rows of an ACT step's reductions.
// The values are added in the same order as before: the sums are bit-identical.
__device__ __forceinline__ double pcm_slot_sum(const float *__restrict__ partial, size_t VH, int e, int s0, int step, int nslots)
{
    double acc = 0.0;
    int s = s0;
    for (; s + 7 * step < nslots; s += 8 * step) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(s + u * step) * VH + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; s < nslots; s += step) acc += (double)partial[(size_t)s * VH + e];
    return acc;
}

__device__ __forceinline__ int pcm_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Squared distance in the reference's order: (a-b)*(a-b) for x, y, z summed left to right.
// The translation unit is compiled with -ffp-contract=off, so no FMA is formed here.
__device__ __forceinline__ float pcm_sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    float d = dx * dx;
    d = d + dy * dy;
    d = d + dz * dz;
    return d;
}

// get_bt_idx of the reference (knn_query_cuda_kernel.cu:45-56): first i with q < off[i].
// With b > 0 (number of clouds known) the same answer by bisection: log2(b) dependent loads
// instead of up to b.  b <= 0 keeps the reference's linear scan (its ABI does not carry b).
__device__ __forceinline__ int pcm_cloud_of(int q, const int *__restrict__ off, int b)
{
    if (b <= 0) {
        int i = 0;
        while (!(q < off[i])) i++;
        return i;
    }
    int lo = 0, hi = b - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (q < off[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}
