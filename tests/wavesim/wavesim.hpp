// wavesim.hpp -- a CPU execution model of the gfx950 wave64 programming model, for TESTS ONLY.
//
// Purpose: run the product's HIP kernel SOURCES (pointcloudmatters_amd/csrc/*.hip, unmodified except for three mechanical
// rewrites listed in tests/wavesim/build.py) on the host, through the same extern "C" entry points, so that their results can be
// compared with the CPU oracle in the `-m "not gpu"` suite -- evidence about the kernels' logic (index arithmetic, tie orders,
// reductions, cross-lane exchanges) that does not need a GPU.  It is NOT a fallback: nothing under pointcloudmatters_amd/ can
// reach it, it models no timing, no memory hierarchy and no data races (lanes run one after another), and the GPU parity tests
// remain the evidence for the hardware.
//
// Model.  One workgroup at a time.  Every work-item is a fiber (own stack, hand-written x86-64 context switch).  A fiber runs
// until it (a) returns, (b) reaches __syncthreads(), or (c) reaches a cross-lane operation (shuffle, DPP, ballot, readlane,
// bpermute, wave barrier, MFMA).  When no fiber of the workgroup can run, the scheduler resolves what they wait for:
//   * per wave, the lanes waiting at the SAME cross-lane operation (same call site, same opcode) form the active set -- the EXEC
//     mask: lanes that returned, sit in another branch, or wait at a barrier are inactive -- the operation is evaluated for the
//     set with the ISA's semantics and the lanes continue;
//   * when every live lane of the workgroup waits at __syncthreads(), the barrier opens (waves that ended do not count: s_barrier).
// Arithmetic is the host's IEEE fp32 (compile with -ffp-contract=off): the same un-contracted operations the device code is built
// with; transcendental builtins (v_exp_f32 ...) are the libm functions -- kernels that use them are tolerance-tested, not bit-tested.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

namespace wavesim {

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

enum State { READY, AT_BARRIER, AT_CROSS, DONE };
enum Op { OP_SHFL, OP_DPP, OP_BALLOT, OP_READLANE, OP_READFIRST, OP_BPERMUTE, OP_WAVE_BARRIER, OP_MFMA, OP_ANY_ALL, OP_TR16 };

// A cross-lane call site: a unique marker address plus its place in program order (0 for headers / 1 for the translation unit's own
// file in the top bits, the source line below).  When lanes of one wave wait at DIFFERENT sites, the site that comes first in program
// order is resolved first and its lanes run on -- like the hardware, which finishes a divergent branch before the lanes that skipped
// it continue past the reconvergence point: the lanes then usually meet at the later site and execute it together.
struct Site {
    const void *marker;
    uint32_t order;
};

struct Lane {
    void *sp = nullptr;       // saved stack pointer while switched out
    char *stack = nullptr;
    dim3 tid;
    int linear = 0, lane = 0, wave = 0;
    State state = READY;
    // cross-lane mailbox
    Op op;
    const void *site;         // marker of the call site: lanes at different call sites never exchange
    uint32_t order;           // program-order key of the site
    uint64_t in[2];           // payload (value, old value)
    int64_t arg[4];           // operation parameters
    uint64_t out;
    const void *pin;          // MFMA: pointers to this lane's operand / result registers
    void *pout;
    bool barrier_pred = false;  // __syncthreads_or
};

struct Block {
    dim3 bid, bdim, gdim;
    std::vector<Lane> lanes;
    bool barrier_or = false;
};

extern Lane *g_cur;
extern Block *g_blk;
extern void *g_sched_sp;
extern char *g_dyn_smem;
extern size_t g_dyn_smem_bytes;
extern long g_stat_switches, g_stat_cross, g_stat_barriers;

extern "C" void wavesim_switch(void **save_sp, void *load_sp);

inline void yield_to_scheduler() { wavesim_switch(&g_cur->sp, g_sched_sp); }

inline char *dyn_smem() { return g_dyn_smem; }

// ---- cross-lane plumbing: deposit, wait for the resolver, pick up ------------------------------------------------------------
inline uint64_t cross(Op op, Site site, uint64_t v0, uint64_t v1, int64_t a0 = 0, int64_t a1 = 0, int64_t a2 = 0, int64_t a3 = 0,
                      const void *pin = nullptr, void *pout = nullptr)
{
    Lane *l = g_cur;
    l->op = op, l->site = site.marker, l->order = site.order, l->in[0] = v0, l->in[1] = v1;
    l->arg[0] = a0, l->arg[1] = a1, l->arg[2] = a2, l->arg[3] = a3;
    l->pin = pin, l->pout = pout;
    l->state = AT_CROSS;
    yield_to_scheduler();
    return l->out;
}

template <class T> inline uint64_t to_bits(T v)
{
    static_assert(sizeof(T) <= 8, "cross-lane payloads are at most 64 bits");
    uint64_t b = 0;
    std::memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T> inline T from_bits(uint64_t b)
{
    T v;
    std::memcpy(&v, &b, sizeof(T));
    return v;
}

void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t dyn_bytes);
void note_launch(const char *kernel_expr);  // launch census: counts per call-site kernel expression
void syncthreads();
int syncthreads_or(int pred);

// MFMA shapes (operands described to the resolver through arg[])
enum Mfma { MFMA_32x32x8_BF16_1K, MFMA_32x32x16_BF16, MFMA_16x16x32_BF16 };

}  // namespace wavesim
