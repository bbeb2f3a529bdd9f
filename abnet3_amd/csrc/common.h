// common.h -- error plumbing shared by the translation units of libabnet3_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/abnet3_hip.h"

namespace abn {

void set_error(const char* fmt, ...);

#define ABN_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            ::abn::set_error(__VA_ARGS__);          \
            return ABN_E_ARG;                       \
        }                                           \
    } while (0)

// Launch errors only: never synchronises (callers may be capturing a graph).
#define ABN_CHECK_LAUNCH(what)                                               \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            ::abn::set_error("%s: %s", what, hipGetErrorString(e_));         \
            return ABN_E_LAUNCH;                                             \
        }                                                                    \
    } while (0)

// The library's A/B switches (kernel choice only, never results beyond fp32 summation order), read from
// the environment ONCE, at the first call that asks.  abn_reload_switches() (tests and A/B tools that flip
// a switch inside one process) reads them again.
struct Switches {
    bool planes;              // ABN_PLANES=0: never the operand-plane kernels
    bool fused;               // ABN_FUSED=0: per-layer kernels only
    int64_t fused_min_rows;   // ABN_FUSED_MIN_ROWS: -1 = each path's own default
    bool bn_planes;           // ABN_BN_PLANES=0: BatchNorm training on the per-layer kernels
    bool bn_persist;          // ABN_BN_PERSIST=0: BatchNorm training one launch per layer (tower_planes.h), never the resident tower (tower_bn_persist.h)
    bool wgrad_xcd;           // ABN_WGRAD_XCD=0: weight-gradient workgroups in launch order
    bool bf16x3_planes;       // ABN_BF16X3_PLANES=0: (GEMM kernels) split per fragment instead of per tile
    bool bwd_pair;            // ABN_BWD_PAIR=0: wgrad and dgrad of a layer as two grids
    int gemm_tile;            // ABN_GEMM_TILE=0..3: force a tile shape
    bool dtw_f40, dtw_pc;     // ABN_DTW_F40=0 / ABN_DTW_PC=0: the general DTW kernels
    int wgrad_tile128;        // ABN_WGRAD_TILE128: 128 x 128 weight-gradient tiles (fp16 x 2) never (0) / always (1) / from 2048 rows (default)
    int dtw_wgs_per_cu;       // ABN_DTW_WGS: workgroups per CU of the gang DTW kernel's persistent grid (default 6)
    int wgrad_wgs_heavy, wgrad_wgs_light;   // ABN_WGRAD_WGS_HEAVY / _LIGHT: workgroups per layer of the 128 x 128 weight-gradient launch (tiles x slabs; >= 16 tiles: heavy)
    int64_t wgrad_rows_per_slab;   // ABN_WGRAD_ROWS_PER_SLAB: fewest batch rows one split-K slab of the weight gradients sums (default 128)
    bool wide;                // ABN_WIDE=0: small batches on the single-launch chains / per-layer GEMMs, not tower_wide.h
    int64_t wide_max_rows;    // ABN_WIDE_MAX_ROWS: most (virtual) rows the layer-per-launch kernels take, -1 = default
    int wide_max_groups;      // ABN_WIDE_MAXG: most workgroups per 32-row block of a layer-per-launch kernel (1 .. 8, default 8)
    bool wgrad_step;          // ABN_WGRAD_STEP=0: small batches' weight gradients as slabs + slab_reduce_step_kernel, never tower_wgrad_step.h's one launch
    bool dtw_dealt;           // ABN_DTW_SCHED=0: the gang DTW kernel on round 4's schedule (a pair per slot), not the dealt one
    int oneshot_wgs;          // ABN_ONESHOT_WGS: most workgroups of abn_allreduce_oneshot (default 32, <= 256)
};
const Switches& switches();
void reload_switches();

// ---------------------------------------------------------------------------------------------------------
// "The last workgroup adds up": every workgroup publishes one float64 partial and draws a ticket; the one that
// draws the last ticket sums all partials in a FIXED order (deterministic, unlike an atomic sum).  Used by the
// pair loss (loss.hip) and by the loss phases of the data-gradient launches (tower_planes.h, tower_wide.h).
//
// What has to hold: when a workgroup reads ticket n - 1, the other n - 1 partials are visible to it.
//  * C++ memory model form (-DABN_STRICT_FENCES): the ticket is an acq_rel read-modify-write at agent scope.  On
//    gfx942 / gfx950 a release at agent scope is `buffer_wbl2 sc1` -- write back EVERY dirty line of this XCD's L2 --
//    and an acquire `buffer_inv sc1`: with the 3 MB of loss gradients a workgroup has just written that is a
//    cache-wide flush per workgroup (pair_loss_kernel, 4096 pairs, rocprofv3: 17.6 us against 11.7; tools/loss_fence_ab.py).
//  * Default form: the partial is stored with a relaxed AGENT-scope atomic store -- per the AMDGPU memory model
//    (LLVM AMDGPUUsage, gfx942 table: "store atomic monotonic agent: global_store sc1=1") a write-through to the
//    level all XCDs share --, `s_waitcnt vmcnt(0)` then holds the wave until that store is acknowledged, and only
//    then is the ticket drawn -- itself a relaxed agent-scope atomic, executed at that same level.  The partials are
//    read back with relaxed agent-scope atomic loads (sc1: they bypass the non-coherent levels).  The only thing a
//    release would add is the write-back of the workgroup's OTHER (non-atomic) stores, which the summing workgroup
//    never reads.  The signal fences keep the compiler from moving the store or the ticket across the drain.
// Both forms give the same bits (the sum's order is fixed); tools/variants.sh builds the strict one for comparison.
__device__ __forceinline__ bool abn_ticket_publish(double* slot, double value, unsigned* counter, unsigned n_tickets)
{
    __hip_atomic_store(slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef ABN_STRICT_FENCES
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    return ticket == n_tickets - 1;
}
__device__ __forceinline__ double abn_ticket_partial(const double* slot)
{
    return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace abn
