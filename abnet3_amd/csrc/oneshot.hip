// oneshot.hip -- the small-message gradient all-reduce SURVEY.md section 5 / 8e prescribes for the data-parallel step
// (abnet3/trainer.py:239 -> :240: between loss.backward() and optimizer.step(); the reference itself has no multi-process
// path): a ONE-SHOT reduce-scatter + all-gather over peer-mapped buffers instead of a ring.
//
// A C2 step exchanges 2.29 MB (571 600 fp32 gradients) per ~160 us of compute.  On the 8-GPU node every GPU reaches every
// other over its own xGMI link (7 links x ~153 GB/s): a ring all-reduce moves 2 (R - 1) / R of the message through ONE link
// in 2 (R - 1) dependent steps -- per-link bound and latency-dominated (25-40 us exposed) -- while
// here every rank PUSHES shard s of its bucket straight into rank s's mailbox (R - 1 links at once, 286 KB each), sums the
// R contributions to its own shard in RANK ORDER (deterministic: every replica computes the same bits, replicas stay
// bit-identical), and pushes the reduced shard into every rank's result area (R - 1 links at once again): two hops.
//
// Everything is one kernel launch per rank on the caller's stream (graph-capturable; no host call inside): the peers'
// mailboxes are mapped once, by the caller, through hipIpcMemHandle (fine-grained device memory).  Hand-overs are flags
// in the RECEIVER's mailbox written after a system-scope release; tags are the mailbox's own call counter (every rank makes
// the same sequence of calls), so nothing is zeroed between calls.  Every spin is bounded: a rank whose peer never arrives
// raises its mailbox's failure word, hands NaN to its peers in place of the shard it could not reduce and leaves NaN in its
// own bucket: the step's gradients read NaN on every rank (loud, and the same on all replicas), and the call returns.
//
// Mailbox of rank s (abn_oneshot_mail_bytes; zero before the first call):
//   header   256 B   word 0: calls made; word 1: workgroups of the call in flight that have finished phase 1;
//                    word 2: ... phase 2; word 3: ... the call; word 16: failure
//   flags1   R x 128 B   flags1[r] = tag: rank r's contribution to MY shard is in slots[r]
//   flags2   R x 128 B   flags2[r] = tag: rank r's reduced shard is in result
//   slots    R x shard_cap floats
//   result   cap floats
#include <stdlib.h>

#include "common.h"

namespace abn {

constexpr int OS_HDR = 256;
constexpr int OS_NT = 256;
// how long a rank waits for a peer before it gives up, on the chip-wide 100 MHz clock: a peer may be late by as much as a
// garbage collection, a slow loader or a checkpoint being written, none of which is an error -- the bound is there so that a
// peer that DIED does not hang the stream for ever (torch.distributed's own collectives wait minutes)
#ifndef OS_WAIT_TICKS
#define OS_WAIT_TICKS 6000000000ull        // 60 s
#endif

__host__ __device__ inline int64_t os_shard_cap(int64_t cap, int world) { return ((cap + world - 1) / world + 63) / 64 * 64; }
__host__ __device__ inline int64_t os_flags1_off() { return OS_HDR; }
__host__ __device__ inline int64_t os_flags2_off(int world) { return OS_HDR + (int64_t)world * 128; }
__host__ __device__ inline int64_t os_slots_off(int world) { return OS_HDR + (int64_t)world * 256; }
__host__ __device__ inline int64_t os_result_off(int64_t cap, int world) { return os_slots_off(world) + (int64_t)world * os_shard_cap(cap, world) * 4; }
__host__ __device__ inline int64_t os_mail_bytes(int64_t cap, int world) { return os_result_off(cap, world) + (cap + 63) / 64 * 64 * 4; }

struct OneShotP {
    int rank, world;
    char* mail[ABN_ONESHOT_MAX_RANKS];
    int64_t cap;
    float* buf;
    int64_t n;
};

typedef float os4 __attribute__((ext_vector_type(4)));

// true once *p == tag (system scope); false: gave up (the failure word of the local mailbox is raised)
__device__ __forceinline__ bool os_wait(const unsigned* p, unsigned tag, unsigned* fail)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == tag) return true;
        if ((spins & 63u) == 63u && (__builtin_amdgcn_s_memrealtime() - t0 > OS_WAIT_TICKS ||
                                     __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) {
            __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        if (spins < 4096u) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(64);
    }
}

// "every workgroup of this launch has finished the phase": the last one to arrive (a device-scope ticket in the local
// header) raises the phase's flags in every peer's mailbox -- behind a system-scope release of everything this rank wrote
__device__ __forceinline__ void os_phase_done(const OneShotP& q, unsigned* hdr, int word, int64_t flags_off, unsigned tag)
{
    __threadfence_system();                       // this workgroup's pushes are visible to the peers
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(hdr + word, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(hdr + word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int s = 0; s < q.world; ++s)
                __hip_atomic_store(reinterpret_cast<unsigned*>(q.mail[s] + flags_off + (int64_t)q.rank * 128), tag, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(OS_NT) void oneshot_allreduce_kernel(OneShotP q)
{
    __shared__ int ok_s;
    const int R = q.world, me = q.rank;
    char* const mine = q.mail[me];
    unsigned* const hdr = reinterpret_cast<unsigned*>(mine);
    const unsigned tag = hdr[0] + 1u;             // (advanced by the last workgroup to leave: every workgroup read it at its start)
    const int64_t shard = os_shard_cap(q.cap, R);
    const int64_t n4 = (q.n + 3) / 4;             // buf is padded to 4 floats by the caller's contract (n % 4 == 0)
    const int64_t per = ((q.n + R - 1) / R + 3) / 4 * 4;       // floats of a shard of THIS call (the last one may be short)
    // ---- phase 1: my share of shard s -> rank s's slots[me]
    for (int ds = 0; ds < R; ++ds) {
        const int s = (me + ds) % R;              // (every rank starts with another peer: the links are used together)
        const int64_t lo = (int64_t)s * per, hi = min(lo + per, q.n);
        float* const dst = reinterpret_cast<float*>(q.mail[s] + os_slots_off(R)) + (int64_t)me * shard;
        for (int64_t i = lo / 4 + (int64_t)blockIdx.x * OS_NT + threadIdx.x; i < (hi + 3) / 4; i += (int64_t)gridDim.x * OS_NT)
            reinterpret_cast<os4*>(dst)[i - lo / 4] = reinterpret_cast<const os4*>(q.buf)[i];
    }
    (void)n4;
    os_phase_done(q, hdr, 1, os_flags1_off(), tag);
    // ---- phase 2: the R contributions to MY shard, summed in rank order, -> every rank's result
    if (threadIdx.x == 0) {
        bool ok = true;
        for (int r = 0; r < R && ok; ++r) ok = os_wait(reinterpret_cast<const unsigned*>(mine + os_flags1_off() + (int64_t)r * 128), tag, hdr + 16);
        ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    bool ok = ok_s != 0;
    {
        // A rank that gave up on a contribution has no sum to hand out -- and must not let its peers take the previous call's
        // shard (or zeros) for one: it hands out NaN instead, so that the bucket reads NaN on EVERY rank and the replicas fail
        // together instead of drifting apart.
        if (ok) __threadfence_system();           // (acquire: the slots' bytes behind the flags)
        const int64_t lo = (int64_t)me * per, hi = min(lo + per, q.n);
        const float* const slots = reinterpret_cast<const float*>(mine + os_slots_off(R));
        const float nanv = __builtin_nanf("");
        for (int64_t i = (int64_t)blockIdx.x * OS_NT + threadIdx.x; i < (hi - lo + 3) / 4; i += (int64_t)gridDim.x * OS_NT) {
            os4 acc = {nanv, nanv, nanv, nanv};
            if (ok) {
                acc = __builtin_nontemporal_load(reinterpret_cast<const os4*>(slots) + i);
                for (int r = 1; r < R; ++r) acc += __builtin_nontemporal_load(reinterpret_cast<const os4*>(slots + (int64_t)r * shard) + i);
            }
            for (int ds = 0; ds < R; ++ds) {
                const int s = (me + ds) % R;
                reinterpret_cast<os4*>(reinterpret_cast<float*>(q.mail[s] + os_result_off(q.cap, R)) + lo)[i] = acc;
            }
        }
    }
    __syncthreads();
    os_phase_done(q, hdr, 2, os_flags2_off(R), tag);
    // ---- phase 3: every rank's reduced shard is in my result -> the bucket
    if (threadIdx.x == 0) {
        bool ok2 = ok;
        for (int r = 0; r < R && ok2; ++r) ok2 = os_wait(reinterpret_cast<const unsigned*>(mine + os_flags2_off(R) + (int64_t)r * 128), tag, hdr + 16);
        ok_s = ok2 ? 1 : 0;
    }
    __syncthreads();
    ok = ok_s != 0;
    __threadfence_system();
    const float* const res = reinterpret_cast<const float*>(mine + os_result_off(q.cap, R));
    for (int64_t i = (int64_t)blockIdx.x * OS_NT + threadIdx.x; i < (q.n + 3) / 4; i += (int64_t)gridDim.x * OS_NT) {
        os4 v = __builtin_nontemporal_load(reinterpret_cast<const os4*>(res) + i);
        if (!ok) v = os4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        reinterpret_cast<os4*>(q.buf)[i] = v;
    }
    // ---- the call is over on this rank once every workgroup is here: the counter moves on
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(hdr + 3, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(hdr + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(hdr, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace abn

using namespace abn;

extern "C" {

int64_t abn_oneshot_mail_bytes(int32_t world, int64_t cap_floats)
{
    if (world < 1 || world > ABN_ONESHOT_MAX_RANKS || cap_floats < 1) return -1;
    return os_mail_bytes(cap_floats, world);
}

int abn_allreduce_oneshot(const abn_oneshot_ctx* ctx, float* buf, int64_t n, void* stream)
{
    ABN_REQUIRE(ctx != nullptr && buf != nullptr, "allreduce_oneshot: null argument");
    ABN_REQUIRE(ctx->world >= 1 && ctx->world <= ABN_ONESHOT_MAX_RANKS && ctx->rank >= 0 && ctx->rank < ctx->world,
                "allreduce_oneshot: rank %d of %d", ctx->rank, ctx->world);
    ABN_REQUIRE(n >= 0 && n <= ctx->cap_floats && n % 4 == 0, "allreduce_oneshot: n=%lld must be a multiple of 4 and at most the mailboxes' capacity %lld",
                (long long)n, (long long)ctx->cap_floats);
    ABN_REQUIRE(aligned16(buf), "allreduce_oneshot: the bucket must be 16-byte aligned");
    if (n == 0) return ABN_OK;
    OneShotP q = {};
    q.rank = ctx->rank; q.world = ctx->world; q.cap = ctx->cap_floats; q.buf = buf; q.n = n;
    for (int s = 0; s < ctx->world; ++s) {
        ABN_REQUIRE(ctx->mail[s] != nullptr && aligned16(ctx->mail[s]), "allreduce_oneshot: mailbox %d is not mapped", s);
        q.mail[s] = reinterpret_cast<char*>(ctx->mail[s]);
    }
    // a small grid: the call is bound by the links and the hand-overs' latency, not by the CUs, and every workgroup of every
    // rank must be resident while it waits for the peers (two ranks may share one GPU in the tests)
    // (ABN_ONESHOT_WGS lifts the cap of 32 workgroups: DESIGN.md section 4 has the two-process timings at 32 / 64 / 128)
    const int64_t cap_wgs = switches().oneshot_wgs;
    int64_t wgs = (n / 4 + OS_NT * 8 - 1) / (OS_NT * 8);
    wgs = wgs < 1 ? 1 : (wgs > cap_wgs ? cap_wgs : wgs);
    if (wgs > 256) wgs = 256;                     // every workgroup resident
    hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3((unsigned)wgs), dim3(OS_NT), 0, (hipStream_t)stream, q);
    ABN_CHECK_LAUNCH("allreduce_oneshot");
    return ABN_OK;
}

}  // extern "C"
