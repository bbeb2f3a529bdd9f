// handoff_probe.hip -- what a hand-over between two workgroups costs on this chip, by where they sit and how they store.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/handoff_probe.hip -o tools/probes/handoff_probe.bin && tools/probes/handoff_probe.bin
// 256 workgroups of one wavefront (one per CU).  Every workgroup publishes the XCD it runs on (HW_REG_XCC_ID); workgroup 0
// picks a partner on its own XCD and one on another, and plays ping-pong with each: a 16-byte granule {.., tag} stored one
// way, polled with sc1 loads, answered the other way.  The round trip is timed on workgroup 0's clock (s_memtime; the XCDs
// count from bases of their own, so only a there-and-back is measurable), a hop = half of it.  Store kinds:
//   0  sc1 (write-through, leaves the L2: what tower_bn_persist.h uses; valid between any two workgroups)
//   1  plain (stays in the storing XCD's L2: can only ever be seen by a reader on the SAME XCD -- the question is whether
//      such a reader sees it at all with sc1 loads, and how much sooner)
// The other 253 workgroups leave at once: these are latencies on an idle chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef int v4i __attribute__((ext_vector_type(4)));

struct Probe {
    unsigned* xcc;          // [256] XCC_ID + 1 (0: not yet published)
    unsigned* role;         // [256] 0 unknown, 1 leave, 2 echo
    v4i* ab;                // [2 partners][2 kinds] mailbox 0 -> partner
    v4i* ba;                // [2 partners][2 kinds] mailbox partner -> 0
    unsigned long long* out;   // [2 partners][2 kinds][2]: cycles, rounds seen (0: gave up)
    int* partner;           // [2]
    int rounds;
};

__device__ __forceinline__ void st16(v4i* p, v4i v, int kind)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 16, 0x00020000);
    if (kind == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rs, 0, 0, 16);      // sc1
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, 0, 0, 0);                  // plain
}
__device__ __forceinline__ v4i ld16_sc1(v4i* p)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 16, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 16);
}
__device__ __forceinline__ bool wait_tag(v4i* p, int tag)
{
    for (int spin = 0; spin < (1 << 18); ++spin) {
        if (ld16_sc1(p)[3] == tag) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

__global__ void probe_kernel(Probe P)
{
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    __hip_atomic_store(&P.xcc[b], xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0) {
        // wait for everybody's XCD, pick the partners
        int same = -1, other = -1;
        for (int w = 1; w < (int)gridDim.x; ++w) {
            unsigned v = 0;
            for (int spin = 0; spin < (1 << 20) && v == 0; ++spin) v = __hip_atomic_load(&P.xcc[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == 0) continue;
            if (v - 1u == xcc && same < 0) same = w;
            if (v - 1u != xcc && other < 0) other = w;
        }
        __hip_atomic_store(&P.partner[0], same, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&P.partner[1], other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        for (int w = 1; w < (int)gridDim.x; ++w)
            __hip_atomic_store(&P.role[w], (w == same || w == other) ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int pi = 0; pi < 2; ++pi) {
            const int w = pi == 0 ? same : other;
            if (w < 0) continue;
            for (int kind = 0; kind < 2; ++kind) {
                if (kind == 1 && pi == 1) continue;                 // plain stores never leave the XCD
                v4i* ab = P.ab + (pi * 2 + kind) * 16, * ba = P.ba + (pi * 2 + kind) * 16;      // 256 bytes apart: lines of their own
                unsigned long long cyc = 0; int seen = 0;
                for (int r = 1; r <= P.rounds; ++r) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                    st16(ab, v4i{r, r, r, r}, kind);
                    if (!wait_tag(ba, r)) break;
                    cyc += __builtin_amdgcn_s_memtime() - t0;
                    ++seen;
                }
                // release the partner from this kind (it waits for tag rounds + 1; sc1: it must see it)
                st16(ab, v4i{0, 0, 0, P.rounds + 1}, 0);
                P.out[(pi * 2 + kind) * 2] = cyc;
                P.out[(pi * 2 + kind) * 2 + 1] = (unsigned long long)seen;
            }
        }
        return;
    }
    unsigned role = 0;
    for (int spin = 0; spin < (1 << 22) && role == 0; ++spin) role = __hip_atomic_load(&P.role[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (role != 2u) return;
    // which partner am I?  (the partner table is written before the roles, through the same L2 path: read it sc1-wise)
    const int pi = __hip_atomic_load(&P.partner[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == b ? 0 : 1;
    for (int kind = 0; kind < 2; ++kind) {
        if (kind == 1 && pi == 1) continue;
        v4i* ab = P.ab + (pi * 2 + kind) * 16, * ba = P.ba + (pi * 2 + kind) * 16;
        for (int r = 1; r <= P.rounds + 1; ++r) {
            // wait for round r, or for the release tag
            bool got = false, released = false;
            for (int spin = 0; spin < (1 << 18) && !got; ++spin) {
                const int t = ld16_sc1(ab)[3];
                if (t == r) got = true;
                else if (t == P.rounds + 1) { got = true; released = true; }
                else __builtin_amdgcn_s_sleep(1);
            }
            if (!got || released || r == P.rounds + 1) break;
            st16(ba, v4i{r, r, r, r}, kind);
        }
    }
}

int main()
{
    const int G = 256, rounds = 200;
    Probe P;
    hipMalloc(&P.xcc, G * 4); hipMalloc(&P.role, G * 4);
    P.ab = nullptr; P.ba = nullptr;
    hipMalloc(&P.out, 8 * 8); hipMalloc(&P.partner, 8);
    v4i* abase; v4i* bbase;                                          // the two directions in allocations of their own
    hipMalloc(&abase, 4 * 256 * 16); hipMalloc(&bbase, 4 * 256 * 16);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(P.xcc, 0, G * 4); hipMemset(P.role, 0, G * 4); hipMemset(abase, 0, 4 * 256 * 16); hipMemset(bbase, 0, 4 * 256 * 16);
        hipMemset(P.out, 0, 64); hipMemset(P.partner, 0xff, 8);
        P.ab = abase; P.ba = bbase; P.rounds = rounds;
        hipLaunchKernelGGL(probe_kernel, dim3(G), dim3(64), 0, 0, P);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        unsigned long long out[8]; int partner[2]; unsigned xcc[G];
        hipMemcpy(out, P.out, 64, hipMemcpyDeviceToHost); hipMemcpy(partner, P.partner, 8, hipMemcpyDeviceToHost);
        hipMemcpy(xcc, P.xcc, G * 4, hipMemcpyDeviceToHost);
        int rr = 0;
        for (int b = 0; b < G; ++b) rr += (xcc[b] == xcc[b % 8]);
        printf("run %d: workgroup 0 on XCD %u, partner on the same XCD: %d, on another: %d; blocks b and b %% 8 on one XCD: %d of %d\n", rep, xcc[0] - 1, partner[0], partner[1], rr, G);
        const char* names[4] = {"same XCD, sc1 stores  ", "same XCD, plain stores", "other XCD, sc1 stores ", "(unused)"};
        for (int i = 0; i < 3; ++i) {
            if (out[2 * i + 1] == 0) printf("  %s: never seen (0 of %d rounds)\n", names[i], rounds);
            else printf("  %s: %llu of %d rounds, round trip %.0f cycles, one hop %.0f\n", names[i], out[2 * i + 1], rounds,
                        (double)out[2 * i] / out[2 * i + 1], (double)out[2 * i] / out[2 * i + 1] / 2);
        }
    }
    return 0;
}
