// Cross-workgroup synchronisation inside ONE launch on gfx950 (8 XCDs with private L2s, per-CU vector L1s that other CUs'
// stores never refresh): the grid barrier and the store / acquire helpers of the persistent training program
// (km_trainp.hip) and of its timing harness (tools/micro/persist_bench.hip).
//
// Protocol (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility", Valid forms; recipe of
// cdna_hip_programming.md section 6, Guideline 16):
//   producer   every byte another workgroup will read is stored WRITE-THROUGH (sc1: st_wt below), so no L2 write-back
//              (release fence) is needed; every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at
//              its barrier, ONE lane signals with an agent-scope atomic add;
//   consumer   ONE lane polls ONE word with relaxed agent-scope loads (+ s_sleep), then ONE agent-scope acquire
//              (buffer_inv sc1: drops this CU's stale L1 lines), s_waitcnt vmcnt(0), the workgroup's barrier -- then plain loads.
// Every spin is bounded: a workgroup that waits longer than the limit sets the timeout word and returns false; every later
// wait of every workgroup sees the word and returns at once, so a protocol bug ends the launch with an error code instead of
// hanging the GPU.  All workgroups must be co-resident (the host sizes the grid from the occupancy query, minus a margin).
#pragma once

#include <hip/hip_runtime.h>

namespace kmsync {

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                  // global_load_dword sc1
}
__device__ __forceinline__ void st_relaxed(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                     // global_store_dword sc1
}
// write-through store of a value another workgroup reads later in this launch
__device__ __forceinline__ void st_wt(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_wt4(float* p, float4 v) {                                  // p 16-byte aligned
    typedef unsigned long long u64;
    union { float4 f; u64 u[2]; } c; c.f = v;
    __hip_atomic_store(reinterpret_cast<u64*>(p), c.u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<u64*>(p) + 1, c.u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool WT> __device__ __forceinline__ void st(float* p, float v) { if constexpr (WT) st_wt(p, v); else *p = v; }
template <bool WT> __device__ __forceinline__ void st4(float* p, float4 v) {
    if constexpr (WT) st_wt4(p, v); else *reinterpret_cast<float4*>(p) = v;
}

// State of one launch, in device memory, ALL ZERO before the launch (the host memsets it once at allocation; the last
// workgroup to leave a launch puts it back to zero, grid_exit):
//   word 0        arrivals at the flat barrier (monotonic over the launch: barrier k completes at k * nwg)
//   word 16       timeout / error word (0 = fine)
//   word 32       workgroups that have left the launch
//   words 64 + 32 s, s < 8          arrivals of shard s (hierarchical barrier), each on a 128-byte line of its own
//   words 320 + 32 s, s < 8         generation of shard s: the number of completed barriers, written by the shard's last arriver
//   word 576      arrivals of shard leaders at the top level
constexpr int kWords = 640;
constexpr int kTimeoutWord = 16, kExitWord = 32, kShard = 64, kGen = 320, kTop = 576;
constexpr unsigned kSpinLimit = 1u << 22;       // polls of ~0.1 us each: ~0.4 s, far beyond any phase of the step

__device__ __forceinline__ bool spin_until(const unsigned* word, unsigned target, unsigned* state) {
    unsigned spins = 0;
    while (ld_relaxed(word) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0u) {
            if (ld_relaxed(state + kTimeoutWord) != 0u) return false;
            if (spins >= kSpinLimit) { st_relaxed(state + kTimeoutWord, 1u); return false; }
        }
    }
    return true;
}

// Flat barrier: barrier number `epoch` (1, 2, ...) of this launch over `nwg` workgroups.  Called by every thread of every
// workgroup; stores of the phase before it must be write-through (st_wt) or followed by the caller's own release.
// Returns false when the launch is being abandoned (timeout word set).
// ok_lds: one int of the caller's (dynamic) LDS -- a static __shared__ here would shift the dynamic region's base off its
// 16-byte alignment (cdna_hip_programming.md, Guideline 17).
template <bool ACQUIRE = true>
__device__ __forceinline__ bool grid_barrier_flat(unsigned* state, unsigned epoch, unsigned nwg, int* ok_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = spin_until(state, epoch * nwg, state);
        if (ACQUIRE) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // buffer_inv sc1: this CU's L1 forgets what others rewrote
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *ok_lds = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// Hierarchical barrier: workgroups are dealt to 8 shards by blockIdx.x % 8 (under the observed round-robin placement a shard
// is the set of workgroups of one XCD -- a speed matter only, any placement is correct); a shard's last arriver joins the top
// level and, when all 8 leaders are there, publishes the shard's generation; everyone else polls its shard's generation word.
// 256 arrivals on one word cost ~3 us serialised at the memory side; 32 per shard in parallel + 8 at the top cost a fraction.
// nwg must be a multiple of 8.
template <bool ACQUIRE = true>
__device__ __forceinline__ bool grid_barrier_xcd(unsigned* state, unsigned epoch, unsigned nwg, int* ok_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned s = blockIdx.x & 7u, per = nwg >> 3;
        bool ok = true;
        const unsigned prev = __hip_atomic_fetch_add(state + kShard + 32 * s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == epoch * per) {                          // last of the shard: top level, then publish the generation
            __hip_atomic_fetch_add(state + kTop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = spin_until(state + kTop, epoch * 8u, state);
            st_relaxed(state + kGen + 32 * s, epoch);
        } else {
            ok = spin_until(state + kGen + 32 * s, epoch, state);
        }
        if (ACQUIRE) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *ok_lds = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// Leaving the launch: the last workgroup out puts the barrier state back to zero (the timeout word stays: the host reads
// and clears it), so the next launch -- or the next replay of a captured one -- starts clean without a memset node.
__device__ __forceinline__ void grid_exit(unsigned* state, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(state + kExitWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == nwg) {
            st_relaxed(state, 0u);
            st_relaxed(state + kTop, 0u);
            for (unsigned s = 0; s < 8u; ++s) { st_relaxed(state + kShard + 32 * s, 0u); st_relaxed(state + kGen + 32 * s, 0u); }
            st_relaxed(state + kExitWord, 0u);
        }
    }
}

}  // namespace kmsync
