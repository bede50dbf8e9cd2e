// What does a phase boundary cost on an MI355X: a dependent kernel launch, or a grid barrier inside one persistent launch?
// Decides the structure of the 8-window training step (km_trainp.hip).
//
// Workload: NPH dependent phases on 256 workgroups x 256 threads.  In phase p workgroup w reads the 16 KB slice that
// ANOTHER workgroup ((37 w + 11) mod nwg) wrote in phase p - 1, adds 1 and writes its own slice (ping-pong buffers, so the
// lines a CU reads in phase p are lines it read in phase p - 2: an L1-warm consumer, the case the visibility rules are about).
// The chain is checked: every value must equal NPH.  Optional busy work per phase (--work N: N dependent FMAs per thread).
//
//   launches      one launch per phase, 16-byte arguments
//   launches4k    one launch per phase, a 4000-byte by-value argument (phase_kernel's Phase struct)
//   flat          one launch, flat counter barrier, write-through stores + acquire
//   flat_fence    one launch, flat counter barrier, plain stores + release fence + acquire (the textbook form)
//   xcd           one launch, hierarchical barrier (8 shards + top), write-through stores + acquire
//
// hipcc --offload-arch=gfx950 -O3 -I koemorph_amd/csrc tools/micro/persist_bench.hip -o tools/micro/bin/persist_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "km_gridsync.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NT = 256, SLICE = 4096;      // floats per workgroup and phase = 16 KB

struct Big { int pad[1000]; };

template <bool WT>
__device__ __forceinline__ void phase_body(const float* prev, float* cur, int wg, int nwg, int work) {
    const int src = (37 * wg + 11) % nwg;
    const float4* s = reinterpret_cast<const float4*>(prev + (size_t)src * SLICE);
    float* d = cur + (size_t)wg * SLICE;
#pragma unroll
    for (int i = 0; i < SLICE / 4 / NT; ++i) {
        float4 v = s[threadIdx.x + NT * i];
        float w = 0.f;
        for (int k = 0; k < work; ++k) w = fmaf(w, 1.0000001f, v.x * 1e-30f);
        v.x += 1.f + w * 0.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        kmsync::st4<WT>(d + 4 * (threadIdx.x + NT * i), v);
    }
}

__global__ __launch_bounds__(NT) void k_phase(const float* prev, float* cur, int nwg, int work) {
    phase_body<false>(prev, cur, blockIdx.x, nwg, work);
}
__global__ __launch_bounds__(NT) void k_phase4k(Big b, const float* prev, float* cur, int nwg, int work) {
    if (b.pad[threadIdx.x] == 12345) return;
    phase_body<false>(prev, cur, blockIdx.x, nwg, work);
}

// MODE 0 flat + write-through, 1 flat + plain stores + release fence, 2 hierarchical + write-through
template <int MODE>
__global__ __launch_bounds__(NT) void k_persist(float* a, float* b, int nph, int work, unsigned* state) {
    __shared__ __attribute__((aligned(16))) int lds[4];
    const int nwg = gridDim.x;
    for (int p = 0; p < nph; ++p) {
        const float* prev = (p & 1) ? b : a;
        float* cur = (p & 1) ? a : b;
        if (MODE == 1) phase_body<false>(prev, cur, blockIdx.x, nwg, work);
        else phase_body<true>(prev, cur, blockIdx.x, nwg, work);
        if (p + 1 == nph) break;
        bool ok;
        if (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            ok = kmsync::grid_barrier_flat(state, (unsigned)(p + 1), (unsigned)nwg, lds);
        } else if (MODE == 0) {
            ok = kmsync::grid_barrier_flat(state, (unsigned)(p + 1), (unsigned)nwg, lds);
        } else {
            ok = kmsync::grid_barrier_xcd(state, (unsigned)(p + 1), (unsigned)nwg, lds);
        }
        if (!ok) break;
    }
    kmsync::grid_exit(state, (unsigned)nwg);
}

int main(int argc, char** argv) {
    int nwg = 256, nph = 14, work = 0, reps = 300, warm = 3000;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--wgs")) nwg = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--phases")) nph = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--work")) work = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--warm")) warm = atoi(argv[i + 1]);
    }
    if (nwg % 8 || nwg > 1024) { printf("--wgs must be a multiple of 8, at most 1024\n"); return 1; }
    float *a, *b;
    unsigned* state;
    const size_t bytes = (size_t)nwg * SLICE * sizeof(float);
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&state, kmsync::kWords * 4));
    CK(hipMemset(state, 0, kmsync::kWords * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> host((size_t)nwg * SLICE);
    Big big{};
    const char* names[5] = {"launches", "launches4k", "flat", "flat_fence", "xcd"};
    auto run = [&](int v) {
        CK(hipMemsetAsync(a, 0, bytes, st));
        if (v == 0) for (int p = 0; p < nph; ++p) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(NT), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, nwg, work);
        else if (v == 1) for (int p = 0; p < nph; ++p) hipLaunchKernelGGL(k_phase4k, dim3(nwg), dim3(NT), 0, st, big, (p & 1) ? b : a, (p & 1) ? a : b, nwg, work);
        else if (v == 2) hipLaunchKernelGGL(k_persist<0>, dim3(nwg), dim3(NT), 0, st, a, b, nph, work, state);
        else if (v == 3) hipLaunchKernelGGL(k_persist<1>, dim3(nwg), dim3(NT), 0, st, a, b, nph, work, state);
        else hipLaunchKernelGGL(k_persist<2>, dim3(nwg), dim3(NT), 0, st, a, b, nph, work, state);
    };
    printf("workgroups %d x %d threads, %d phases, 16 KB per workgroup and phase, work %d\n", nwg, NT, nph, work);
    for (int v = 0; v < 5; ++v) {
        for (int i = 0; i < warm / 10; ++i) run(v);
        CK(hipStreamSynchronize(st));
        // the memset is part of every repetition of every variant alike; it is timed separately below and subtracted
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) run(v);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        // check the chain of the last repetition
        float* last = (nph & 1) ? b : a;
        CK(hipMemcpy(host.data(), last, bytes, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (float x : host) bad += x != (float)nph;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) CK(hipMemsetAsync(a, 0, bytes, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms0 = 0.f; CK(hipEventElapsedTime(&ms0, e0, e1));
        unsigned tmo = 0;
        CK(hipMemcpy(&tmo, state + kmsync::kTimeoutWord, 4, hipMemcpyDeviceToHost));
        const double us = (ms - ms0) * 1e3 / reps;
        printf("%-11s %8.2f us per chain  %6.2f us per phase   wrong values %zu of %zu   timeout %u\n", names[v], us, us / nph, bad, host.size(), tmo);
        if (tmo) { CK(hipMemset(state, 0, kmsync::kWords * 4)); }
    }
    return 0;
}
