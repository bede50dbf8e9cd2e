// One forward attention block of the training step per workgroup (8 windows x 8 heads), alone on the chip, with wall-clock stamps:
// where do its ~9 us beyond the launch go?  Variants: regs (round-3 block), dma (LDS-DMA staged block).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I koemorph_amd/csrc tools/micro/attn_train_bench.hip -o tools/micro/bin/attn_train_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "koemorph.h"
#include "km_device.h"
namespace km {
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ unsigned long long* g_stamps;
#define KM_TILE_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#include "km_train_attn_dev.h"
template <int V>
__global__ __launch_bounds__(256, 2) void k_attn(ElemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    KM_TILE_STAMP(0);
    if (V == 0) attn_fwd_mfma_dev<32>(a, blockIdx.x, smem); else attn_fwd_dma32_dev<5>(a, blockIdx.x, smem);
    KM_TILE_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KM_TILE_STAMP(7);
}
__global__ void k_fill(float* p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v * (float)((i * 2654435761u) % 1000) * 1e-3f - v * 0.5f;
}
__global__ void k_empty() {}
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
    using namespace km;
    const int B = 8, H = 8, d = 256, NK = 80, reps = 200;
    int drop = argc > 1 ? atoi(argv[1]) : 1;
    float *Q, *KV, *P[2], *A[2]; unsigned char* mask;
    CK(hipMalloc(&Q, 28 * d * 4)); CK(hipMalloc(&KV, (size_t)B * NK * 2 * d * 4));
    for (int v = 0; v < 2; ++v) { CK(hipMalloc(&P[v], (size_t)B * H * 28 * NK * 4)); CK(hipMalloc(&A[v], (size_t)B * 28 * d * 4)); }
    CK(hipMalloc(&mask, (size_t)B * H * 28 * NK));
    std::vector<unsigned char> hm((size_t)B * H * 28 * NK);
    for (size_t i = 0; i < hm.size(); ++i) hm[i] = (i * 7919u) % 10 != 0;
    CK(hipMemcpy(mask, hm.data(), hm.size(), hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, Q, (size_t)28 * d, 0.3f);
    unsigned long long* stamps; CK(hipMalloc(&stamps, 64 * 8 * 8)); CK(hipMemset(stamps, 0, 64 * 8 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const float sc = 1.0f / sqrtf(32.f); unsigned scb; memcpy(&scb, &sc, 4);
    for (int v = 0; v < 2; ++v) {
        ElemArgs a{};
        a.p0 = Q; a.p1 = KV; a.q0 = P[v]; a.q1 = A[v]; a.i0 = d; a.i1 = 32; a.i2 = NK; a.i3 = H; a.mask = drop ? mask : nullptr; a.f0 = 1.0f / 0.9f; a.u0 = scb;
        const size_t lds = (v == 0 ? (size_t)attn_mfma_fwd_lds_floats(32, NK) : (size_t)kAttnDmaFwdLdsFloats) * 4;
        auto launch = [&]() {
            if (v == 0) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k_attn<0>, dim3(B * H), dim3(256), lds, st, a); }
            else { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k_attn<1>, dim3(B * H), dim3(256), lds, st, a); }
        };
        for (int i = 0; i < 3000; ++i) { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, KV, (size_t)B * NK * 2 * d, 0.5f); launch(); }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, KV, (size_t)B * NK * 2 * d, 0.5f); launch(); }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms = 0, ms0 = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, KV, (size_t)B * NK * 2 * d, 0.5f); hipLaunchKernelGGL(k_empty, dim3(B * H), dim3(256), 0, st); }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms0, e0, e1));
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, KV, (size_t)B * NK * 2 * d, 0.5f);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
        launch(); CK(hipStreamSynchronize(st));
        unsigned long long* nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &nul, sizeof(nul)));
        std::vector<unsigned long long> h(64 * 8); CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double avg[8] = {0}; unsigned long long first = ~0ull, last = 0;
        for (int b = 0; b < 64; ++b) { if (h[b * 8] < first) first = h[b * 8]; if (h[b * 8 + 7] > last) last = h[b * 8 + 7];
            for (int i = 0; i < 8; ++i) if (h[b * 8 + i]) avg[i] += (double)(h[b * 8 + i] - h[b * 8]) * 0.01 / 64; }
        printf("%s (dropout %d): %.2f us per launch beyond an empty launch; stamps (us after entry): requested %.2f  landed+barrier %.2f  scores done %.2f  softmax done %.2f  end %.2f  drained %.2f; first entry -> last exit %.2f\n",
               v == 0 ? "regs" : "dma ", drop, (ms - ms0) * 1e3 / reps, avg[1], avg[2], avg[4], avg[3], avg[6], avg[7], (double)(last - first) * 0.01);
    }
    std::vector<float> p0((size_t)B * H * 28 * NK), p1(p0.size()), a0((size_t)B * 28 * d), a1(a0.size());
    CK(hipMemcpy(p0.data(), P[0], p0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), P[1], p1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a0.data(), A[0], a0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a1.data(), A[1], a1.size() * 4, hipMemcpyDeviceToHost));
    double dp = 0, da = 0; for (size_t i = 0; i < p0.size(); ++i) dp = fmax(dp, fabs((double)p0[i] - p1[i])); for (size_t i = 0; i < a0.size(); ++i) da = fmax(da, fabs((double)a0[i] - a1[i]));
    printf("max |P_dma - P_regs| %.3g, max |A_dma - A_regs| %.3g (A[0] = %g)\n", dp, da, a0[0]);
    return 0;
}
