// Where does a small product of the 8-window training step spend its launch?  One C (M x N) = A (M x K) W^T product of the
// step's shapes through the tile code of km_gemm_dev.h, as its own launch, with wall-clock stamps (s_memrealtime, 100 MHz)
// taken by thread 0 of every workgroup: entry, operands initialised, first k-tile in LDS, k loop done, stores issued, stores
// drained.  A producer kernel rewrites A before every launch (as the previous phase of the step does: the operand does not
// sit in this XCD's L2).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I koemorph_amd/csrc -I tools/micro tools/micro/tile_bench.hip -o tools/micro/bin/tile_bench
//   tile_bench [--m 640] [--n 512] [--k 256] [--variant narrow2|narrow4|narrow8|wide2|wide4]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "km_gemm.h"

namespace km {
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ unsigned long long* g_stamps;
#define KM_TILE_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) { g_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); if ((i) == 0 || (i) == 7) g_stamps[(size_t)blockIdx.x * 8 + ((i) ? 5 : 4)] = clock64(); } } while (0)
#define TB_STAMP(i) KM_TILE_STAMP(i)
#include "km_gemm_dev.h"
#include "km_gemm_dma_dev.h"
#include "tile_ks_dev.h"

template <int BM, int D>
__global__ __launch_bounds__(256) void k_narrow(GemmArgs g, int gx, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    TB_STAMP(0);
    gemm_tile_dev<BM, D, true, true>(g, blockIdx.x % gx, blockIdx.x / gx, 0, smem);
    TB_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TB_STAMP(7);
}
template <int BM, int KG>
__global__ __launch_bounds__(256 * KG) void k_wide(GemmArgs g, int gx, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    TB_STAMP(0);
    gemm_tile_ks_dev<BM, KG, 2, true, true>(g, blockIdx.x % gx, blockIdx.x / gx, 0, smem);
    TB_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TB_STAMP(7);
}
template <int BM>
__global__ __launch_bounds__(256) void k_dma(GemmArgs g, int gx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    TB_STAMP(0);
    gemm_tile_dma_dev<BM, BM == 32 ? 8 : 4, 0, 0>(g, blockIdx.x % gx, blockIdx.x / gx, 0, smem);
    TB_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TB_STAMP(7);
}
__global__ void k_fill(float* p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i % 7) * 0.01f;
}
__global__ void k_empty() {}
}  // namespace km

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    using namespace km;
    int M = 640, N = 512, K = 256, bm = 32, reps = 200;
    std::string variant = "narrow2";
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--m")) M = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--n")) N = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--k")) K = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--bm")) bm = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--variant")) variant = argv[i + 1];
    }
    float *A, *W, *C;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    GemmArgs g{};
    g.alpha = 1.f; g.batch2 = 1; g.kb_count = 1;
    g.A = A; g.a_rs = K; g.a_cs = 1; g.B = W; g.b_rs = 1; g.b_cs = K; g.C = C; g.c_rs = N; g.M = M; g.N = N; g.K = K;
    const int gx = (N + 63) / 64, gy = (M + bm - 1) / bm, blocks = gx * gy;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, (size_t)blocks * 8 * 8));
    CK(hipMemset(stamps, 0, (size_t)blocks * 8 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, W, (size_t)N * K, 0.5f);
    auto launch = [&](unsigned long long* sp) {
#define NARROW(BM_, D_) do { size_t l = (size_t)ggd::lds_floats(BM_) * 4; hipFuncSetAttribute(reinterpret_cast<const void*>(&k_narrow<BM_, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((k_narrow<BM_, D_>), dim3(blocks), dim3(256), l, st, g, gx, sp); } while (0)
#define WIDE(BM_, KG_) do { size_t l = (size_t)KG_ * ggd::lds_floats(BM_) * 4; hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide<BM_, KG_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((k_wide<BM_, KG_>), dim3(blocks), dim3(256 * KG_), l, st, g, gx, sp); } while (0)
#define DMA(BM_) do { size_t l = (size_t)gdma::lds_floats(BM_, gdma::ring_stages(BM_, true)) * 4; hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<BM_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((k_dma<BM_>), dim3(blocks), dim3(256), l, st, g, gx); } while (0)
        if (variant == "dma") { if (bm == 32) DMA(32); else DMA(64); return; }
        if (bm == 32) {
            if (variant == "narrow2") NARROW(32, 2); else if (variant == "narrow4") NARROW(32, 4); else if (variant == "narrow8") NARROW(32, 8);
            else if (variant == "wide2") WIDE(32, 2); else if (variant == "wide4") WIDE(32, 4); else { printf("unknown variant\n"); exit(1); }
        } else {
            if (variant == "narrow2") NARROW(64, 2); else if (variant == "narrow4") NARROW(64, 4); else if (variant == "narrow8") NARROW(64, 8);
            else if (variant == "wide2") WIDE(64, 2); else if (variant == "wide4") WIDE(64, 4); else { printf("unknown variant\n"); exit(1); }
        }
    };
    // warm-up (clocks), then timed: producer + product per repetition; the producer alone is timed afterwards and subtracted
    for (int i = 0; i < 3000; ++i) { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, A, (size_t)M * K, 1.0f); launch(nullptr); }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, A, (size_t)M * K, 1.0f); launch(nullptr); }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms = 0, ms0 = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, A, (size_t)M * K, 1.0f); hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st); }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms0, e0, e1));
    // the variant against the register-staged tile on the same operands (claimed bit-identical)
    {
        std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
        const std::string keep = variant;
        CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 4, st));
        launch(nullptr); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
        variant = "narrow2";
        CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 4, st));
        launch(nullptr); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost));
        variant = keep;
        size_t diff = 0; double mx = 0;
        for (size_t i = 0; i < c0.size(); ++i) { if (memcmp(&c0[i], &c1[i], 4)) ++diff; const double d = fabs((double)c0[i] - c1[i]); if (d > mx || d != d) mx = d; }
        printf("%s vs narrow2: %zu of %zu elements differ bitwise, max |diff| %.3g (C[0] = %g, C[last] = %g)\n", variant.c_str(), diff, c0.size(), mx, c1[0], c1.back());
    }
    // one stamped launch
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, A, (size_t)M * K, 1.0f);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
    launch(stamps);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)blocks * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull, last = 0;
    for (int b = 0; b < blocks; ++b) { if (h[b * 8] < first) first = h[b * 8]; if (h[b * 8 + 7] > last) last = h[b * 8 + 7]; }
    double mhz = 0;
    for (int b = 0; b < blocks; ++b) mhz += (double)(h[b * 8 + 5] - h[b * 8 + 4]) / ((double)(h[b * 8 + 7] - h[b * 8]) * 0.01) / blocks;
    double avg[8] = {0}, mx[8] = {0};
    for (int b = 0; b < blocks; ++b)
        for (int i = 0; i < 8; ++i) {
            if (!h[b * 8 + i] || i == 4 || i == 5) continue;
            const double t = (double)(h[b * 8 + i] - h[b * 8]) * 0.01;
            avg[i] += t / blocks; if (t > mx[i]) mx[i] = t;
        }
    printf("%s bm %d: %d x %d x %d, %d workgroups: %.2f us per launch beyond an empty launch of the same grid (%.2f with it)\n", variant.c_str(), bm, M, N, K, blocks,
           (ms - ms0) * 1e3 / reps, ms * 1e3 / reps);
    printf("  in-kernel, us after the workgroup's entry (mean / max over workgroups): init %.2f/%.2f  first tile in LDS %.2f/%.2f  k loop done %.2f/%.2f  "
           "epilogue issued %.2f/%.2f  stores drained %.2f/%.2f;  first entry -> last exit %.2f us\n",
           avg[1], mx[1], avg[2], mx[2], avg[3], mx[3], avg[6], mx[6], avg[7], mx[7], (double)(last - first) * 0.01);
    printf("  s_memtime ticks per us of s_memrealtime inside the kernel: %.0f\n", mhz);
    return 0;
}
