// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate on gfx950 with 1 or 2 waves per SIMD and 10 / 20 independent
// accumulators per wave (the shapes of the fused core kernel's GEMM loops).  Wall-clock rate vs the 157.3 TFLOP/s peak
// tells how much of the "81-89 % of ideal" seen in those loops is the clock under sustained MFMA load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, float s) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = s;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int NACC> void run(const char* name, float* d, int threads) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f);
    hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
    printf("%-34s %8.3f ms  %7.2f TFLOP/s  (%.3f of 157.3)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<10>("10 acc, 512 thr (2 waves/SIMD)", d, 512);
    run<20>("20 acc, 512 thr (2 waves/SIMD)", d, 512);
    run<10>("10 acc, 256 thr (1 wave/SIMD)", d, 256);
    run<20>("20 acc, 256 thr (1 wave/SIMD)", d, 256);
    run<20>("20 acc, 512 thr again", d, 512);
    return 0;
}
