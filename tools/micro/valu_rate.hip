// Microbenchmark: issue rate of scalar v_fma_f32 / v_add_f32 vs packed v_pk_fma_f32 / v_pk_add_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[16]; v2f p[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    const v2f ps = v2f{s, s * 1.0001f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], s, 0.5f);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], ps, ps);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = a[i] + s;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = p[i] + ps;
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, float* d) {
    const int iters = 20000, blocks = 256 * 4;   // 4 blocks/CU = 4 waves/SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * iters * 16;          // 16 scalar results per iteration either way
    printf("%-14s %8.3f ms  %7.2f T lane-ops/s  (%s)\n", name, ms, lane_ops / ms / 1e9,
           MODE < 2 ? "x2 = TFLOP/s for fma" : "adds");
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_add_f32", d); run<3>("v_pk_add_f32", d);
    return 0;
}
