// Shader clock under a latency-bound load: one workgroup per CU runs a dependent v_fma chain of known length; cycles from
// s_memtime, time from s_memrealtime (100 MHz).  hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o tools/micro/bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, unsigned long long* t, int n) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x * 1e-9f;
    for (int i = 0; i < n; ++i) x = fmaf(x, 1.000001f, 1e-9f);
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
    float* out; unsigned long long* t; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&t, 256 * 16);
    for (int n : {2000, 20000, 200000}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, t, n);
            hipDeviceSynchronize();
            unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("n %d: %llu s_memtime ticks in %.2f us: %.0f ticks/us; %.2f ticks per dependent fma\n", n, h[0], h[1] * 0.01, h[0] / (h[1] * 0.01), (double)h[0] / n);
        }
    }
    return 0;
}
