// Stand-alone timing of scores_softmax_kernel<512> and attn_out_vr_kernel<512, HPW> at the C4 shape (256 windows,
// d_model 512, 8 or 16 heads) with parts compiled out (-DKM_SC_SKIP=bits / -DKM_VR_SKIP=bits, see km_attn_dev.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I koemorph_amd/csrc -I include tools/micro/attn_bench.hip -o attn_bench
//   ./attn_bench [heads=8] [windows=256]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "km_attn_dev.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static float* dev_random(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 16777216.f - 0.5f) * scale; }
    float* d = nullptr;
    if (hipMalloc(&d, n * 4) != hipSuccess) return nullptr;
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 8, B = argc > 2 ? atoi(argv[2]) : 256, D = 512, rows = H * 28, MT = (rows + 15) / 16;
    float* Y = dev_random((size_t)B * 80 * D, 2.f, 1);
    float* qk = dev_random((size_t)MT * 32 * 256, 0.2f, 2);
    float* S = dev_random((size_t)B * rows * 80, 0.02f, 3);
    float* wv = dev_random((size_t)D * D, 0.1f, 4);
    float* wf = dev_random((size_t)D * 256, 0.1f, 5);
    float* bf = dev_random(256, 0.1f, 6);
    float* w2 = dev_random(256, 0.1f, 7);
    float* b2 = dev_random(1, 0.1f, 8);
    float* ze = dev_random(B, 0.1f, 9);
    float* ws = dev_random(52, 1.f, 10);
    float* out = dev_random((size_t)B * 52, 1.f, 11);
    if (!Y || !qk || !S || !wv || !wf || !out) { printf("alloc failed\n"); return 1; }
    constexpr int lds = (2 * 16 * 81 * 4 + 32 * (512 + 8) + 8 * 32) * (int)sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&km::attn_out_vr_kernel<512, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&km::attn_out_vr_kernel<512, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // `warm` untimed launches first (about 0.3 s): the chip reaches its steady-state clock only under sustained load
    const int warm = 3000, reps = 200;
    float ms;
    for (int it = 0; it < warm + reps; ++it) {
        if (it == warm) CK(hipEventRecord(e0, 0));
        if (rows <= 256) hipLaunchKernelGGL((km::scores_softmax_kernel<512, 2>), dim3(B), dim3(512), 0, 0, Y, qk, S, rows);
        else hipLaunchKernelGGL((km::scores_softmax_kernel<512, 4>), dim3(B), dim3(512), 0, 0, Y, qk, S, rows);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    {
        const double us = ms * 1e3 / reps, fl = 2.0 * MT * 16 * 80 * D * B;
        printf("scores  KM_SC_SKIP=%d H=%d B=%d: %.1f us, %.1f TFLOP/s executed (%.3f of 157.3)\n", KM_SC_SKIP, H, B, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3);
    }
    for (int it = 0; it < warm + reps; ++it) {
        if (it == warm) CK(hipEventRecord(e0, 0));
        if (H == 8) hipLaunchKernelGGL((km::attn_out_vr_kernel<512, 1>), dim3(B), dim3(512), lds, 0, S, Y, wv, wf, bf, w2, b2, ze, ws, out, (float*)nullptr);
        else hipLaunchKernelGGL((km::attn_out_vr_kernel<512, 2>), dim3(B), dim3(512), lds, 0, S, Y, wv, wf, bf, w2, b2, ze, ws, out, (float*)nullptr);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    {
        const double us = ms * 1e3 / reps, fl = 3232.0 * 8 * 2048 * B;       // MFMAs per wave x waves x FLOP per MFMA
        printf("attn_vr KM_VR_SKIP=%d H=%d B=%d: %.1f us, %.1f TFLOP/s executed (%.3f of 157.3)\n", KM_VR_SKIP, H, B, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3);
    }
    return 0;
}
