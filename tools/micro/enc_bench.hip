// Stand-alone timing of encoder_ln_kernel<8, 4, true> at the C4 shape (256 windows, 513 frames, d_model 512) with parts
// of the kernel compiled out (-DKM_ENC_SKIP=bits, see km_encoder_dev.h): what each part costs when the others are gone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I koemorph_amd/csrc -I include -DKM_ENC_SKIP=0 tools/micro/enc_bench.hip -o enc_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "km_encoder_dev.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, NF = 513, T = 512, KP = 528, D = 512;
    std::vector<float> h((size_t)B * NF * 80);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3f + (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    std::vector<float> w((size_t)D * KP);
    for (size_t i = 0; i < w.size(); ++i) w[i] = ((float)((i * 40503u) & 0xfff) / 4096.f - 0.5f) * 0.05f;
    std::vector<unsigned> mx(B);
    for (int i = 0; i < B; ++i) { float one = 1.0f; mx[i] = *reinterpret_cast<unsigned*>(&one); }
    std::vector<float> ones(D, 1.f);
    float *dx, *dw, *db, *dg, *dbe, *dy; unsigned* dm;
    CK(hipMalloc(&dx, h.size() * 4)); CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&db, D * 4)); CK(hipMalloc(&dg, D * 4));
    CK(hipMalloc(&dbe, D * 4)); CK(hipMalloc(&dy, (size_t)B * 80 * D * 4)); CK(hipMalloc(&dm, B * 4));
    CK(hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, ones.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, ones.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbe, ones.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dm, mx.data(), B * 4, hipMemcpyHostToDevice));
    km::LogParams lp{};
    lp.log_mode = KM_LOG_DB_MAX; lp.amin = 1e-10f; lp.top_db = 80.f; lp.db_add = 80.f; lp.db_scale = 1.f / 80.f; lp.log_eps = 1e-8f;
    km::EncSrc src{dx, dm, NF, T, lp};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // 3000 untimed launches first (about 0.3 s): the chip reaches its steady-state clock only under sustained load
    const int warm = 3000, reps = 200;
    for (int it = 0; it < warm + reps; ++it) {
        if (it == warm) CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((km::encoder_ln_kernel<8, 4, true>), dim3(B), dim3(512), 0, 0, (const float*)nullptr, dw, db, dg, dbe, dy, KP, src);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, fl = 2.0 * 80 * KP * D * B;
    printf("KM_ENC_SKIP=%d B=%d: %.1f us per launch, %.1f TFLOP/s executed (%.3f of 157.3)\n", KM_ENC_SKIP, B, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3);
    return 0;
}
