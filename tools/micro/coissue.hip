// Microbenchmark: what a VALU / packed-fp32 / LDS wave costs an fp32-MFMA wave that shares its SIMD (gfx950), and
// what it gets.  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) issue v_mfma_f32_16x16x4_f32 back to back
// on 16 independent accumulators (the fused core's shape); waves 4-7 (their SIMD partners) run a filler role until the
// MFMA waves are done.  Reported: cycles per MFMA with each partner, and the partner's instructions per cycle, against
// the same roles running alone.  This is the experiment behind DESIGN.md section 7.1 (can the FFT front end hide under
// the core's matrix work if both are resident on a CU?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum Role { IDLE = 0, FMA = 1, PKFMA = 2, LDSRD = 3, PKADD = 4, MIX = 5 };

struct Res { unsigned long long a_cycles, a_count, b_cycles, b_count; };

template <bool A_MFMA, int BPRIO>
__global__ __launch_bounds__(512) void k(Res* res, float* sink, int a_iters, int role, int b_fixed_iters) {
    __shared__ volatile int done[4];
    __shared__ float lbuf[8 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 4) done[tid] = 0;
    for (int i = tid; i < 8 * 1024; i += 512) lbuf[i] = i * 0.001f;
    __syncthreads();
    if (wave < 4) {
        // ---- role A: MFMA (or nothing) ----
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        float a = lane * 0.001f, b = 1.0001f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (A_MFMA) {
            for (int it = 0; it < a_iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        }
        float r = 0;
        for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        asm volatile("" ::"v"(r));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) done[wave] = r == 0.12345f ? 2 : 1;      // depends on the accumulators: cannot be hoisted above the loop
        if (lane == 0 && blockIdx.x == 0 && wave == 0) { res->a_cycles = t1 - t0; res->a_count = (unsigned long long)a_iters * 16; }
        sink[blockIdx.x * 512 + tid] = r;
    } else {
        // ---- role B: filler, until the partner (wave - 4) is done (or a fixed count when A is idle) ----
        float x[16];
        v2f y[16];
        for (int i = 0; i < 16; ++i) { x[i] = lane + i; y[i] = v2f{(float)lane, (float)i}; }
        const float c1 = 0.999f, c2 = 0.001f;
        const v2f pc1 = {0.999f, 0.999f}, pc2 = {0.001f, 0.002f};
        const float* lp = lbuf + (wave * 1024 + lane * 2);
        unsigned long long n = 0;
        __builtin_amdgcn_s_setprio(BPRIO);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0;; ++it) {
            if (it >= b_fixed_iters) break;
            if (role == FMA) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
                n += 64;
            } else if (role == PKFMA) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(pc1), "v"(pc2));
                n += 64;
            } else if (role == PKADD) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(pc2));
                n += 64;
            } else if (role == LDSRD) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(y[i]) : "v"((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lp), "i"(512 * (i & 7)));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                n += 64;
            } else if (role == MIX) {
                // the front end's blend: 5 packed VALU per LDS access
#pragma unroll
                for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(y[12 + i]) : "v"((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lp), "i"(512 * i));
#pragma unroll
                    for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(pc1), "v"(pc2));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[0]) : "v"(y[12]));
                }
                n += 64;
            } else {
                break;
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float r = 0;
        for (int i = 0; i < 16; ++i) r += x[i] + y[i].x + y[i].y;
        if (lane == 0 && blockIdx.x == 0 && wave == 4) { res->b_cycles = t1 - t0; res->b_count = n; }
        sink[blockIdx.x * 512 + tid] = r;
    }
}

int main() {
    Res* d; float* sink; Res h;
    hipMalloc(&d, sizeof(Res)); hipMalloc(&sink, 256 * 512 * 4);
    const char* names[] = {"idle", "v_fma_f32", "v_pk_fma_f32", "ds_read_b64", "v_pk_add_f32", "fft-like mix (12 pk : 3 lds : 1 pk)"};
    const int a_iters = 4000;
    printf("%-38s %14s %16s %16s\n", "partner role (waves 4-7)", "cyc/MFMA", "partner instr/cyc", "partner alone");
    for (int prio = 0; prio < 2; ++prio)
    for (int role = 0; role < 6; ++role) {
        if (role == 0) printf("---- partner waves at s_setprio %d (MFMA waves at 0) ----\n", prio ? 3 : 0);
        double alone = 0;
        if (role != IDLE) {
            hipMemset(d, 0, sizeof(Res));
            hipLaunchKernelGGL((k<false, 0>), dim3(256), dim3(512), 0, 0, d, sink, 0, role, 3000);
            hipDeviceSynchronize(); hipMemcpy(&h, d, sizeof(Res), hipMemcpyDeviceToHost);
            alone = (double)h.b_count / (double)h.b_cycles;
        }
        // the partner runs a fixed count sized to outlast the MFMA waves by ~1.5x when alone; its rate WHILE the MFMA waves
        // run = (its instructions - alone_rate x its tail after they finished) / their duration
        const double a_alone_cycles = 32.0 * 16 * a_iters;
        const int b_iters = role == IDLE ? 0 : (int)(1.5 * a_alone_cycles * alone / 64.0);
        hipMemset(d, 0, sizeof(Res));
        for (int rep = 0; rep < 2; ++rep) {
            if (prio) hipLaunchKernelGGL((k<true, 3>), dim3(256), dim3(512), 0, 0, d, sink, a_iters, role, b_iters);
            else hipLaunchKernelGGL((k<true, 0>), dim3(256), dim3(512), 0, 0, d, sink, a_iters, role, b_iters);
        }
        hipDeviceSynchronize(); hipMemcpy(&h, d, sizeof(Res), hipMemcpyDeviceToHost);
        double co = 0.0;
        if (h.b_cycles) {
            const double tail = (double)h.b_cycles > (double)h.a_cycles ? (double)h.b_cycles - (double)h.a_cycles : 0.0;
            co = ((double)h.b_count - alone * tail) / ((double)h.b_cycles - tail);
        }
        printf("%-38s %14.2f %16.4f %16.4f   (partner ran %.0f cycles, MFMA waves %.0f)\n", names[role],
               (double)h.a_cycles / (double)h.a_count, co, alone, (double)h.b_cycles, (double)h.a_cycles);
    }
    return 0;
}
