// Microbenchmark: issue cost of the cross-lane register moves the FFT exchanges can be built from, against plain VALU
// instructions, at 1 / 2 / 4 waves per SIMD:  hipcc --offload-arch=gfx950 -O3 xlane_rate.hip -o bin/xlane_rate && bin/xlane_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = a0 * 2, b1 = a1 * 2, b2 = a2 * 2, b3 = a3 * 2;
    const unsigned long long mk = 0xF0F0F0F0F0F0F0F0ull;
    const int src = ((threadIdx.x ^ 5) & 63) << 2;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 8 independent v_add_f32
            asm volatile(REP8("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (MODE == 1) {   // v_mov_b32_dpp quad_perm
            asm volatile(REP8("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
        } else if (MODE == 2) {   // v_mov_b32_dpp row_ror:8
            asm volatile(REP8("v_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
        } else if (MODE == 3) {   // v_cndmask_b32_dpp (vcc set once)
            asm volatile("s_mov_b64 vcc, %8\n\t"
                         REP8("v_cndmask_b32_dpp %0, %4, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %1, %5, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                              "v_cndmask_b32_dpp %2, %6, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %3, %7, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(mk) : "vcc");
        } else if (MODE == 4) {   // v_cndmask_b32 without DPP
            asm volatile("s_mov_b64 vcc, %8\n\t"
                         REP8("v_cndmask_b32 %0, %4, %0, vcc\n\tv_cndmask_b32 %1, %5, %1, vcc\n\tv_cndmask_b32 %2, %6, %2, vcc\n\tv_cndmask_b32 %3, %7, %3, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(mk) : "vcc");
        } else if (MODE == 5) {   // v_permlane32_swap
            asm volatile(REP8("v_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        } else if (MODE == 6) {   // v_permlane16_swap
            asm volatile(REP8("v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        } else if (MODE == 7) {   // ds_bpermute_b32
            asm volatile(REP8("ds_bpermute_b32 %0, %8, %4\n\tds_bpermute_b32 %1, %8, %5\n\tds_bpermute_b32 %2, %8, %6\n\tds_bpermute_b32 %3, %8, %7\n\t")
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(src));
        } else if (MODE == 8) {   // v_pk_fma_f32
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {b0, b1}, p3 = {b2, b3};
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\tv_pk_fma_f32 %0, %0, %3, %2\n\tv_pk_fma_f32 %1, %1, %3, %2\n\t")
                         : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y;
        } else if (MODE == 9) {   // v_mov_b32_dpp row_shr:4 bound_ctrl
            asm volatile(REP8("v_mov_b32_dpp %0, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %1, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                              "v_mov_b32_dpp %2, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %7 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
        } else if (MODE == 10) {  // v_add_f32_dpp
            asm volatile(REP8("v_add_f32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                              "v_add_f32_dpp %2, %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
        } else if (MODE == 11) {  // v_cndmask_b32_e64 with an SGPR-pair mask (no vcc)
            asm volatile(REP8("v_cndmask_b32_e64 %0, %4, %0, %8\n\tv_cndmask_b32_e64 %1, %5, %1, %8\n\tv_cndmask_b32_e64 %2, %6, %2, %8\n\tv_cndmask_b32_e64 %3, %7, %3, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(mk));
        } else if (MODE == 12) {  // v_bfi_b32 (mask in a VGPR)
            asm volatile(REP8("v_bfi_b32 %0, %8, %4, %0\n\tv_bfi_b32 %1, %8, %5, %1\n\tv_bfi_b32 %2, %8, %6, %2\n\tv_bfi_b32 %3, %8, %7, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(src));
        } else if (MODE == 13) {  // v_cndmask_b32 vcc, destination not a source (no dependent chain)
            asm volatile("s_mov_b64 vcc, %8\n\t"
                         REP8("v_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %6, vcc\n\tv_cndmask_b32 %2, %6, %7, vcc\n\tv_cndmask_b32 %3, %7, %4, vcc\n\t")
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(mk) : "vcc");
        } else if (MODE == 14) {  // v_fmac_f32_dpp
            asm volatile(REP8("v_fmac_f32_dpp %0, %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f32_dpp %2, %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %7, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
        } else if (MODE == 15) {  // ds_swizzle_b32 (swap within quads)
            asm volatile(REP8("ds_swizzle_b32 %0, %4 offset:0x80b1\n\tds_swizzle_b32 %1, %5 offset:0x80b1\n\tds_swizzle_b32 %2, %6 offset:0x80b1\n\tds_swizzle_b32 %3, %7 offset:0x80b1\n\t")
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
        } else if (MODE == 16) {  // v_mov_b32 (plain)
            asm volatile(REP8("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
        } else if (MODE == 17) {  // v_pk_add_f32
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {b0, b1}, p3 = {b2, b3};
            asm volatile(REP8("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\tv_pk_add_f32 %0, %0, %3\n\tv_pk_add_f32 %1, %1, %2\n\t")
                         : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y;
        } else if (MODE == 18) {  // v_fma_f32
            asm volatile(REP8("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        } else if (MODE == 19) {  // v_cndmask_b32 vcc, vcc written by a VALU compare
            asm volatile("v_cmp_lt_f32 vcc, %4, %5\n\t"
                         REP8("v_cndmask_b32 %0, %4, %0, vcc\n\tv_cndmask_b32 %1, %5, %1, vcc\n\tv_cndmask_b32 %2, %6, %2, vcc\n\tv_cndmask_b32 %3, %7, %3, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");
        } else if (MODE == 20) {  // v_cndmask_b32 vcc, vcc written ONCE for 8 x 32 selects (loop inside the assembly)
            if (it % 8 == 0)
            asm volatile("s_mov_b64 vcc, %8\n\ts_mov_b32 s20, 8\n\t1:\n\t"
                         REP8("v_cndmask_b32 %0, %4, %0, vcc\n\tv_cndmask_b32 %1, %5, %1, vcc\n\tv_cndmask_b32 %2, %6, %2, vcc\n\tv_cndmask_b32 %3, %7, %3, vcc\n\t")
                         "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(mk) : "vcc", "scc", "s20");
        } else if (MODE == 21) {  // v_cndmask_b32_dpp, vcc written by a VALU compare
            asm volatile("v_cmp_lt_f32 vcc, %4, %5\n\ts_nop 4\n\t"
                         REP8("v_cndmask_b32_dpp %0, %4, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %1, %5, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                              "v_cndmask_b32_dpp %2, %6, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %3, %7, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");
        } else if (MODE == 22) {  // v_cndmask_b32_e64 with vcc named as the SGPR pair
            asm volatile("s_mov_b64 vcc, %8\n\t"
                         REP8("v_cndmask_b32_e64 %0, %4, %0, vcc\n\tv_cndmask_b32_e64 %1, %5, %1, vcc\n\tv_cndmask_b32_e64 %2, %6, %2, vcc\n\tv_cndmask_b32_e64 %3, %7, %3, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(mk) : "vcc");
        } else if (MODE == 23) {  // v_addc_co_u32 (reads and writes vcc)
            asm volatile("s_mov_b64 vcc, %8\n\t"
                         REP8("v_addc_co_u32 %0, vcc, %4, %0, vcc\n\tv_addc_co_u32 %1, vcc, %5, %1, vcc\n\tv_addc_co_u32 %2, vcc, %6, %2, vcc\n\tv_addc_co_u32 %3, vcc, %7, %3, vcc\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(mk) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
}

template <int MODE> void run(const char* name, float* d, int wps) {
    const int iters = 4000, blocks = 256 * wps;     // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);   // clock ramp
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)wps * iters * 32;      // instructions one SIMD issues
    printf("%-28s %d waves/SIMD %8.3f ms  %6.2f cycles per instruction and SIMD at 2.4 GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int wps : {1, 2, 4}) {
        run<0>("v_add_f32", d, wps); run<8>("v_pk_fma_f32", d, wps); run<4>("v_cndmask_b32", d, wps);
        run<1>("v_mov_b32_dpp quad_perm", d, wps); run<2>("v_mov_b32_dpp row_ror:8", d, wps); run<9>("v_mov_b32_dpp row_shr:4", d, wps);
        run<10>("v_add_f32_dpp quad_perm", d, wps);
        run<3>("v_cndmask_b32_dpp", d, wps); run<5>("v_permlane32_swap_b32", d, wps); run<6>("v_permlane16_swap_b32", d, wps);
        run<7>("ds_bpermute_b32", d, wps); run<15>("ds_swizzle_b32", d, wps);
        run<11>("v_cndmask_b32_e64 sgpr mask", d, wps); run<13>("v_cndmask_b32 no chain", d, wps); run<12>("v_bfi_b32", d, wps);
        run<14>("v_fmac_f32_dpp", d, wps); run<16>("v_mov_b32", d, wps); run<17>("v_pk_add_f32", d, wps); run<18>("v_fma_f32", d, wps);
        run<19>("v_cndmask_b32 vcc<-v_cmp", d, wps); run<20>("v_cndmask_b32 vcc set once", d, wps); run<21>("v_cndmask_b32_dpp vcc<-v_cmp", d, wps);
        run<22>("v_cndmask_b32_e64 vcc", d, wps); run<23>("v_addc_co_u32 vcc", d, wps);
    }
    return 0;
}
