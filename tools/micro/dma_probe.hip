// Semantics of buffer_load_dwordx4 ... lds on gfx950 (the staging primitive of km_gemm_dma_dev.h): where does lane l's 16 bytes
// land, what happens to lanes whose offset lies beyond the descriptor's range, how do the immediate and the scalar offset
// enter.  hipcc --offload-arch=gfx950 -O3 tools/micro/dma_probe.hip -o tools/micro/bin/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* a, float* out, int nfloats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 2048; i += 64) smem[i] = 7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, (unsigned)nfloats * 4, 0x00020000);
    const int lane = threadIdx.x & 63;
    // instruction 1: lane l reads 16 bytes at byte offset 16 (l ^ 5); lanes 60..63 are sent out of range
    const unsigned off = lane >= 60 ? 0x80000000u : (unsigned)((lane ^ 5) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem), 16, off, 0, 0, 0);
    // instruction 2: into smem + 1024 floats... destination base via the pointer, source shifted by soffset 2048 bytes and imm 32 bytes
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem + 1024), 16, lane * 16, 2048, 32, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = smem[i];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *a, *out;
    hipMalloc(&a, n * 4); hipMalloc(&out, 2048 * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, a, out, n);
    std::vector<float> o(2048);
    hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0, bad3 = 0;
    for (int l = 0; l < 60; ++l) for (int c = 0; c < 4; ++c) bad1 += o[l * 4 + c] != (float)((l ^ 5) * 4 + c);
    printf("instruction 1: lane l -> LDS bytes [16 l, 16 l + 16): %s (%d wrong)\n", bad1 ? "NO" : "yes", bad1);
    printf("  out-of-range lanes 60..63 left in LDS:");
    for (int i = 240; i < 256; ++i) printf(" %g", o[i]);
    printf("   (7 = untouched, 0 = zero fill)\n");
    for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) bad2 += o[1024 + l * 4 + c] != (float)(l * 4 + c + 512 + 8);
    printf("instruction 2: source = voffset + soffset + imm, destination = pointer + 16 l: %s (%d wrong; first values %g %g)\n", bad2 ? "NO" : "yes", bad2, o[1024], o[1025]);
    for (int l = 0; l < 62; ++l) for (int c = 0; c < 4; ++c) bad3 += o[1024 + 8 + l * 4 + c] != (float)(l * 4 + c + 512 + 8);
    printf("   the immediate offset moves the DESTINATION too (pointer + imm + 16 l): %s (%d wrong)\n", bad3 ? "NO" : "yes", bad3);
    return 0;
}
