// The BM x 64 tile of km_gemm_dev.h with its operands staged by LDS-DMA (buffer_load_dwordx4 ... lds) -- the fast path of the
// training program's products (km_trainp.hip), round 4.
//
// Why: tools/micro/tile_bench.hip on the register-staged tile at the 8-window shapes: a k-step of 32 took 0.64 us = 1500
// cycles for 512 cycles of MFMA work.  The step was ~200 instructions per wave -- per slot a buffer load, four selects for the
// K / row tails, address arithmetic and one to four ds_writes, an integer division for the contraction batch -- issued by ONE
// wave per SIMD (8+ cycles per instruction), and the compiler's s_waitcnt vmcnt(0) in front of every commit turned the
// D-deep prefetch into a one-deep one.  Here a k-step is, per wave: BM / 32 + 2 DMA instructions (no VGPR round trip, no
// ds_write, no select), one counted s_waitcnt, one s_barrier, 6 - 8 ds_read_b128 (or their b32 form) and the 16 - 32 MFMAs.
//
//   * Staging.  256 threads x 16 bytes = 4 KB per DMA instruction; the LDS destination of lane l is M0 base + 16 l
//     (tools/micro/dma_probe.hip: lane-linear, lanes beyond the descriptor's range write ZEROS, the immediate offset moves
//     source AND destination, the scalar offset only the source), so the LDS image is filled in slot order and what a slot
//     holds is chosen through its SOURCE address:
//       mode 0, k-contiguous operand (activations, nn.Linear weights): image [row][8 chunks of 4 k], chunk c of row r stored
//         at chunk position c ^ ((r >> 1) & 7).  Eight consecutive lanes read one whole 128-byte line of a row.  A fragment
//         (4 consecutive k of 16 consecutive rows) is one ds_read_b128 per lane, and the 16 lanes of a row tile hit 16
//         different 16-byte bank groups;
//       mode 1, row-contiguous operand (the transposed operands of the input- and weight-gradient products): image
//         [32 k][ROWS], a slot = four consecutive rows of one k, row group g of k stored at g ^ (4 ((k >> 2) & 1)) (rows
//         16 .. 31 swapped with 0 .. 15 in every other group of four k).  A fragment is four ds_read_b32 at compile-time
//         offsets; lane groups lg and lg + 1 of a 32-lane half then read different halves of the 32 banks.
//     Both operands use the k partition of km_gemm_dev.h (lane group lg supplies k = 16 kb + 4 lg + s to MFMA s).
//   * Pipeline.  A ring of NS stage buffers (NS = 8 when a product's workgroups are alone on their CUs: K = 256 is
//     requested whole at entry; NS = 4 when several workgroups share a CU), NS - 1 stages in flight: wait until stage `it`
//     has landed (s_waitcnt vmcnt((NS - 2) L), L loads per thread and stage -- stages past the end are issued out of range so
//     that the count stays uniform), s_barrier (everyone's part of stage `it` has landed, everyone is done reading stage
//     it - 1), request stage it + NS - 1 into the buffer just freed, then fragments + MFMAs of stage `it`.  Raw s_barrier
//     and inline-asm waits: __syncthreads() would drain the DMA queue (cdna_hip_programming.md, "Pipelining across barriers").
//   * Eligibility (gemm_dma_ok): K a multiple of 32, no contraction batch, 16-byte aligned slots, operand modes (0,0) NT,
//     (0,1) NN or (1,1) TN.  Rows outside the matrix are zero-filled by the descriptor's range check (they only feed output
//     elements that are never stored).  Everything else runs on the register-staged tile.
// Arithmetic: the same MFMAs in the same order as gemm_tile_dev -- results are bit-identical to the register-staged tile.
// Included inside namespace km after km_gemm_dev.h.
#pragma once

#include "km_gemm_dev.h"

namespace gdma {
constexpr int BN = 64, BK = 32;
constexpr int stage_floats(int BM) { return (BM + BN) * BK; }
constexpr int lds_floats(int BM, int NS) { return NS * stage_floats(BM); }
}

typedef __attribute__((address_space(3))) void* km_lds_ptr;

// host + device: may this product run on the DMA tile?  ma / mb: operand modes (0 k-contiguous, 1 row-contiguous)
__host__ __device__ inline bool gemm_dma_ok(const GemmArgs& g, int* ma_out, int* mb_out) {
    if (g.kb_count > 1 || g.K < 32 || (g.K & 31)) return false;
    const int ma = g.a_cs == 1 ? 0 : (g.a_rs == 1 ? 1 : -1), mb = g.b_rs == 1 ? 0 : (g.b_cs == 1 ? 1 : -1);
    if (ma < 0 || mb < 0 || (ma == 1 && mb == 0)) return false;
    const int64_t a_step = ma == 0 ? g.a_rs : g.a_cs, b_step = mb == 0 ? g.b_cs : g.b_rs;     // stride between 16-byte slots' rows / k rows
    if ((a_step & 3) || (b_step & 3)) return false;
    if ((reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.B) & 15)) return false;
    if ((g.a_bs1 & 3) || (g.a_bs2 & 3) || (g.b_bs1 & 3) || (g.b_bs2 & 3)) return false;
    if (ma_out) *ma_out = ma;
    if (mb_out) *mb_out = mb;
    return true;
}

template <int ROWS, int MODE>
struct DmaOperand {
    static constexpr int NI = ROWS / 32;                 // DMA instructions per thread and stage
    static constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[NI];                                   // byte offset of the thread's slot in k-tile 0 (OOB: row outside the matrix)
    unsigned kstep;                                      // bytes from one k-tile to the next

    // rs: stride between rows, ks: stride between k (one of them is 1: the contiguous direction)
    __device__ __forceinline__ void init(const float* b, int64_t rs, int64_t ks, int row0, int nrows, int64_t extent_floats, int tid) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, (unsigned)(extent_floats * 4), 0x00020000);
        kstep = (unsigned)(gdma::BK * ks * 4);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s = i * 256 + tid;
            if constexpr (MODE == 0) {
                const int row = s >> 3, cp = s & 7, kg = cp ^ ((row >> 1) & 7), g = row0 + row;
                voff[i] = g < nrows ? (unsigned)(((int64_t)g * rs + 4 * kg) * 4) : OOB;
            } else {
                constexpr int per = ROWS / 4;
                const int k = s / per, gp = s - k * per, grp = gp ^ (4 * ((k >> 2) & 1)), g0 = row0 + 4 * grp;
                voff[i] = g0 < nrows ? (unsigned)(((int64_t)k * ks + g0) * 4) : OOB;
            }
        }
    }
    // request k-tile nk into the stage image `img` (wave-uniform pointer); dead: a stage past the end (everything out of range)
    __device__ __forceinline__ void issue(float* img, int wave, int nk, bool dead) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned o = dead ? OOB : voff[i] + (unsigned)nk * kstep;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (km_lds_ptr)(img + (i * 256 + 64 * wave) * 4), 16, o, 0, 0, 0);
        }
    }
};

// per-lane read position of one 16-row tile of an operand image (rows rowbase .. rowbase + 15, rowbase a multiple of 16)
template <int ROWS, int MODE>
struct DmaFragment {
    int base, sw;
    __device__ __forceinline__ void init(int rowbase, int lg, int lj) {
        const int R = rowbase + lj;
        if constexpr (MODE == 0) { base = R * 32; sw = (R >> 1) & 7; }
        else { base = 4 * lg * ROWS + (R ^ (16 * (lg & 1))); sw = 0; }
    }
    // the lane's operand values of the four MFMAs of k block kb (k = 16 kb + 4 lg + s, s = 0 .. 3)
    __device__ __forceinline__ f32x4 read(const float* img, int kb, int lg) const {
        if constexpr (MODE == 0) {
            return *reinterpret_cast<const f32x4*>(img + base + (((4 * kb + lg) ^ sw) << 2));
        } else {
            const float* p = img + base + 16 * kb * ROWS;
            return f32x4{p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]};
        }
    }
};

template <int N> __device__ __forceinline__ void km_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// tile (bx, by) of output batch bz; smem = gdma::lds_floats(BM, ns) floats, 16-byte aligned; ns = 4 or 8 ring stages
template <int BM, int MA, int MB>
__device__ __forceinline__ void gemm_tile_dma_dev(const GemmArgs& g, int bx, int by, int bz, float* smem, int ns) {
    using namespace gdma;
    constexpr int MT = BM / 32, SF = stage_floats(BM), L = BM / 32 + 2;      // L: DMA instructions per thread and stage
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: it enters LDS-DMA destinations
    const int lg = lane >> 4, lj = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int z1 = bz / g.batch2, z2 = bz - z1 * g.batch2;
    const int m0 = by * BM, n0 = bx * BN;
    DmaOperand<BM, MA> ta;
    DmaOperand<BN, MB> tb;
    ta.init(g.A + z1 * g.a_bs1 + z2 * g.a_bs2, g.a_rs, g.a_cs, m0, g.M, gemm_operand_extent(g.M, g.a_rs, g.K, g.a_cs, 1, 0), tid);
    tb.init(g.B + z1 * g.b_bs1 + z2 * g.b_bs2, g.b_cs, g.b_rs, n0, g.N, gemm_operand_extent(g.N, g.b_cs, g.K, g.b_rs, 1, 0), tid);
    float* C = g.C + z1 * g.c_bs1 + z2 * g.c_bs2;
    DmaFragment<BM, MA> fa[MT];
    DmaFragment<BN, MB> fb[2];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i].init(16 * MT * wm + 16 * i, lg, lj);
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) fb[jn].init(32 * wn + 16 * jn, lg, lj);
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = f32x4{0, 0, 0, 0}; acc[i][1] = f32x4{0, 0, 0, 0}; }
    const int kt = g.K / BK, mask = ns - 1;
    KM_TILE_STAMP(1);
    auto issue = [&](int j) {
        float* img = smem + (j & mask) * SF;
        ta.issue(img, wave, j, j >= kt);
        tb.issue(img + BM * BK, wave, j, j >= kt);
    };
    for (int j = 0; j < ns - 1; ++j) issue(j);
    for (int it = 0; it < kt; ++it) {
        if (ns == 8) km_wait_vmcnt<6 * L>(); else km_wait_vmcnt<2 * L>();       // stage `it` of this thread has landed
        __builtin_amdgcn_s_barrier();                                          // ... and everyone's; stage it - 1 is free
        if (it == 0) KM_TILE_STAMP(2);
        issue(it + ns - 1);
        const float* As = smem + (it & mask) * SF;
        const float* Bs = As + BM * BK;
#pragma unroll
        for (int kb = 0; kb < BK / 16; ++kb) {
            f32x4 af[MT], bf[2];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = fa[i].read(As, kb, lg);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) bf[jn] = fb[jn].read(Bs, kb, lg);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    acc[i][0] = KM_MFMA(af[i][q], bf[0][q], acc[i][0]);
                    acc[i][1] = KM_MFMA(af[i][q], bf[1][q], acc[i][1]);
                }
        }
    }
    km_wait_vmcnt<0>();           // the out-of-range stages behind the last one are still writing zeros into the ring
    KM_TILE_STAMP(3);
    gemm_tile_epilogue<MT>(g, C, acc, m0 + 16 * MT * wm, n0 + 32 * wn, lg, lj);
    __syncthreads();              // the caller may reuse smem (another tile of the same workgroup)
}
