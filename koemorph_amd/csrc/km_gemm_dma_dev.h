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
//   * Pipeline.  A ring of NS stage buffers (gdma::ring_stages), NS - 2 stages in flight under a step's MFMAs.  Step `it`:
//     s_waitcnt vmcnt((NS - 3) L) -- stage it + 1 of this thread has landed (L loads per thread and stage; stages past the end are
//     issued out of range so that the count stays uniform) --, s_barrier (everyone's part of it has landed; everyone's
//     fragment reads of stage it - 1 are complete), request stage it + NS - 1 into the buffer of stage it - 1, fetch the
//     fragments of stage it + 1 into the second register set, then the MFMAs of stage `it` on the set fetched a step earlier:
//     LDS latency and the DMA issue sit under the matrix pipe.  Raw s_barrier and inline-asm waits: __syncthreads() would
//     drain the DMA queue (cdna_hip_programming.md, "Pipelining across barriers").
//   * Eligibility (gemm_dma_ok): K a multiple of 32, no contraction batch, 16-byte aligned slots, operand modes (0,0) NT,
//     (0,1) NN or (1,1) TN.  Rows outside the matrix are zero-filled by the descriptor's range check (they only feed output
//     elements that are never stored).  Everything else runs on the register-staged tile.
// Arithmetic: the same MFMAs in the same order as gemm_tile_dev -- results are bit-identical to the register-staged tile.
// Included inside namespace km after km_gemm_dev.h.
#pragma once

#include "km_device.h"
#include "km_gemm_dev.h"

namespace gdma {
constexpr int BN = 64, BK = 32;
constexpr int stage_floats(int BM) { return (BM + BN) * BK; }
constexpr int lds_floats(int BM, int NS) { return NS * stage_floats(BM); }
// ring stages: 8 for the 32-row tiles of a phase whose workgroups are alone on their CUs (the 8-window step: the whole K = 256 is
// requested at entry, 96 KB of LDS), else 4 (48 / 64 KB: several workgroups per CU hide each other's prologue and epilogue)
#ifndef KM_DMA_NS64
#define KM_DMA_NS64 3
#endif
#ifndef KM_DMA_NS32
#define KM_DMA_NS32 4
#endif
constexpr int ring_stages(int BM, bool alone) { return BM == 32 ? (alone ? 8 : KM_DMA_NS32) : KM_DMA_NS64; }
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
    // request k-tile nk into the stage image `img` (wave-uniform pointer); dead = 0x80000000 for a stage past the end (the OR
    // sends every lane out of range: branch-free, the DMA count per stage stays uniform), else 0
    __device__ __forceinline__ void issue(float* img, int wave, int nk, unsigned dead) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned o = (voff[i] + (unsigned)nk * kstep) | dead;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (km_lds_ptr)(img + (i * 256 + 64 * wave) * 4), 16, o, 0, 0, 0);
        }
    }
};

// per-lane read positions of the NT 16-row tiles of an operand image that a wave owns (rows rowbase + 16 t + lj).  The swizzles
// depend on lj and lg only, so the tiles differ by compile-time offsets in mode 0 and by one position per tile in mode 1.
template <int ROWS, int MODE, int NT>
struct DmaFragments {
    int pos[MODE == 0 ? 2 : NT];
    __device__ __forceinline__ void init(int rowbase, int lg, int lj) {
        if constexpr (MODE == 0) {
            const int sw = (lj >> 1) & 7;                        // (R >> 1) & 7 with R = rowbase + 16 t + lj, rowbase a multiple of 16
            pos[0] = (rowbase + lj) * 32 + (((lg) ^ sw) << 2);   // k block 0: chunk lg
            pos[1] = (rowbase + lj) * 32 + (((4 + lg) ^ sw) << 2);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) pos[t] = 4 * lg * ROWS + ((rowbase + 16 * t + lj) ^ (16 * (lg & 1)));
        }
    }
    // the lane's operand values of the four MFMAs of k block kb (k = 16 kb + 4 lg + s, s = 0 .. 3) for tile t
    __device__ __forceinline__ f32x4 read(const float* img, int t, int kb) const {
        if constexpr (MODE == 0) {
            return *reinterpret_cast<const f32x4*>(img + pos[kb] + t * 16 * 32);
        } else {
            const float* p = img + pos[t] + 16 * kb * ROWS;
            return f32x4{p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]};
        }
    }
};

template <int N> __device__ __forceinline__ void km_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Epilogue of the DMA tile: C through a buffer descriptor -- a row outside the matrix lies beyond its range (the store is dropped,
// a load returns 0), a column outside sets the offset's top bit -- so no per-element branch; row offsets advance by adds.
// alpha, bias (per column / per row), beta C, activation, dropout as in gemm_tile_epilogue.  The operands of the epilogue that do
// not depend on the product -- bias values, dropout bytes -- are requested BEFORE the k loop (prefetch): behind it they were one
// more memory round trip on the critical path of every forward phase (~0.8 us, 2 us with the dropout bytes).
template <int MT>
struct DmaEpilogue {
    float bias_a, bias_b, bias_m[MT][4];
    unsigned char keep_a[MT][4], keep_b[MT][4];
    __device__ __forceinline__ void prefetch(const GemmArgs& g, int mw, int nw, int lg, int lj) {
        const int n_a = nw + lj, n_b = nw + 16 + lj;
        const int nca = n_a < g.N ? n_a : g.N - 1, ncb = n_b < g.N ? n_b : g.N - 1;
        bias_a = 0.f; bias_b = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { bias_m[i][r] = 0.f; keep_a[i][r] = 1; keep_b[i][r] = 1; }
        if (g.bias_mode == 1) { bias_a = g.bias[nca]; bias_b = g.bias[ncb]; }
        else if (g.bias_mode == 2) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int m = mw + 16 * i + 4 * lg + r; bias_m[i][r] = g.bias[m < g.M ? m : g.M - 1]; }
        }
        if (g.drop) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mw + 16 * i + 4 * lg + r;
                    const unsigned char* row = g.drop + gemm_drop_row(g.drop_map, m < g.M ? m : g.M - 1) * g.N;
                    keep_a[i][r] = row[nca]; keep_b[i][r] = row[ncb];
                }
        }
    }
    // stats != null: per output row and per 32 columns (this wave's), the mean of the stored values and their sum of squared
    // deviations from it go to stats[(row_base + row) * npart + nw / 32] (float2) for the rows' reader (km_device.h: LnXform); N % 32 == 0
    __device__ __forceinline__ void finish(const GemmArgs& g, float* C, const f32x4 (&acc)[MT][2], int mw, int nw, int lg, int lj,
                                           float* stats = nullptr, int64_t row_base = 0) const {
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(C, 0, (unsigned)(((int64_t)(g.M - 1) * g.c_rs + g.N) * 4), 0x00020000);
        const int n_a = nw + lj, n_b = nw + 16 + lj;
        const unsigned oob_a = n_a < g.N ? 0u : 0x80000000u, oob_b = n_b < g.N ? 0u : 0x80000000u;
        const unsigned rstep = (unsigned)g.c_rs * 4u;
        const unsigned row0 = (unsigned)(mw + 4 * lg) * rstep + (unsigned)n_a * 4u;           // element (i = 0, r = 0, jn = 0)
        float va[MT][4], vb[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                va[i][r] = g.alpha * acc[i][0][r] + (bias_a + bias_m[i][r]);
                vb[i][r] = g.alpha * acc[i][1][r] + (bias_b + bias_m[i][r]);
            }
        if (g.beta != 0.f) {
            float oa[MT][4], ob[MT][4];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned o = row0 + (unsigned)(16 * i + r) * rstep;
                    oa[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rc, o | oob_a, 0, 0));
                    ob[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rc, (o + 64u) | oob_b, 0, 0));
                }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { va[i][r] += g.beta * oa[i][r]; vb[i][r] += g.beta * ob[i][r]; }
        }
        if (g.relu != 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { va[i][r] = gemm_act(va[i][r], g.relu); vb[i][r] = gemm_act(vb[i][r], g.relu); }
        }
        if (g.drop) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    va[i][r] = keep_a[i][r] ? va[i][r] * g.drop_scale : 0.f;
                    vb[i][r] = keep_b[i][r] ? vb[i][r] * g.drop_scale : 0.f;
                }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned o = row0 + (unsigned)(16 * i + r) * rstep;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(va[i][r]), rc, o | oob_a, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vb[i][r]), rc, (o + 64u) | oob_b, 0, 0);
            }
        if (stats) {        // workgroup-uniform
            const int npart = g.N >> 5;
            float2* st = reinterpret_cast<float2*>(stats) + row_base * npart + (nw >> 5);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float mean = row16_sum(va[i][r] + vb[i][r]) * (1.0f / 32.0f);      // the 16 lanes of a DPP row hold one output row
                    const float da = va[i][r] - mean, db = vb[i][r] - mean;
                    const float m2 = row16_sum(da * da + db * db);
                    const int row = mw + 16 * i + 4 * lg + r;
                    if (lj == 0 && row < g.M) st[(int64_t)row * npart] = make_float2(mean, m2);
                }
        }
    }
};

// tile (bx, by) of output batch bz; smem = gdma::lds_floats(BM, NS) floats, 16-byte aligned; NS = 4 or 8 ring stages
// XFA (xf != null): operand A holds 10 log10(power) rows and every fragment is taken through db_finish() with the reference of
// output batch entry z1 (km_device.h: DbXform) on its way to the MFMAs
// XFA == 2 (ln != null): operand A holds rows whose LayerNorm the product reads: the row statistics are pooled from the producer's
// parts (ln->stats, DmaEpilogue), gamma / beta sit in LDS behind the ring (2 K floats), and every fragment goes through ln_apply().
// stats != null: this product leaves such parts for ITS output rows (row r of batch entry z1 = row z1 M + r).
template <int BM, int NS, int MA, int MB, int XFA = 0>
__device__ __forceinline__ void gemm_tile_dma_dev(const GemmArgs& g, int bx, int by, int bz, float* smem, const DbXform* xf = nullptr,
                                                  const LnXform* ln = nullptr, float* stats = nullptr) {
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
    DmaFragments<BM, MA, MT> fa;
    DmaFragments<BN, MB, 2> fb;
    fa.init(16 * MT * wm, lg, lj);
    fb.init(32 * wn, lg, lj);
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = f32x4{0, 0, 0, 0}; acc[i][1] = f32x4{0, 0, 0, 0}; }
    const int kt = g.K / BK;
    float xscale = 0.f, xc1 = 0.f;      // XFA == 1: set behind the prologue's DMA requests
    float lrs[MT], lnm[MT];             // XFA == 2: rstd and -mean rstd of this lane's fragment rows
    float* gb = smem + NS * SF;         //           gamma [K], beta [K]
    KM_TILE_STAMP(1);
    auto issue = [&](int j) {
        float* img = smem + (j % NS) * SF;
        const unsigned dead = j >= kt ? 0x80000000u : 0u;
        ta.issue(img, wave, j, dead);
        tb.issue(img + BM * BK, wave, j, dead);
    };
    // fragments of one stage: [tile][k block]
    auto fetch = [&](int j, f32x4 (&af)[MT][2], f32x4 (&bf)[2][2]) {
        const float* As = smem + (j % NS) * SF;
        const float* Bs = As + BM * BK;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i][kb] = fa.read(As, i, kb);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) bf[jn][kb] = fb.read(Bs, jn, kb);
        }
    };
    auto mfmas = [&](int it, const f32x4 (&af)[MT][2], const f32x4 (&bf)[2][2]) {
        (void)it;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x4 at[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                at[i] = af[i][kb];
                if constexpr (XFA == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) at[i][q] = db_finish_fast(at[i][q], xscale, xc1);
                }
            }
            if constexpr (XFA == 2) {
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gb + 32 * it + 16 * kb + 4 * lg);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(gb + g.K + 32 * it + 16 * kb + 4 * lg);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) at[i][q] = ln_apply(at[i][q], lrs[i], lnm[i], gm[q], bt[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    acc[i][0] = KM_MFMA(at[i][q], bf[0][kb][q], acc[i][0]);
                    acc[i][1] = KM_MFMA(at[i][q], bf[1][kb][q], acc[i][1]);
                }
        }
    };
    // one step: stage it + 1 has landed everywhere, the buffer of stage it - 1 is free; request stage it + NS - 1 into it, fetch
    // the fragments of stage it + 1 (they arrive under the MFMAs of stage it, whose fragments were fetched a step earlier)
    auto step = [&](int it, const f32x4 (&af)[MT][2], const f32x4 (&bf)[2][2], f32x4 (&af_n)[MT][2], f32x4 (&bf_n)[2][2]) {
        km_wait_vmcnt<(NS - 3) * L>();
        __builtin_amdgcn_s_barrier();
        issue(it + NS - 1);
        fetch(it + 1, af_n, bf_n);
        mfmas(it, af, bf);
        // (measured and dropped: sched_group_barrier patterns pinning one MFMA, one DMA request, one fragment read in turn -- the
        // waits and the barrier cut the step into scheduling regions of their own: 13.4 us against 13.3 at K = 1024)
    };
    // the epilogue's own operands first: plain loads, OLDER than every DMA request, so the counted waits below (which
    // leave the youngest requests in flight) cover them
    DmaEpilogue<MT> ep;
    ep.prefetch(g, m0 + 16 * MT * wm, n0 + 32 * wn, lg, lj);
    float2 lparts[XFA == 2 ? MT : 1][8];      // XFA == 2: the LayerNorm parts of this lane's fragment rows (d <= 256: registers)
    int64_t lrow[XFA == 2 ? MT : 1];
    float gbv[2][2];
    if constexpr (XFA == 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int k = tid + 256 * u < g.K ? tid + 256 * u : 0; gbv[u][0] = ln->gamma[k]; gbv[u][1] = ln->beta[k]; }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = m0 + 16 * MT * wm + 16 * i + lj;
            lrow[i] = (int64_t)z1 * g.M + (row < g.M ? row : g.M - 1);
            const float2* st = reinterpret_cast<const float2*>(ln->stats) + lrow[i] * ln->npart;
#pragma unroll
            for (int p = 0; p < 8; ++p) lparts[i][p] = st[p < ln->npart ? p : 0];
        }
    }
#pragma unroll
    for (int j = 0; j < NS - 1; ++j) issue(j);
    // XFA: the A fragments are 10 log10(power) values; the rest of the dB conversion (reference of batch entry z1, top_db floor,
    // affine) is applied to them in registers, ahead of the MFMAs that consume them (db_finish_fast: 2 vector instructions per
    // value under the matrix pipe) instead of a conversion pass (and a launch) ahead of the product.  The reference is read here, behind the DMA
    // requests: its round trip runs beside the first stage's.
    if constexpr (XFA == 2) {
        // the values requested ahead of the DMA prologue (below) are used here, behind it: the compiler's counted wait for them leaves
        // the younger DMA requests in flight
        for (int u = 0; u < 2; ++u) { const int k = tid + 256 * u; if (k < g.K) { gb[k] = gbv[u][0]; gb[g.K + k] = gbv[u][1]; } }
        for (int k = tid + 512; k < g.K; k += 256) { gb[k] = ln->gamma[k]; gb[g.K + k] = ln->beta[k]; }      // K > 512: the rest, plainly
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // in LDS before the prologue's (raw) barrier
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (ln->npart <= 8) ln_combine8(lparts[i], ln->npart, ln->eps, lrs[i], lnm[i]);
            else ln_combine(reinterpret_cast<const float2*>(ln->stats) + lrow[i] * ln->npart, ln->npart, ln->eps, lrs[i], lnm[i]);
        }
    }
    if constexpr (XFA == 1) {
        float ref_db, floor_db;
        log_window_consts(xf->lp, __uint_as_float(xf->ref_bits[z1]), ref_db, floor_db);
        xscale = xf->lp.db_scale; xc1 = db_fast_c1(xf->lp, ref_db);
    }
    km_wait_vmcnt<(NS - 2) * L>();                 // stage 0 of this thread has landed
    __builtin_amdgcn_s_barrier();                  // ... and everyone's
    KM_TILE_STAMP(2);
    f32x4 af0[MT][2], bf0[2][2], af1[MT][2], bf1[2][2];
    fetch(0, af0, bf0);
    int it = 0;
    for (; it + 1 < kt; it += 2) {                 // two steps per iteration: the fragment register sets alternate statically
        step(it, af0, bf0, af1, bf1);
        step(it + 1, af1, bf1, af0, bf0);
    }
    if (it < kt) step(it, af0, bf0, af1, bf1);
    km_wait_vmcnt<0>();           // the out-of-range stages behind the last one are still writing zeros into the ring
    KM_TILE_STAMP(3);
    ep.finish(g, C, acc, m0 + 16 * MT * wm, n0 + 32 * wn, lg, lj, stats, (int64_t)z1 * g.M);
    __syncthreads();              // the caller may reuse smem (another tile of the same workgroup)
}
