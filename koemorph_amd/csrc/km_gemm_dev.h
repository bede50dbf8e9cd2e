// Device side of the strided exact-fp32 MFMA GEMM for SMALL problems (training step at 8 windows per GPU: 4 to 160 tiles per
// product).  One BM x 64 tile (BM = 32 or 64) per 256 threads, BK = 32, operands through strides like gemm_kernel.  What a
// handful of workgroups needs is a short dependent chain per k-step, so:
//   * per-thread source pointers are computed ONCE; a k-step adds an offset (gemm_kernel spent more vector cycles on index
//     arithmetic per step than on its 32 MFMAs -- and on gfx950 fp32 MFMA and VALU share one ALU budget,
//     tools/micro/coissue.hip);
//   * an operand that is contiguous along k (A row-major, nn.Linear weights) or along its rows (transposed operands of
//     the weight-gradient products) moves as 16-byte loads; anything else falls back to scalar loads of the same slots;
//   * LDS image [k / 4][row][k % 4]: a lane's fragments for FOUR consecutive MFMAs are one ds_read_b128 (lane group g
//     contracts k = 16 kb + 4 g + s in MFMA s; valid because A and B use the same partition), 16 lanes = 16 rows = all
//     64 banks; k-contiguous operands are committed with one ds_write_b128 per global float4;
//   * the loads of the next D k-tiles are in flight in registers, LDS tiles are double buffered: one barrier per k-step.
// Included inside namespace km of a .hip translation unit that defines f32x4 and KM_MFMA.
#pragma once

#include "km_gemm.h"

#ifndef KM_TILE_STAMP
#define KM_TILE_STAMP(i)      /* tools/micro/tile_bench.hip: wall-clock stamps inside a tile */
#endif

namespace ggd {
constexpr int BN = 64, BK = 32;
constexpr int lds_floats(int BM) { return 2 * (BM + BN) * BK; }      // A and B tiles, two buffers each
constexpr int LDS_FLOATS = lds_floats(64);                            // 32 KB
}

// One operand tile of ROWS rows x 32 k: slot e of a thread is four elements that are CONTIGUOUS in memory:
//   mode 0 (k-contiguous operand):   slot = (row r, k group kg): k = 4 kg .. 4 kg + 3
//   mode 1 (row-contiguous operand): slot = (k, rows 4 r4 .. 4 r4 + 3)
// Loads are branch-free buffer loads (one b128 when VEC, else four b32): rows outside the matrix point past the
// descriptor's range and read zeros, elements past K or past the last row are cleared with selects.  No branch encloses a
// load, so the compiler counts them (s_waitcnt vmcnt(N)) and the D-deep prefetch really overlaps: with `if (in range)`
// around the loads every commit waited for vmcnt(0), i.e. for the tiles requested last.
//
// Round 4 -- who loads what.  In mode 0 consecutive lanes used to take consecutive ROWS (64 lanes x 16 bytes out of 64
// different cache lines per instruction).  tools/micro/tile_bench.hip: every further k-tile of prefetch delayed the first
// tile's arrival by 0.33 us -- the texture path looks up one line per cycle, so such an instruction costs 64 cycles and the
// tile loop was bound by ISSUING its loads (36 GB/s per CU), not by their latency.  Now the eight k groups of a row are
// eight consecutive lanes: an instruction covers 8 rows x one whole 128-byte line.  The LDS image [k / 4][row][k % 4] is
// XOR-swizzled (row ^ (k / 4) within a plane) so that the eight lanes of a line, which land in eight different planes,
// hit eight different bank groups with one ds_write_b128 each; the fragment reads (16 consecutive rows of one plane) stay
// conflict-free because the swizzle permutes rows inside aligned groups of eight.
typedef unsigned int gd_u32x4 __attribute__((ext_vector_type(4)));
template <int ROWS, bool VEC>
struct OperandTile {
    static constexpr int E = ROWS * 8 / 256;
    static constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned off[E];         // byte offset of the slot in k-tile 0 (OOB: the row is outside the matrix)
    int kq[E];               // mode 0: first k of the slot inside a tile; mode 1: its k
    int nrow[E];             // mode 1: rows of the slot inside the matrix (0 .. 4)
    int lds[E];              // float offset of the slot's first element in the LDS image (mode 1: of its row group, see commit)
    unsigned kstep;          // bytes from one k-tile to the next
    int mode, K;

    __device__ __forceinline__ void init(const float* b, int64_t rs, int64_t ks, int row0, int nrows, int K_, int64_t extent_floats,
                                         int tid) {
        K = K_;
        mode = ks == 1 ? 0 : 1;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, (unsigned)(extent_floats * 4), 0x00020000);
        kstep = (unsigned)(ggd::BK * ks * 4);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + 256 * e;
            if (mode == 1) {
                const int r4 = idx & (ROWS / 4 - 1), k = idx / (ROWS / 4), g0 = row0 + 4 * r4;
                const int plane = k >> 2;                                   // k / 4: 0 .. 7 inside a tile, the same in every tile
                kq[e] = k;
                nrow[e] = nrows - g0 < 0 ? 0 : (nrows - g0 > 4 ? 4 : nrows - g0);
                off[e] = g0 < nrows ? (unsigned)(((int64_t)k * ks + g0) * 4) : OOB;
                // rows 4 r4 + j, j < 4, sit at (4 r4 + j) ^ plane = ((4 r4) ^ (plane & 4)) + (j ^ (plane & 3))
                lds[e] = (plane * ROWS + ((4 * r4) ^ (plane & 4))) * 4 + (k & 3);
            } else {
                const int kg = idx & 7, r = idx >> 3, g = row0 + r;         // eight consecutive lanes = one 128-byte line of a row
                kq[e] = 4 * kg;
                nrow[e] = 4;
                off[e] = g < nrows ? (unsigned)(((int64_t)g * rs + 4 * kg) * 4) : OOB;
                lds[e] = (kg * ROWS + (r ^ kg)) * 4;
            }
        }
    }
    // global -> registers for k-tile `nk` (kb_bytes = byte offset of the contraction batch)
    __device__ __forceinline__ void load(unsigned kb_bytes, int nk, bool dead, float4 (&v)[E]) const {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            // a row outside the matrix keeps its offset >= 2^31 (operands are < 2^31 bytes); a stage past the end is all OOB
            const unsigned o = dead ? OOB : off[e] + kb_bytes + (unsigned)nk * kstep;
            float4 t;
            if constexpr (VEC) {
                const gd_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
                t = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
            } else {
                t.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o, 0, 0));
                t.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o + 4, 0, 0));
                t.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o + 8, 0, 0));
                t.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o + 12, 0, 0));
            }
            v[e] = t;
        }
    }
    // clear what lies past K / past the last row (selects only), then registers -> the swizzled LDS image: a k-contiguous slot is
    // one ds_write_b128, a slot of four rows four ds_write_b32 (row j of the slot at offset 4 (j ^ (plane & 3)) floats).
    __device__ __forceinline__ void commit(float* tile, int nk, const float4 (&v)[E]) const {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = nk * ggd::BK + kq[e];
            const int nv = mode == 1 ? (k < K ? nrow[e] : 0) : K - k;
            float* d = tile + lds[e];
            const float4 z = make_float4(nv > 0 ? v[e].x : 0.f, nv > 1 ? v[e].y : 0.f, nv > 2 ? v[e].z : 0.f, nv > 3 ? v[e].w : 0.f);
            if (mode == 0) {
                *reinterpret_cast<float4*>(d) = z;
            } else {
                const int x = (kq[e] >> 2) & 3;
                d[4 * x] = z.x;                    // j = 0 -> 0 ^ x
                d[4 * (1 ^ x)] = z.y;
                d[4 * (2 ^ x)] = z.z;
                d[4 * (3 ^ x)] = z.w;
            }
        }
    }
};

// fragment of four consecutive k (one plane) for 16 rows starting at the multiple of 16 `rowbase`: lane (lg, lj) reads plane
// 4 kb + lg, row rowbase + (lj ^ plane)
template <int ROWS>
__device__ __forceinline__ f32x4 tile_fragment(const float* tile, int kb, int lg, int lj, int rowbase) {
    const int plane = 4 * kb + lg;
    return *reinterpret_cast<const f32x4*>(tile + (plane * ROWS + rowbase + (lj ^ (plane & 7))) * 4);
}

// floats spanned by one output batch of an operand (all contraction batches included)
__host__ __device__ inline int64_t gemm_operand_extent(int64_t nrows, int64_t rs, int64_t K, int64_t ks, int kb_count, int64_t kbs) {
    return (nrows - 1) * rs + (K - 1) * ks + 1 + (int64_t)((kb_count > 1 ? kb_count : 1) - 1) * kbs;
}
// 16-byte loads are legal for the operand (base of every output batch 16-byte aligned, slot offsets multiples of 4 floats)
__host__ __device__ inline bool gemm_operand_vec(const float* base, int64_t rs, int64_t ks, int64_t bs1, int64_t bs2, int64_t kbs) {
    const bool al = (reinterpret_cast<uintptr_t>(base) & 15) == 0 && (bs1 & 3) == 0 && (bs2 & 3) == 0 && (kbs & 3) == 0;
    return ks == 1 ? (al && (rs & 3) == 0) : (al && (ks & 3) == 0);
}

// accumulators of one wave (MT row tiles x 2 column tiles of 16 x 16) -> C: alpha, bias, beta C, activation, dropout.
// Every load (bias, the old C for beta, the dropout bytes) is issued before the first store -- a store to C between them would
// order each later load behind it (C may alias anything as far as the compiler knows): 8 - 16 dependent round trips per
// lane, 1.4 - 2.5 us per tile (tools/micro/tile_bench.hip).  Out-of-range elements load from a clamped, valid address.
template <int MT>
__device__ __forceinline__ void gemm_tile_epilogue(const GemmArgs& g, float* C, const f32x4 (&acc)[MT][2], int mw, int nw, int lg, int lj) {
    // (uniform branches around whole groups of loads, never `cond ? load : constant` per element: the compiler turns that into a
    // select of ADDRESSES with the constant in scratch memory, and the 4 KB kernel argument of phase_kernel follows it there)
    float add[MT][2][4], old[MT][2][4];
    unsigned char keep[MT][2][4];
    const int n_a = nw + lj < g.N ? nw + lj : g.N - 1, n_b = nw + 16 + lj < g.N ? nw + 16 + lj : g.N - 1;     // clamped columns of the lane
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) { add[i][jn][r] = 0.f; old[i][jn][r] = 0.f; keep[i][jn][r] = 1; }
    if (g.bias_mode == 1) {
        const float ba = g.bias[n_a], bb = g.bias[n_b];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { add[i][0][r] = ba; add[i][1][r] = bb; }
    } else if (g.bias_mode == 2) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw + 16 * i + 4 * lg + r;
                const float bv = g.bias[m < g.M ? m : g.M - 1];
                add[i][0][r] = bv; add[i][1][r] = bv;
            }
    }
    if (g.beta != 0.f) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw + 16 * i + 4 * lg + r;
                const float* row = C + (int64_t)(m < g.M ? m : g.M - 1) * g.c_rs;
                old[i][0][r] = row[n_a]; old[i][1][r] = row[n_b];
            }
    }
    if (g.drop) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw + 16 * i + 4 * lg + r;
                const unsigned char* row = g.drop + gemm_drop_row(g.drop_map, m < g.M ? m : g.M - 1) * g.N;
                keep[i][0][r] = row[n_a]; keep[i][1][r] = row[n_b];
            }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mw + 16 * i + 4 * lg + r;
            float* row = C + (int64_t)m * g.c_rs;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int n = nw + 16 * jn + lj;
                if (m < g.M && n < g.N) {
                    float v = g.alpha * acc[i][jn][r] + add[i][jn][r] + g.beta * old[i][jn][r];
                    v = gemm_act(v, g.relu);
                    if (g.drop) v = keep[i][jn][r] ? v * g.drop_scale : 0.f;
                    row[n] = v;
                }
            }
        }
}

// tile (bx, by) of output batch bz; smem = ggd::lds_floats(BM) floats, 16-byte aligned
template <int BM, int D, bool VA, bool VB>
__device__ __forceinline__ void gemm_tile_dev(const GemmArgs& g, int bx, int by, int bz, float* smem) {
    using namespace ggd;
    constexpr int MT = BM / 32;                              // 16-row MFMA tiles per wave (waves 2 x 2, each (BM / 2) x 32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lg = lane >> 4, lj = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int z1 = bz / g.batch2, z2 = bz - z1 * g.batch2;
    const int m0 = by * BM, n0 = bx * BN;
    OperandTile<BM, VA> ta;
    OperandTile<BN, VB> tb;
    ta.init(g.A + z1 * g.a_bs1 + z2 * g.a_bs2, g.a_rs, g.a_cs, m0, g.M, g.K,
            gemm_operand_extent(g.M, g.a_rs, g.K, g.a_cs, g.kb_count, g.a_kbs), tid);
    tb.init(g.B + z1 * g.b_bs1 + z2 * g.b_bs2, g.b_cs, g.b_rs, n0, g.N, g.K,
            gemm_operand_extent(g.N, g.b_cs, g.K, g.b_rs, g.kb_count, g.b_kbs), tid);
    float* C = g.C + z1 * g.c_bs1 + z2 * g.c_bs2;
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = f32x4{0, 0, 0, 0}; acc[i][1] = f32x4{0, 0, 0, 0}; }
    const int kbn = g.kb_count > 0 ? g.kb_count : 1;
    const int kt = (g.K + BK - 1) / BK, total = kbn * kt;
    // D stages of loads are ALWAYS in flight -- stages past the end are requested out of range and read zeros -- so the
    // compiler can count them: the commit of the oldest stage waits with vmcnt(E (D - 1)), not for everything.
    constexpr int EA = OperandTile<BM, VA>::E, EB = OperandTile<BN, VB>::E;
    float4 ra[D][EA], rb[D][EB];
    auto stage = [&](int it, float4 (&a)[EA], float4 (&b)[EB]) {
        const int nb = it / kt, nk = it - nb * kt;
        const bool dead = it >= total;                                          // uniform: a stage past the end reads zeros
        ta.load((unsigned)(nb * g.a_kbs * 4), nk, dead, a);
        tb.load((unsigned)(nb * g.b_kbs * 4), nk, dead, b);
    };
    KM_TILE_STAMP(1);
#pragma unroll
    for (int s = 0; s < D; ++s) stage(s, ra[s], rb[s]);
    for (int it0 = 0; it0 < total; it0 += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {                          // static register slots: no rotation (a move would wait for the load)
            const int it = it0 + s;
            float* As = smem + (s & 1) * ((BM + BN) * BK);     // D is even: step parity == slot parity
            float* Bs = As + BM * BK;
            const int nk_c = it % kt;
            ta.commit(As, nk_c, ra[s]);                        // a stage past the end commits zeros
            tb.commit(Bs, nk_c, rb[s]);
            // one barrier per step: the tile written two steps from now reuses this buffer, and every wave passes the
            // NEXT step's barrier (behind its own MFMAs of this step) before anyone gets there
            __syncthreads();
            if (it == 0) KM_TILE_STAMP(2);
            stage(it + D, ra[s], rb[s]);
            if (it < total) {                                  // uniform; only LDS reads and MFMAs are conditional
#pragma unroll
                for (int kb = 0; kb < BK / 16; ++kb) {
                    f32x4 af[MT], bf[2];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        af[i] = tile_fragment<BM>(As, kb, lg, lj, 16 * MT * wm + 16 * i);
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        bf[jn] = tile_fragment<BN>(Bs, kb, lg, lj, 32 * wn + 16 * jn);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            acc[i][0] = KM_MFMA(af[i][q], bf[0][q], acc[i][0]);
                            acc[i][1] = KM_MFMA(af[i][q], bf[1][q], acc[i][1]);
                        }
                }
            }
        }
    }
    static_assert(D % 2 == 0, "the LDS double buffer is indexed by slot parity");
    KM_TILE_STAMP(3);
    gemm_tile_epilogue<MT>(g, C, acc, m0 + 16 * MT * wm, n0 + 32 * wn, lg, lj);
    __syncthreads();      // the caller may reuse smem (another tile of the same workgroup)
}
