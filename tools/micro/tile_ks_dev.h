// Intra-workgroup split-K form of the 32/64 x 64 tile (KG groups of four waves share a tile's contraction), kept for
// tools/micro/tile_bench.hip only: measured in round 4, no faster than the four-wave tile inside the training step
// (0.2113 against 0.2005 ms per 8-window step) -- a tile's MFMA work (1.7 us for 32 x 64 x 256 on one CU) and its instruction
// count per k-step, not the length of its k chain, were what bounded it.  Included after km_gemm_dev.h.
#pragma once

// ---- the same tile on KG x 4 waves: the contraction dealt to KG groups of four waves (intra-workgroup split-K) --------------
// At 8 windows per step a product is 4 - 160 tiles on 256 CUs: ONE four-wave workgroup per CU, one wave per SIMD, and nothing
// to hide the chain of a k-step behind (global load -> ds_write -> barrier -> ds_read -> MFMA: ~0.7 us per 32-deep step whatever
// the prefetch depth; rocprofv3 with train_op_per_launch: a K = 256 product 10 - 11 us, K = 640 18 us, an element-wise
// operation 3 - 5 us).  More tiles do not help -- the chain per tile stays -- so the chain itself is cut: a workgroup of
// 256 KG threads, group kg runs the k-steps kg, kg + KG, ... of the SAME output tile on its own pair of LDS buffers (all
// groups' loads are in flight at once: K = 256 is requested whole at kernel entry), the partial accumulators of groups
// 1 .. KG - 1 meet group 0's in LDS and are added in group order (a fixed order: bit-reproducible), group 0 runs the epilogue.
// Four waves per SIMD also issue ~3x the instructions per cycle of one (tools/micro/xlane_rate.hip).
// smem = KG * ggd::lds_floats(BM) floats.
template <int BM, int KG, int D, bool VA, bool VB>
__device__ __forceinline__ void gemm_tile_ks_dev(const GemmArgs& g, int bx, int by, int bz, float* smem) {
    using namespace ggd;
    constexpr int MT = BM / 32;
    const int kg = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
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
    const int iters = (total + KG - 1) / KG;                  // steps per group; the same for every group (one barrier each)
    float* gsm = smem + kg * lds_floats(BM);
    constexpr int EA = OperandTile<BM, VA>::E, EB = OperandTile<BN, VB>::E;
    float4 ra[D][EA], rb[D][EB];
    auto stage = [&](int j, float4 (&a)[EA], float4 (&b)[EB]) {          // step j of this group = k-tile j * KG + kg of the product
        const int it = j * KG + kg;
        const int nb = it / kt, nk = it - nb * kt;
        const bool dead = it >= total;                                          // uniform per wave: a stage past the end reads zeros
        ta.load((unsigned)(nb * g.a_kbs * 4), nk, dead, a);
        tb.load((unsigned)(nb * g.b_kbs * 4), nk, dead, b);
    };
    KM_TILE_STAMP(1);
#pragma unroll
    for (int s = 0; s < D; ++s) stage(s, ra[s], rb[s]);
    for (int j0 = 0; j0 < iters; j0 += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int j = j0 + s, it = j * KG + kg;
            float* As = gsm + (s & 1) * ((BM + BN) * BK);
            float* Bs = As + BM * BK;
            ta.commit(As, it % kt, ra[s]);
            tb.commit(Bs, it % kt, rb[s]);
            __syncthreads();                                   // all KG groups: every group runs the same number of steps
            if (j == 0) KM_TILE_STAMP(2);
            stage(j + D, ra[s], rb[s]);
            if (it < total) {
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
    // partial accumulators of groups 1 .. KG - 1 -> LDS [group - 1][wave][MT * 2 tiles][lane] (one ds_write_b128 per tile),
    // added by group 0 in group order
    __syncthreads();                                           // every group is done with its tile buffers
    KM_TILE_STAMP(3);
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (kg > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) red[(((kg - 1) * 4 + wave) * (MT * 2) + i * 2 + jn) * 64 + lane] = acc[i][jn];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (int q = 1; q < KG; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const f32x4 t = red[(((q - 1) * 4 + wave) * (MT * 2) + i * 2 + jn) * 64 + lane];
                    acc[i][jn][0] += t[0]; acc[i][jn][1] += t[1]; acc[i][jn][2] += t[2]; acc[i][jn][3] += t[3];
                }
        gemm_tile_epilogue<MT>(g, C, acc, m0 + 16 * MT * wm, n0 + 32 * wn, lg, lj);
    }
    __syncthreads();      // the caller may reuse smem
}
