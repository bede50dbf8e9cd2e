// Strided-batched exact-fp32 MFMA GEMM shared by km_generic.hip (inference on generic shapes, legacy model)
// and km_train.hip (training forward / backward).
#pragma once

#include <cstdint>

namespace km {

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    int64_t a_rs, a_cs, b_rs, b_cs, c_rs;       // A(m,k) = A[m*a_rs + k*a_cs], B(k,n) = B[k*b_rs + n*b_cs], C(m,n) = C[m*c_rs + n]
    int batch2;                                  // output batch index z = z1 * batch2 + z2
    int64_t a_bs1, a_bs2, b_bs1, b_bs2, c_bs1, c_bs2;
    // contraction batch: the K loop runs kb_count times with A += a_kbs, B += b_kbs (C = sum_kb A_kb B_kb);
    // used for gradients that sum over the windows of a batch
    int kb_count;
    int64_t a_kbs, b_kbs;
    float alpha, beta;
    int bias_mode;                               // 0 none, 1 bias[n], 2 bias[m]
    int relu;                                    // epilogue activation (gemm_act): 0 none, 1 ReLU, 2 exact-erf GELU, 3 SiLU, 4 LeakyReLU(0.1)
    // training-mode dropout in the epilogue (after the activation): v *= drop[mask_row(m) * N + n] ? drop_scale : 0.
    // drop_map: 0 rows map 1:1, 1 rows are (window, mouth slot) -> window * 52 + MOUTH_INDICES[slot], 2 the same for
    // (window, expression slot).  Only the 64 x 64 tile path (gemm_kernel / gemm_tile_dev) implements it.
    const unsigned char* drop;
    float drop_scale;
    int drop_map;
};

#if defined(__HIPCC__)
// erf(a) in fp32, branch-free, 24 vector instructions: both minimax polynomials of N. Juffa's single-precision erff
// (x + x P(x^2) below 0.9277, 1 - exp(Q(|x|)) above: each < 1 ulp from erf with an exact exp) and a select; exp through v_exp_f32
// (Q <= -1.06, so exp(Q) <= 0.35 and its few-ulp error is a fraction of an ulp of the result): <= 1.5 ulp over all, checked
// against math.erf in tests/test_erf_poly.py.  The device library's erff is 36 instructions plus a divergent branch; fp32 VALU
// work comes straight out of the fp32 matrix pipe's time, and GELU is 32 values per lane and hidden chunk in the
// fused KoeMorphModel encoder.  NaN stays NaN.
__device__ __forceinline__ float km_erff(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}

// GemmArgs::relu.  NaN stays NaN in every branch, as in torch.
__device__ __forceinline__ float gemm_act(float v, int code) {
    if (code == 1) return v < 0.f ? 0.f : v;                                          // nn.ReLU
    if (code == 2) return 0.5f * v * (1.0f + km_erff(v * 0.70710678118654752f));      // nn.GELU (exact erf)
    if (code == 3) return v / (1.0f + expf(-v));                                      // nn.SiLU: x sigmoid(x)
    if (code == 4) return v < 0.f ? 0.1f * v : v;                                     // nn.LeakyReLU(0.1)
    return v;
}
#endif

// row of a (rows, N) activation buffer -> row of the (B, 52, N) dropout mask (see GemmArgs::drop_map)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int64_t gemm_drop_row(int map, int64_t m) {
    if (map == 1) { const int64_t b = m / 28; const int q = (int)(m - b * 28); return b * 52 + (q < 27 ? 14 + q : 51); }
    if (map == 2) { const int64_t b = m / 24; const int q = (int)(m - b * 24); return b * 52 + (q < 14 ? q : q + 27); }
    return m;
}

int launch_gemm(const GemmArgs& g, int batch, void* stream);
int launch_softmax_rows(float* x, int64_t rows, int w, void* stream);

}  // namespace km
