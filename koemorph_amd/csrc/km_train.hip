// Training step of the dual-stream core for gfx950: unfolded forward with saved activations, loss, backward,
// global-norm clipping and fused AdamW.  SURVEY.md section 8 row a13 / 8(f) rank 1.
//
// Replaces, per rank, the body of SequentialTrainer.train_epoch (reference src/train_sequential.py:158-181):
//   outputs = model(audio); loss = criterion(...); loss.backward(); clip_grad_norm_(params, 1.0); AdamW.step()
// for the 28 tensors of DualStreamCrossAttention + smoothing_alpha (837 738 fp32 at d=256/T=256).  The mel and
// emotion features carry no gradient in the reference (NumPy round trip), so backward stops at the core inputs.
// Gradients land in ONE flat caller-owned bucket in state-dict order, which is what the data-parallel build
// all-reduces over RCCL (koemorph_amd/parallel.py) before km_train_adamw.
//
// Arithmetic: eval-mode (dropout p = 0) so that parity against torch.autograd on the reference module is exact up
// to summation order (tests/golden/core_*_grads.npz).  Every contraction is the exact-fp32 MFMA GEMM of
// km_generic.hip (NT / NN / TN through strides, contraction over the batch through the k-batch loop), so the
// path is shape generic.
#include <hip/hip_runtime.h>

#include <cmath>

#include "km_context.h"
#include "km_device.h"
#include "km_gemm.h"

namespace km {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#include "km_train_tail.h"

// ---- elementwise / reduction kernels --------------------------------------------------------------------

// y = LayerNorm(x) out of place, statistics saved for the backward pass (one wave per row)
__global__ __launch_bounds__(256) void ln_fwd_save_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows,
                                                          int d, const float* __restrict__ gam, const float* __restrict__ bet,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / d;
    float v = 0.f;
    for (int i = lane; i < d; i += 64) { const float t = p[i] - mean; v += t * t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / d + 1e-5f);
    for (int i = lane; i < d; i += 64) y[row * d + i] = (p[i] - mean) * rstd * gam[i] + bet[i];
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)), dxhat = dy * gamma; dy is overwritten by dx
__global__ __launch_bounds__(256) void ln_bwd_kernel(float* __restrict__ dy, const float* __restrict__ x, int64_t rows, int d,
                                                     const float* __restrict__ gam, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float xh = (x[row * d + i] - mu) * rs, dxh = dy[row * d + i] * gam[i];
        s1 += dxh;
        s2 += dxh * xh;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    s1 /= d; s2 /= d;
    for (int i = lane; i < d; i += 64) {
        const float xh = (x[row * d + i] - mu) * rs, dxh = dy[row * d + i] * gam[i];
        dy[row * d + i] = rs * (dxh - s1 - xh * s2);
    }
}

// dgamma[n] = sum_rows dy * xhat, dbeta[n] = sum_rows dy   (dy BEFORE ln_bwd_kernel overwrites it)
// Rows are split over blockIdx.y (chunk rows each); with gridDim.y > 1 the outputs are per-chunk partials
// [y][d] that reduce_partials_kernel sums in a fixed order (deterministic split reduction).
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t rows,
                                                            int d, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ dgam, float* __restrict__ dbet, int64_t chunk) {
    __shared__ float sg[4][64], sb[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = (r0 + chunk) < rows ? (r0 + chunk) : rows;
    dgam += (int64_t)blockIdx.y * d; dbet += (int64_t)blockIdx.y * d;
    float ag = 0.f, ab = 0.f;
    if (c < d)
        for (int64_t r = r0 + rg; r < r1; r += 4) {
            const float g = dy[r * d + c];
            ag += g * (x[r * d + c] - mean[r]) * rstd[r];
            ab += g;
        }
    sg[rg][threadIdx.x & 63] = ag; sb[rg][threadIdx.x & 63] = ab;
    __syncthreads();
    if (rg == 0 && c < d) {
        dgam[c] = sg[0][threadIdx.x] + sg[1][threadIdx.x] + sg[2][threadIdx.x] + sg[3][threadIdx.x];
        dbet[c] = sb[0][threadIdx.x] + sb[1][threadIdx.x] + sb[2][threadIdx.x] + sb[3][threadIdx.x];
    }
}

// out[n] (+)= sum over rows of m[r * rs + n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ m, int64_t rows, int64_t rs, int n, float* __restrict__ out,
                                                     int accumulate, int64_t chunk) {
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = (r0 + chunk) < rows ? (r0 + chunk) : rows;
    out += (int64_t)blockIdx.y * n;
    float a = 0.f;
    if (c < n)
        for (int64_t r = r0 + rg; r < r1; r += 4) a += m[r * rs + c];
    sh[rg][threadIdx.x & 63] = a;
    __syncthreads();
    if (rg == 0 && c < n) {
        const float s = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// out[c] (+)= sum_y part[y][c]
__global__ void reduce_partials_kernel(const float* __restrict__ part, int S, int n, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float s = 0.f;
    for (int y = 0; y < S; ++y) s += part[(int64_t)y * n + c];
    out[c] = accumulate ? out[c] + s : s;
}

// dS = P * (dP - rowsum(dP * P)), in place on dP; rows of width w <= 128
__global__ __launch_bounds__(256) void softmax_bwd_kernel(float* __restrict__ dp, const float* __restrict__ p, int64_t rows, int w) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float p0 = lane < w ? p[row * w + lane] : 0.f, p1 = lane + 64 < w ? p[row * w + lane + 64] : 0.f;
    const float g0 = lane < w ? dp[row * w + lane] : 0.f, g1 = lane + 64 < w ? dp[row * w + lane + 64] : 0.f;
    float s = g0 * p0 + g1 * p1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane < w) dp[row * w + lane] = p0 * (g0 - s);
    if (lane + 64 < w) dp[row * w + lane + 64] = p1 * (g1 - s);
}

// z[row] = h[row] . w + b, one wave per row (decoder output layer, blendshape_decoder[3])
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ h, int64_t rows, int n, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ z) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s = fmaf(h[row * n + i], w[i], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) z[row] = s + b[0];
}

// per-row logit gradients from dz (B, 52): g[b*28 + slot] for the mouth rows, g[B*28 + b] = sum over the 24 expression
// coefficients (they share one row), and db2 = sum of everything
__global__ __launch_bounds__(64) void row_grads_kernel(const float* __restrict__ dz, int B, float* __restrict__ g, float* __restrict__ db2) {
    __shared__ float red[64];
    const int i = threadIdx.x;
    float tot = 0.f;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float v = i < 52 ? dz[(int64_t)b * 52 + i] : 0.f;
        const int slot = i < 52 ? tr_mouth_slot(i) : 0;
        if (i < 52 && slot >= 0) g[(int64_t)b * 28 + slot] = v;
        red[i] = (i < 52 && slot < 0) ? v : 0.f;
        __syncthreads();
        if (i == 0) {
            float s = 0.f;
            for (int k = 0; k < 52; ++k) s += red[k];
            g[(int64_t)B * 28 + b] = s;
        }
        __syncthreads();
        tot += v;
    }
    if (gridDim.x == 1) {       // single block: also the bias gradient, summed in a fixed order
        red[i] = tot;
        __syncthreads();
        if (i == 0) {
            float s = 0.f;
            for (int k = 0; k < 64; ++k) s += red[k];
            db2[0] = s;
        }
    }
}

// dH[row][m] = g[row] * w2[m] * [H[row][m] > 0]
__global__ void relu_outer_bwd_kernel(const float* __restrict__ g, const float* __restrict__ h, const float* __restrict__ w2,
                                      int64_t rows, int n, float* __restrict__ dh) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const int64_t r = i / n;
    const int m = (int)(i - r * n);
    dh[i] = h[i] > 0.f ? g[r] * w2[m] : 0.f;
}

// out[m] (+)= sum_r g[r] * h[r][m]   (rows split over blockIdx.y like colsum_kernel)
__global__ __launch_bounds__(256) void wcolsum_kernel(const float* __restrict__ h, const float* __restrict__ g, int64_t rows, int n,
                                                      float* __restrict__ out, int accumulate, int64_t chunk) {
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = (r0 + chunk) < rows ? (r0 + chunk) : rows;
    out += (int64_t)blockIdx.y * n;
    float a = 0.f;
    if (c < n)
        for (int64_t r = r0 + rg; r < r1; r += 4) a = fmaf(g[r], h[r * n + c], a);
    sh[rg][threadIdx.x & 63] = a;
    __syncthreads();
    if (rg == 0 && c < n) {
        const float s = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        out[c] = accumulate ? out[c] + s : s;
    }
}

__global__ __launch_bounds__(64 * TAIL_NW) void train_tail_kernel(TailArgs a) { train_tail_dev<TAIL_NW>(a); }

__global__ void zero_kernel(float* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// ---- optimizer ---------------------------------------------------------------------------------------------

// sum of squares of the flat gradient, deterministic two-stage reduction
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part,
                                                            int* __restrict__ steps, int alpha_live) {
    __shared__ float sh[256];
    // the 1-based AdamW step counters live on the device (graph replay): advanced here, one kernel ahead of their readers
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        steps[0] += 1;
        if (alpha_live) steps[1] += 1;
    }
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += g[i] * g[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(64) void sumsq_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += part[i];
        out[0] = sqrtf(s);
    }
}

__global__ void adamw_tick_kernel(int* __restrict__ steps, int alpha_live) {
    steps[0] += 1;
    if (alpha_live) steps[1] += 1;
}

// torch.nn.utils.clip_grad_norm_(max_norm) + torch.optim.AdamW (decoupled weight decay, bias correction)
// The channel encoder weight is read by the training program as rows of KP floats (16-byte rows for the LDS-DMA tile, zeros
// beyond the KT = T + 3 columns).  The copy is kept beside the master parameters -- rewritten wherever they are: here, and by
// train_refresh_padded_weights after an upload -- instead of being rebuilt by an operation of every step.
struct PaddedCopy {
    float* dst; int64_t off; int64_t n; int kt, kp;      // parameters [off, off + n) are (n / kt) rows of kt floats
    __device__ __forceinline__ void put(int64_t i, float v) const {
        const int64_t j = i - off;
        if (dst && j >= 0 && j < n) { const int64_t r = j / kt; dst[r * kp + (j - r * kt)] = v; }
    }
};

// smoothing_alpha is outside the autograd graph whenever the EMA passes its input through (first call / batch-size
// change / smoothing off): torch leaves its .grad as None and AdamW then skips it entirely (no decay, no moment
// update, its own step counter).  alpha_idx / alpha_live / (abc1, abc2) reproduce that.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ g, int64_t n, const float* __restrict__ part,
                                                    float* __restrict__ gnorm_out, float max_norm, float lr, float b1, float b2,
                                                    float eps, float wd, const int* __restrict__ steps, int64_t alpha_idx,
                                                    int alpha_live, PaddedCopy pc) {
    // global gradient norm from the 256 partial sums: every block runs the same fixed-order tree (a butterfly inside each
    // wave, then the four wave sums in wave order), so all blocks agree bit for bit
    __shared__ float sh[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // One group of four consecutive parameters per thread (818 workgroups for the 837 744 parameters: 8.3 us; four groups per thread --
    // 205 workgroups, one round of the chip -- measured 11.1 us: the kernel is bound by the square root + two divisions per parameter,
    // not by its workgroup count).  Parameters, moments and gradients are requested FIRST: their round trip runs beside the norm's
    // (partials -> butterfly -> barrier) instead of behind it.
    constexpr int GR = 1;
    int64_t i0[GR];
    bool whole[GR];
    float4 p4[GR], m4[GR], v4[GR], g4[GR];
#pragma unroll
    for (int u = 0; u < GR; ++u) {
        i0[u] = ((int64_t)blockIdx.x * (256 * GR) + u * 256 + tid) * 4;
        whole[u] = i0[u] + 3 < n && (alpha_idx < i0[u] || alpha_idx > i0[u] + 3) && (reinterpret_cast<uintptr_t>(g) & 15) == 0;
        p4[u] = m4[u] = v4[u] = g4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (whole[u]) {
            p4[u] = *reinterpret_cast<const float4*>(p + i0[u]); m4[u] = *reinterpret_cast<const float4*>(m + i0[u]);
            v4[u] = *reinterpret_cast<const float4*>(v + i0[u]); g4[u] = *reinterpret_cast<const float4*>(g + i0[u]);
        }
    }
    float s = part[tid];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) sh[wave] = s;
    // the 1-based step counters live on the device (advanced by sumsq_partial_kernel) so that the launch can be replayed
    // from a hipGraph: steps[0] = optimizer step, steps[1] = number of updates smoothing_alpha has received.  The bias
    // corrections are the same for every element but smoothing_alpha: one thread computes them for the block.
    if (tid == 0) {
        const int t = steps[0];
        sh[4] = 1.0f - powf(b1, (float)t);
        sh[5] = 1.0f - powf(b2, (float)t);
    }
    __syncthreads();
    const float gnorm = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3]));
    if (blockIdx.x == 0 && tid == 0) gnorm_out[0] = gnorm;
    float scale = 1.f;
    if (max_norm > 0.f) {
        const float c = max_norm / (gnorm + 1e-6f);          // clip_coef, clamped to 1
        scale = c < 1.f ? c : 1.f;
    }
    const float bc1_all = sh[4], bc2_all = sh[5];
    auto update = [&](float& pi, float& mi, float& vi, float gi, float bc1, float bc2) {
        gi *= scale;
        pi *= (1.0f - lr * wd);
        mi = b1 * mi + (1.0f - b1) * gi;
        vi = b2 * vi + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        pi -= (lr / bc1) * mi / denom;
    };
#pragma unroll
    for (int u = 0; u < GR; ++u) {
        if (i0[u] >= n) continue;
        if (whole[u]) {
            update(p4[u].x, m4[u].x, v4[u].x, g4[u].x, bc1_all, bc2_all);
            update(p4[u].y, m4[u].y, v4[u].y, g4[u].y, bc1_all, bc2_all);
            update(p4[u].z, m4[u].z, v4[u].z, g4[u].z, bc1_all, bc2_all);
            update(p4[u].w, m4[u].w, v4[u].w, g4[u].w, bc1_all, bc2_all);
            *reinterpret_cast<float4*>(p + i0[u]) = p4[u];
            *reinterpret_cast<float4*>(m + i0[u]) = m4[u];
            *reinterpret_cast<float4*>(v + i0[u]) = v4[u];
            pc.put(i0[u], p4[u].x); pc.put(i0[u] + 1, p4[u].y); pc.put(i0[u] + 2, p4[u].z); pc.put(i0[u] + 3, p4[u].w);
            continue;
        }
        for (int64_t i = i0[u]; i < n && i < i0[u] + 4; ++i) {
            float bc1 = bc1_all, bc2 = bc2_all;
            if (i == alpha_idx) {
                // smoothing_alpha is outside the autograd graph whenever the EMA passes its input through: no decay, no moment
                // update, its own step counter
                if (!alpha_live) continue;
                const int t = steps[1];
                bc1 = 1.0f - powf(b1, (float)t); bc2 = 1.0f - powf(b2, (float)t);
            }
            float pi = p[i], mi = m[i], vi = v[i];
            update(pi, mi, vi, g[i], bc1, bc2);
            p[i] = pi; m[i] = mi; v[i] = vi;
            pc.put(i, pi);
        }
    }
}

// dst (rows x kp) <- src (rows x kt), columns kt .. kp zero: the padded copy from scratch (parameter upload)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int kt, int kp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * kp) return;
    const int64_t r = i / kp; const int t = (int)(i - r * kp);
    dst[i] = t < kt ? src[r * kt + t] : 0.f;
}

// ---- host orchestration ----------------------------------------------------------------------------------------

static GemmArgs G(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t c_rs,
                  int64_t M, int64_t N, int64_t K) {
    GemmArgs g{};
    g.alpha = 1.f; g.batch2 = 1; g.kb_count = 1;
    g.A = A; g.a_rs = a_rs; g.a_cs = a_cs; g.B = B; g.b_rs = b_rs; g.b_cs = b_cs; g.C = C; g.c_rs = c_rs;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    return g;
}
// C (rows x N) = A (rows x K) W^T (+ bias), W stored (N x K) like nn.Linear
static GemmArgs NT(const float* A, int64_t a_rs, const float* W, int64_t K, float* C, int64_t c_rs, int64_t rows, int64_t N,
                   const float* bias, int relu) {
    GemmArgs g = G(A, a_rs, 1, W, 1, K, C, c_rs, rows, N, K);
    g.bias = bias; g.bias_mode = bias ? 1 : 0; g.relu = relu;
    return g;
}
// C (rows x N) = A (rows x K) W, W stored (K x N)   (grad wrt the input of a Linear: dX = dY W)
static GemmArgs NN(const float* A, int64_t a_rs, const float* W, int64_t w_rs, float* C, int64_t c_rs, int64_t rows, int64_t N, int64_t K) {
    return G(A, a_rs, 1, W, w_rs, 1, C, c_rs, rows, N, K);
}
// C (M x N) = A^T B with A (rows x M), B (rows x N)   (grad wrt the weight of a Linear: dW = dY^T X)
static GemmArgs TN(const float* A, int64_t a_rs, const float* B, int64_t b_rs, float* C, int64_t c_rs, int64_t M, int64_t N, int64_t rows) {
    return G(A, 1, a_rs, B, b_rs, 1, C, c_rs, M, N, rows);
}

static constexpr int kRedSplit = 32;     // row chunks of a split reduction; scratch = kRedSplit x 2 x (2 d) floats

static int split_of(int64_t rows, int64_t& chunk) {
    int S = (int)((rows + 255) / 256);
    if (S > kRedSplit) S = kRedSplit;
    if (S < 1) S = 1;
    chunk = (rows + S - 1) / S;
    return S;
}

static int colsum(Context* c, const float* m, int64_t rows, int64_t rs, int n, float* out, int accumulate, void* stream,
                  float* red = nullptr) {
    if (!red) red = c->tr_red;
    int64_t chunk;
    const int S = split_of(rows, chunk);
    hipStream_t st = (hipStream_t)stream;
    if (S == 1) {
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((n + 63) / 64), 1), dim3(256), 0, st, m, rows, rs, n, out, accumulate, chunk);
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)S), dim3(256), 0, st, m, rows, rs, n, red, 0, chunk);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, red, S, n, out, accumulate);
    }
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

static int ln_param_grad(Context* c, const float* dy, const float* x, int64_t rows, int d, const float* mean, const float* rstd,
                         float* dgam, float* dbet, void* stream, float* red = nullptr) {
    if (!red) red = c->tr_red;
    int64_t chunk;
    const int S = split_of(rows, chunk);
    hipStream_t st = (hipStream_t)stream;
    if (S == 1) {
        hipLaunchKernelGGL(ln_param_grad_kernel, dim3((unsigned)((d + 63) / 64), 1), dim3(256), 0, st, dy, x, rows, d, mean, rstd, dgam, dbet, chunk);
    } else {
        float* pg = red; float* pb = red + (int64_t)kRedSplit * d;
        hipLaunchKernelGGL(ln_param_grad_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)S), dim3(256), 0, st, dy, x, rows, d, mean, rstd, pg, pb, chunk);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, pg, S, d, dgam, 0);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, pb, S, d, dbet, 0);
    }
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

#define RUN(expr) do { if (int rc_ = (expr)) return rc_; } while (0)

int64_t train_act_floats(Context* c) {
    const int64_t d = c->d, H = c->H, NKk = c->NK, DH = c->DH;
    const int64_t R = NKk, Rq = 28;
    return 2 * R * d /* Y0, Y */ + 2 * R /* mean, rstd */ + 2 * R * d /* KV */ + 2 * H * Rq * NKk /* P, dP */ +
           3 * Rq * d /* A, O1, O2 */ + Rq * DH /* H1 */ + 2 * R * d /* dKV */ + R * d /* dY */ + 3 * Rq * d /* dO2, dO1, dA */ +
           Rq * DH /* dH1 */ + 5 * d /* E0, E, Ve, Oe1, Oe2 */ + 2 /* emo stats */ + DH /* He */ + 3 * d + DH /* emotion grads */ +
           6 * 52 /* bs, out, dz, two loss-tail scratch rows, pad */ + 2 * (Rq + 1) /* zrows, row grads */;
}

struct ParamView { const float* p; float* g; };

int train_forward_backward(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short, const float* emo,
                           const float* target, float mse_w, float l1_w, float* flat_grad, float* loss_dev, float* out_dev,
                           float* ema_state, int ema_first, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t d = c->d, H = c->H, hd = c->hd, T = c->T, KT = c->KT, DH = c->DH, NKk = c->NK, ED = c->ED;
    const int64_t R = B * NKk, Rq = B * 28;
    c->tr_alpha_live = ema_state != nullptr && !ema_first;
    auto P = [&](const char* k) -> const float* { return c->tr_params + c->tr_offset.at(k); };
    auto Gd = [&](const char* k) -> float* { return flat_grad + c->tr_offset.at(k); };
    // carve the activation workspace
    float* w = c->tr_act;
    auto take = [&](int64_t n) { float* p = w; w += n; return p; };
    float* Y0 = take(R * d); float* Y = take(R * d); float* mu = take(R); float* rs = take(R);
    float* KV = take(R * 2 * d); float* Pm = take(B * H * 28 * NKk); float* dP = take(B * H * 28 * NKk);
    float* A = take(Rq * d); float* O1 = take(Rq * d); float* O2 = take(Rq * d); float* H1 = take(Rq * DH);
    float* dKV = take(R * 2 * d); float* dY = take(R * d); float* gA = take(Rq * d); float* gB = take(Rq * d); float* gC = take(Rq * d);
    float* dH1 = take(Rq * DH);
    float* E0 = take(B * d); float* E = take(B * d); float* Ve = take(B * d); float* Oe1 = take(B * d); float* Oe2 = take(B * d);
    float* emu = take(B); float* ers = take(B); float* He = take(B * DH);
    float* geA = take(B * d); float* geB = take(B * d); float* geC = take(B * d); float* dHe = take(B * DH);
    float* bs = take(B * 52); float* outb = take(B * 52); float* dz = take(B * 52);
    float* tfac = take(B * 52); float* txp = take(B * 52);
    float* zrows = take(Rq + B); float* grow = take(Rq + B);
    float* Qb = c->tr_q;            // (28, d)
    float* dQb = c->tr_dq;          // (28, d)

    const float* Wce = P("mel_channel_encoder.weight");
    const float* inw = P("mel_attention.in_proj_weight"); const float* inb = P("mel_attention.in_proj_bias");
    const float* einw = P("emotion_attention.in_proj_weight"); const float* einb = P("emotion_attention.in_proj_bias");
    const float scale = 1.0f / std::sqrt((float)hd);
    const int64_t tv = T_in < T ? T_in : T;

    // Two streams: the emotion stream (forward and backward) and the decoder weight gradients depend on nothing the
    // mel chain produces in between, so they run on an internal side stream next to it -- one fork and one join per
    // half of the step (events only: capturable in a hipGraph) take ~25 small launches off the critical path.
    hipStream_t s2 = (hipStream_t)c->tr_s2;
    void* stream2 = c->tr_s2;
    float* red2 = c->tr_red2;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((c->tr_nparams + 255) / 256)), dim3(256), 0, st, flat_grad, c->tr_nparams);
    HIP_TRY(hipEventRecord((hipEvent_t)c->tr_ev[0], st));
    HIP_TRY(hipStreamWaitEvent(s2, (hipEvent_t)c->tr_ev[0], 0));
    // ================= forward (unfolded, dual_stream_attention.py:189-270) =================
    {   // Y0 = X^T Wce^T + b: long rows then the 3 short-term rows
        GemmArgs g = G(mel, 1, NKk, Wce, 1, KT, Y0, d, NKk, d, tv);
        g.a_bs1 = T_in * NKk; g.c_bs1 = NKk * d; g.bias = P("mel_channel_encoder.bias"); g.bias_mode = 1;
        RUN(launch_gemm(g, (int)B, stream));
        g = G(mel_short, 1, NKk, Wce + T, 1, KT, Y0, d, NKk, d, 3);
        g.a_bs1 = 3 * NKk; g.c_bs1 = NKk * d; g.beta = 1.f;
        RUN(launch_gemm(g, (int)B, stream));
    }
    hipLaunchKernelGGL(ln_fwd_save_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, Y0, Y, R, (int)d, P("mel_norm.weight"),
                       P("mel_norm.bias"), mu, rs);
    RUN(launch_gemm(NT(P("mouth_queries"), d, inw, d, Qb, d, 28, d, inb, 0), 1, stream));                 // Q = mq Wq^T + bq
    RUN(launch_gemm(NT(Y, d, inw + d * d, d, KV, 2 * d, R, 2 * d, inb + d, 0), 1, stream));               // [K | V]
    {   // S = scale * Q_h K_h^T, softmax
        GemmArgs g = G(Qb, d, 1, KV, 1, 2 * d, Pm, NKk, 28, NKk, hd);
        g.alpha = scale; g.batch2 = (int)H; g.a_bs2 = hd; g.b_bs1 = NKk * 2 * d; g.b_bs2 = hd;
        g.c_bs1 = H * 28 * NKk; g.c_bs2 = 28 * NKk;
        RUN(launch_gemm(g, (int)(B * H), stream));
    }
    RUN(launch_softmax_rows(Pm, B * H * 28, (int)NKk, stream));
    {   // A[b][:, h] = P V_h
        GemmArgs g = G(Pm, NKk, 1, KV + d, 2 * d, 1, A, d, 28, hd, NKk);
        g.batch2 = (int)H; g.a_bs1 = H * 28 * NKk; g.a_bs2 = 28 * NKk; g.b_bs1 = NKk * 2 * d; g.b_bs2 = hd;
        g.c_bs1 = 28 * d; g.c_bs2 = hd;
        RUN(launch_gemm(g, (int)(B * H), stream));
    }
    RUN(launch_gemm(NT(A, d, P("mel_attention.out_proj.weight"), d, O1, d, Rq, d, P("mel_attention.out_proj.bias"), 0), 1, stream));
    RUN(launch_gemm(NT(O1, d, P("mel_output_proj.weight"), d, O2, d, Rq, d, P("mel_output_proj.bias"), 0), 1, stream));
    RUN(launch_gemm(NT(O2, d, P("blendshape_decoder.0.weight"), d, H1, DH, Rq, DH, P("blendshape_decoder.0.bias"), 1), 1, stream));
    // emotion stream: one token, softmax == 1 (:216-218, :234-240) -- side stream
    RUN(launch_gemm(NT(emo, ED, P("emotion_encoder.weight"), ED, E0, d, B, d, P("emotion_encoder.bias"), 0), 1, stream2));
    hipLaunchKernelGGL(ln_fwd_save_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s2, E0, E, B, (int)d, P("emotion_norm.weight"),
                       P("emotion_norm.bias"), emu, ers);
    RUN(launch_gemm(NT(E, d, einw + 2 * d * d, d, Ve, d, B, d, einb + 2 * d, 0), 1, stream2));
    RUN(launch_gemm(NT(Ve, d, P("emotion_attention.out_proj.weight"), d, Oe1, d, B, d, P("emotion_attention.out_proj.bias"), 0), 1, stream2));
    RUN(launch_gemm(NT(Oe1, d, P("emotion_output_proj.weight"), d, Oe2, d, B, d, P("emotion_output_proj.bias"), 0), 1, stream2));
    RUN(launch_gemm(NT(Oe2, d, P("blendshape_decoder.0.weight"), d, He, DH, B, DH, P("blendshape_decoder.0.bias"), 1), 1, stream2));
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s2, He, B, (int)DH,
                       P("blendshape_decoder.3.weight"), P("blendshape_decoder.3.bias"), zrows + Rq);
    HIP_TRY(hipEventRecord((hipEvent_t)c->tr_ev[1], s2));
    HIP_TRY(hipStreamWaitEvent(st, (hipEvent_t)c->tr_ev[1], 0));

    // ================= loss and dL/dz =================
    {
        hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((Rq + 3) / 4)), dim3(256), 0, st, H1, Rq, (int)DH,
                           P("blendshape_decoder.3.weight"), P("blendshape_decoder.3.bias"), zrows);
        TailArgs t{};
        t.zrows = zrows; t.h1 = H1; t.he = He; t.w2 = P("blendshape_decoder.3.weight"); t.b2 = P("blendshape_decoder.3.bias");
        t.mel_w = P("mel_weights"); t.emo_w = P("emotion_weights"); t.temperature = c->cfg.temperature; t.target = target;
        t.bs = bs; t.out = outb; t.dz = dz; t.ema_state = ema_state; t.ema_first = ema_first; t.alpha_p = P("smoothing_alpha");
        t.mse_w = mse_w; t.l1_w = l1_w; t.lc = c->tr_loss_cfg; t.fac = tfac; t.xp = txp; t.loss = loss_dev; t.d_melw = Gd("mel_weights"); t.d_emow = Gd("emotion_weights");
        t.d_alpha = Gd("smoothing_alpha"); t.B = (int)B; t.DH = (int)DH; t.expr_rows = 1; t.audio_energy = c->tr_loss_cfg.audio_energy_dev; t.out2 = out_dev;
        hipLaunchKernelGGL(train_tail_kernel, dim3(1), dim3(64 * TAIL_NW), 0, st, t);
    }

    // ================= backward =================
    {   // decoder output layer backward: row logit gradients, dH = g w2 [H > 0], dw2 = sum_rows g H, db2
        hipLaunchKernelGGL(row_grads_kernel, dim3(1), dim3(64), 0, st, dz, (int)B, grow, Gd("blendshape_decoder.3.bias"));
        const float* w2p = P("blendshape_decoder.3.weight");
        hipLaunchKernelGGL(relu_outer_bwd_kernel, dim3((unsigned)((Rq * DH + 255) / 256)), dim3(256), 0, st, grow, H1, w2p, Rq, (int)DH, dH1);
        hipLaunchKernelGGL(relu_outer_bwd_kernel, dim3((unsigned)((B * DH + 255) / 256)), dim3(256), 0, st, grow + Rq, He, w2p, B, (int)DH, dHe);
        HIP_TRY(hipEventRecord((hipEvent_t)c->tr_ev[2], st));
        HIP_TRY(hipStreamWaitEvent(s2, (hipEvent_t)c->tr_ev[2], 0));
        // ---- side stream: decoder output / decoder[0] parameter gradients, then the whole emotion stream backward ----
        int64_t chunk;
        const int S = split_of(Rq, chunk);
        float* dw2 = Gd("blendshape_decoder.3.weight");
        if (S == 1) {
            hipLaunchKernelGGL(wcolsum_kernel, dim3((unsigned)((DH + 63) / 64), 1), dim3(256), 0, s2, H1, grow, Rq, (int)DH, dw2, 0, chunk);
        } else {
            hipLaunchKernelGGL(wcolsum_kernel, dim3((unsigned)((DH + 63) / 64), (unsigned)S), dim3(256), 0, s2, H1, grow, Rq, (int)DH, red2, 0, chunk);
            hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((DH + 255) / 256)), dim3(256), 0, s2, red2, S, (int)DH, dw2, 0);
        }
        hipLaunchKernelGGL(wcolsum_kernel, dim3((unsigned)((DH + 63) / 64), 1), dim3(256), 0, s2, He, grow + Rq, B, (int)DH, dw2, 1, B);
    }
    // decoder[0] (shared by both streams): dW1 = dH1^T O2 + dHe^T Oe2, db1 (side stream); dO2 = dH1 W1 (main), dOe2 = dHe W1 (side)
    RUN(launch_gemm(TN(dH1, DH, O2, d, Gd("blendshape_decoder.0.weight"), d, DH, d, Rq), 1, stream2));
    { GemmArgs g = TN(dHe, DH, Oe2, d, Gd("blendshape_decoder.0.weight"), d, DH, d, B); g.beta = 1.f; RUN(launch_gemm(g, 1, stream2)); }
    RUN(colsum(c, dH1, Rq, DH, (int)DH, Gd("blendshape_decoder.0.bias"), 0, stream2, red2));
    RUN(colsum(c, dHe, B, DH, (int)DH, Gd("blendshape_decoder.0.bias"), 1, stream2, red2));
    RUN(launch_gemm(NN(dHe, DH, P("blendshape_decoder.0.weight"), d, geA, d, B, d, DH), 1, stream2));              // dOe2
    RUN(launch_gemm(TN(geA, d, Oe1, d, Gd("emotion_output_proj.weight"), d, d, d, B), 1, stream2));
    RUN(colsum(c, geA, B, d, (int)d, Gd("emotion_output_proj.bias"), 0, stream2, red2));
    RUN(launch_gemm(NN(geA, d, P("emotion_output_proj.weight"), d, geB, d, B, d, d), 1, stream2));                    // dOe1
    RUN(launch_gemm(TN(geB, d, Ve, d, Gd("emotion_attention.out_proj.weight"), d, d, d, B), 1, stream2));
    RUN(colsum(c, geB, B, d, (int)d, Gd("emotion_attention.out_proj.bias"), 0, stream2, red2));
    RUN(launch_gemm(NN(geB, d, P("emotion_attention.out_proj.weight"), d, geC, d, B, d, d), 1, stream2));             // dVe
    RUN(launch_gemm(TN(geC, d, E, d, Gd("emotion_attention.in_proj_weight") + 2 * d * d, d, d, d, B), 1, stream2));   // only the V third
    RUN(colsum(c, geC, B, d, (int)d, Gd("emotion_attention.in_proj_bias") + 2 * d, 0, stream2, red2));
    RUN(launch_gemm(NN(geC, d, einw + 2 * d * d, d, geA, d, B, d, d), 1, stream2));                                   // dE
    RUN(ln_param_grad(c, geA, E0, B, (int)d, emu, ers, Gd("emotion_norm.weight"), Gd("emotion_norm.bias"), stream2, red2));
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s2, geA, E0, B, (int)d, P("emotion_norm.weight"), emu, ers);
    RUN(launch_gemm(TN(geA, d, emo, ED, Gd("emotion_encoder.weight"), ED, d, ED, B), 1, stream2));
    RUN(colsum(c, geA, B, d, (int)d, Gd("emotion_encoder.bias"), 0, stream2, red2));
    // ---- main stream: the mel chain ----
    RUN(launch_gemm(NN(dH1, DH, P("blendshape_decoder.0.weight"), d, gA, d, Rq, d, DH), 1, stream));               // dO2
    // Every activation gradient of the mel chain has its own buffer (gA = dO2, gB = dO1, gC = dA, dKV, dQb, dY), so the
    // parameter gradients that hang off it (TN products and column sums) go to the side stream as soon as their input
    // exists, and the main stream only carries the chain of input gradients.
    auto to_side = [&](int ev) -> int {      // side stream continues after what the main stream has produced so far
        HIP_TRY(hipEventRecord((hipEvent_t)c->tr_ev[ev], st));
        HIP_TRY(hipStreamWaitEvent(s2, (hipEvent_t)c->tr_ev[ev], 0));
        return KM_OK;
    };
    // mel_output_proj
    RUN(to_side(4));
    RUN(launch_gemm(TN(gA, d, O1, d, Gd("mel_output_proj.weight"), d, d, d, Rq), 1, stream2));
    RUN(colsum(c, gA, Rq, d, (int)d, Gd("mel_output_proj.bias"), 0, stream2, red2));
    RUN(launch_gemm(NN(gA, d, P("mel_output_proj.weight"), d, gB, d, Rq, d, d), 1, stream));                        // dO1
    // out_proj
    RUN(to_side(5));
    RUN(launch_gemm(TN(gB, d, A, d, Gd("mel_attention.out_proj.weight"), d, d, d, Rq), 1, stream2));
    RUN(colsum(c, gB, Rq, d, (int)d, Gd("mel_attention.out_proj.bias"), 0, stream2, red2));
    RUN(launch_gemm(NN(gB, d, P("mel_attention.out_proj.weight"), d, gC, d, Rq, d, d), 1, stream));                 // dA
    float* gin_w = Gd("mel_attention.in_proj_weight"); float* gin_b = Gd("mel_attention.in_proj_bias");
    {   // dP = dA_h V_h^T ; dV_h = P^T dA_h
        GemmArgs g = G(gC, d, 1, KV + d, 1, 2 * d, dP, NKk, 28, NKk, hd);
        g.batch2 = (int)H; g.a_bs1 = 28 * d; g.a_bs2 = hd; g.b_bs1 = NKk * 2 * d; g.b_bs2 = hd; g.c_bs1 = H * 28 * NKk; g.c_bs2 = 28 * NKk;
        RUN(launch_gemm(g, (int)(B * H), stream));
        g = G(Pm, 1, NKk, gC, d, 1, dKV + d, 2 * d, NKk, hd, 28);
        g.batch2 = (int)H; g.a_bs1 = H * 28 * NKk; g.a_bs2 = 28 * NKk; g.b_bs1 = 28 * d; g.b_bs2 = hd; g.c_bs1 = NKk * 2 * d; g.c_bs2 = hd;
        RUN(launch_gemm(g, (int)(B * H), stream));
    }
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((B * H * 28 + 3) / 4)), dim3(256), 0, st, dP, Pm, B * H * 28, (int)NKk);
    {   // dQ_h = scale * sum_b dS K_h  (contraction over the batch through the k-batch loop) ; dK_h = scale * dS^T Q_h
        GemmArgs g = G(dP, NKk, 1, KV, 2 * d, 1, dQb, d, 28, hd, NKk);
        g.alpha = scale; g.batch2 = (int)H; g.a_bs2 = 28 * NKk; g.b_bs2 = hd; g.c_bs2 = hd;
        g.kb_count = (int)B; g.a_kbs = H * 28 * NKk; g.b_kbs = NKk * 2 * d;
        RUN(launch_gemm(g, (int)H, stream));
        g = G(dP, 1, NKk, Qb, d, 1, dKV, 2 * d, NKk, hd, 28);
        g.alpha = scale; g.batch2 = (int)H; g.a_bs1 = H * 28 * NKk; g.a_bs2 = 28 * NKk; g.b_bs2 = hd; g.c_bs1 = NKk * 2 * d; g.c_bs2 = hd;
        RUN(launch_gemm(g, (int)(B * H), stream));
    }
    // in_proj: rows [0,d) = Wq, [d,3d) = [Wk; Wv] -- parameter gradients on the side stream
    RUN(to_side(6));
    RUN(launch_gemm(TN(dQb, d, P("mouth_queries"), d, gin_w, d, d, d, 28), 1, stream2));
    RUN(colsum(c, dQb, 28, d, (int)d, gin_b, 0, stream2, red2));
    RUN(launch_gemm(NN(dQb, d, inw, d, Gd("mouth_queries"), d, 28, d, d), 1, stream2));
    RUN(launch_gemm(TN(dKV, 2 * d, Y, d, gin_w + d * d, d, 2 * d, d, R), 1, stream2));
    RUN(colsum(c, dKV, R, 2 * d, (int)(2 * d), gin_b + d, 0, stream2, red2));
    RUN(launch_gemm(NN(dKV, 2 * d, inw + d * d, d, dY, d, R, d, 2 * d), 1, stream));                                 // dY
    // LayerNorm (its parameter gradients read dY before ln_bwd_kernel rewrites it in place: same stream)
    RUN(ln_param_grad(c, dY, Y0, R, (int)d, mu, rs, Gd("mel_norm.weight"), Gd("mel_norm.bias"), stream));
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, dY, Y0, R, (int)d, P("mel_norm.weight"), mu, rs);
    {   // channel encoder: dWce = sum_b dY0_b^T X_b (long columns on the main stream; the 3 short-term columns and db beside it)
        RUN(to_side(7));
        GemmArgs g = G(dY, 1, d, mel, 1, NKk, Gd("mel_channel_encoder.weight"), KT, d, tv, NKk);
        g.kb_count = (int)B; g.a_kbs = NKk * d; g.b_kbs = T_in * NKk;
        RUN(launch_gemm(g, 1, stream));
        g = G(dY, 1, d, mel_short, 1, NKk, Gd("mel_channel_encoder.weight") + T, KT, d, 3, NKk);
        g.kb_count = (int)B; g.a_kbs = NKk * d; g.b_kbs = 3 * NKk;
        RUN(launch_gemm(g, 1, stream2));
        RUN(colsum(c, dY, R, d, (int)d, Gd("mel_channel_encoder.bias"), 0, stream2, red2));
    }
    HIP_TRY(hipEventRecord((hipEvent_t)c->tr_ev[3], s2));
    HIP_TRY(hipStreamWaitEvent(st, (hipEvent_t)c->tr_ev[3], 0));      // join: every gradient is in the bucket
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

int train_adamw(Context* c, const float* flat_grad, float lr, float b1, float b2, float eps, float wd, float max_norm,
                int64_t step, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = c->tr_nparams;
    const int nb = 256;
    // two launches: partial sums of squares (+ the step counters), then norm + clip + AdamW in one kernel
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, flat_grad, n, c->tr_part, c->tr_steps, c->tr_alpha_live ? 1 : 0);
    (void)step;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, c->tr_params, c->tr_m, c->tr_v, flat_grad, n,
                       c->tr_part, c->tr_gnorm, max_norm, lr, b1, b2, eps, wd, c->tr_steps, c->tr_offset.at("smoothing_alpha"),
                       c->tr_alpha_live ? 1 : 0,
                       PaddedCopy{c->trp_wcep, c->tr_offset.at("mel_channel_encoder.weight"), (int64_t)c->d * c->KT, (int)c->KT, (int)trainp_kp(c)});
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// after the master parameters were written from outside (km_train_init, km_train_set_params): the padded copy from scratch
int train_refresh_padded_weights(Context* c, void* stream) {
    if (!c->trp_wcep) return KM_OK;
    const int64_t KP = trainp_kp(c);
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((c->d * KP + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       c->tr_params + c->tr_offset.at("mel_channel_encoder.weight"), c->trp_wcep, (int64_t)c->d, (int)c->KT, (int)KP);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

}  // namespace km
