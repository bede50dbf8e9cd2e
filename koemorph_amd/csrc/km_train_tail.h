// Loss tail of the training step (decoder output -> prediction -> KoeMorphLoss -> dL/dz), shared by the launch-per-op
// chain (km_train.hip) and the phased step (km_trainp.hip).  Included inside namespace km of a .hip translation unit.
#pragma once
#include "km_gridsync.h"

__device__ __forceinline__ int tr_mouth_slot(int i) { return (i >= 14 && i <= 40) ? i - 14 : (i == 51 ? 27 : -1); }
// blendshape index -> expression query slot 0..23 (EXPRESSION_INDICES = 0..13, 41..50; dual_stream_attention.py:14-45)
__device__ __forceinline__ int tr_expr_slot(int i) { return i < 14 ? i : i - 27; }

struct TailArgs {
    const float* zrows;   // (B*28 + B or B*24) decoder logits of the mouth rows, then of the expression rows
    float* zrows_out;     // or null.  Not null: the tail computes the logits itself, z[r] = h[r] . w2 + b2 over the mouth rows
                          // of h1 and the expr_rows * B rows of he, stores them here and uses them (zrows is ignored)
    float* grow;          // (B*28 + B*24) or null: dL/dz per ROW of the hidden activations (only with expr_rows == 24)
    const float* h1;      // (B*28, DH) post-ReLU hidden of the mouth rows
    const float* he;      // (B, DH)    post-ReLU hidden of the (shared) expression row
    const float* w2; const float* b2;
    const float* mel_w; const float* emo_w;   // (52) raw stream weights
    float temperature;
    const float* target;  // (B, 52)
    float* bs;            // (B, 52) sigmoid outputs
    float* out;           // (B, 52) final (after clamp and EMA)
    float* out2;          // (B, 52) or null: the caller's copy of the prediction
    float* dz;            // (B, 52) dL/dz
    float* ema_state;     // (B, 52) or null
    int ema_first;
    const float* alpha_p; // smoothing_alpha parameter
    float mse_w, l1_w;
    km_loss_config lc;    // extra KoeMorphLoss terms (weights 0 = off)
    float* fac;           // (B, 52) scratch: d y / d f of the clamp + EMA
    float* xp;            // (B, 52) scratch: x - previous EMA state
    float* loss;          // (1)
    float* d_melw; float* d_emow; float* d_alpha;   // gradients (52), (52), (1)
    int B, DH;
    int expr_rows;        // expression rows per window in zrows: 1 (eval-mode arithmetic: the 24 rows are identical) or 24
    const float* audio_energy;   // (B) or null: per-window audio energy for the audio-visual term (losses.py:340-378)
    // more than one workgroup (gridDim.x = G > 1; not with the audio-visual term, which couples the whole batch): window b
    // belongs to workgroup b % G; every workgroup leaves its sums in part[G][64] (0..51 d stream-weight sum, 52 loss, 53 d alpha,
    // 54 d b2) and the one that arrives last (ctr, put back to zero by it) adds them up in workgroup order
    float* part; unsigned* ctr;
    float* d_b2;          // or null: gradient of the decoder's output bias = sum of every dL/dz
    int* drop_ctr;        // or null: the dropout generator's step counter, advanced once per step
    // or null: gradients of the decoder's hidden layer, dH[r][m] = dL/dz[r] w2[m] keep_scale [H[r][m] > 0] (H is post-ReLU,
    // post-dropout: H > 0 <=> kept and active), for the mouth rows (dh1) and the 24 expression rows per window (dhe).  The
    // phased step used to spend a phase of its own on this outer product (round 4: one launch less)
    float* dh1; float* dhe; float keep_scale;
};

// One workgroup: decoder output layer, sigmoid, stream weights, clamp, EMA, loss and the gradient of the loss
// with respect to every pre-sigmoid logit.  B is small in training (8 per GPU), all loops are in a fixed order.
// Pass A computes the prediction y (B,52); pass B the loss terms of KoeMorphLoss (src/model/losses.py:112-178) and
// dL/dy -- the smoothness and landmark terms need the whole row of y, hence two passes.
__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }   // torch: d|x|/dx = sign(x)

// 8 waves: wave w owns windows w, w + 8, ...; lane i < 52 owns coefficient i (lanes up to 63 help with the landmark
// products).  Per-thread partial sums run over a wave's windows in order and are combined across waves in wave order,
// so the result does not depend on timing.
constexpr int TAIL_NW = 8;
constexpr int TAIL_AV_MAX = 1024;     // windows per step the audio-visual term can hold in LDS

template <int NW>
__device__ __forceinline__ void train_tail_dev(const TailArgs& a) {
    __shared__ float wsum_s[52], wm_s[52], we_s[52], dws_p[NW][52], red[NW][64], e_s[NW][52], u_s[NW][136];
    __shared__ float av_g[TAIL_AV_MAX], av_c[2];
    const int i = threadIdx.x & 63, w = threadIdx.x >> 6;
    // wave-wide sums / maxima as xor butterflies: a fixed pairing, so the result does not depend on timing
    auto wsum64 = [](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    auto wmax64 = [](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        return v;
    };
    // operands nothing else depends on are requested first: they arrive under the decoder-output pass below (the stream
    // weight softmaxes used to run -- and wait for their loads -- before it, behind a barrier of their own)
    const float mel_w_i = (w == 0 && i < 52) ? a.mel_w[i] : 0.f, emo_w_i = (w == 0 && i < 52) ? a.emo_w[i] : 0.f;
    const float alpha_raw = a.alpha_p[0];
    const float* zr = a.zrows;
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    const int n_own = a.B > wg ? (a.B - wg + G - 1) / G : 0;          // windows wg, wg + G, ... of this workgroup
    if (a.zrows_out) {   // decoder output layer of this workgroup's windows: 16 lanes per row, lanes stride the hidden units
        // four rows per wave at a time, eight such passes in flight: the stores come after the group's loads (a store to
        // zrows_out between them would order every later load behind it), so a wave pays the memory latency once per 32 rows
        const int64_t rm = (int64_t)a.B * 28;
        const int per_win = 28 + a.expr_rows, rows = n_own * per_win;
        const int sub = i >> 4, l16 = i & 15;
        auto row_of = [&](int lr) -> int64_t {                         // local row -> row of zrows / h1 | he
            const int wl = lr / per_win, r = lr - wl * per_win;
            const int64_t b = wg + (int64_t)wl * G;
            return r < 28 ? b * 28 + r : rm + b * a.expr_rows + (r - 28);
        };
        for (int g0 = w * 4; g0 < rows; g0 += NW * 4 * 8) {
            float zl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int lr = g0 + u * NW * 4 + sub;
                float s = 0.f;
                if (lr < rows) {
                    const int64_t r = row_of(lr);
                    const float* h = r < rm ? a.h1 + r * a.DH : a.he + (r - rm) * a.DH;
                    for (int k = l16; k < a.DH; k += 16) s = fmaf(h[k], a.w2[k], s);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
                zl[u] = s;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int lr = g0 + u * NW * 4 + sub;
                if (l16 == 0 && lr < rows) a.zrows_out[row_of(lr)] = zl[u] + a.b2[0];
            }
        }
        zr = a.zrows_out;
        __threadfence_block();
    }
    if (w == 0) {   // stream weight softmaxes (dual_stream_attention.py:252-253), lane k = coefficient k
        const float vm = i < 52 ? mel_w_i / a.temperature : -INFINITY, ve = i < 52 ? emo_w_i / a.temperature : -INFINITY;
        const float mm = wmax64(vm), me = wmax64(ve);
        const float em = i < 52 ? expf(vm - mm) : 0.f, ee = i < 52 ? expf(ve - me) : 0.f;
        const float sm = wsum64(em), se = wsum64(ee);
        if (i < 52) { wm_s[i] = em / sm; we_s[i] = ee / se; wsum_s[i] = 0.5f * wm_s[i] + 0.5f * we_s[i]; }
    }
    __syncthreads();
    const float alpha = 1.0f / (1.0f + expf(-alpha_raw));
    const float inv_n = 1.0f / (float)(a.B * 52);
    const bool ema_on = a.ema_state && !a.ema_first;
    // ---- pass A: y = EMA(clamp(wsum * sigmoid(z))) ----
    if (i < 52) {
        const int slot = tr_mouth_slot(i);
        for (int b = wg + G * w; b < a.B; b += G * NW) {
            const float z = slot >= 0 ? zr[(int64_t)b * 28 + slot]
                                      : zr[(int64_t)a.B * 28 + (a.expr_rows == 1 ? b : (int64_t)b * 24 + tr_expr_slot(i))];
            const float bs = 1.0f / (1.0f + expf(-z));
            const float f = wsum_s[i] * bs;
            const float x = fminf(fmaxf(f, 0.f), 1.f);
            float y = x, dy_dx = 1.f, xp = 0.f;
            if (a.ema_state) {
                float* st = a.ema_state + (int64_t)b * 52 + i;
                if (!a.ema_first) {
                    const float prev = *st;
                    y = alpha * x + (1.0f - alpha) * prev;
                    dy_dx = alpha;
                    xp = x - prev;          // d y / d smoothing_alpha = (x - prev) * alpha (1 - alpha)
                }
                *st = y;
            }
            a.bs[(int64_t)b * 52 + i] = bs;
            a.out[(int64_t)b * 52 + i] = y;
            if (a.out2) a.out2[(int64_t)b * 52 + i] = y;
            a.fac[(int64_t)b * 52 + i] = dy_dx * ((f >= 0.f && f <= 1.f) ? 1.f : 0.f);
            a.xp[(int64_t)b * 52 + i] = xp;
        }
    }
    __threadfence_block();
    __syncthreads();
    const km_loss_config& lc = a.lc;
    // audio-visual consistency (PerceptualBlendshapeLoss._compute_audiovisual_loss, losses.py:340-378):
    //   1 - cos(m, e), m_b = mean of the 20 mouth coefficients of window b, e_b = audio energy; couples the whole batch.
    //   d/dy[b, i in 12..31] = -(1/20) (e^_b - cos m^_b) / |m|   (the two F.normalize + cosine_similarity collapse to this)
    const bool av_on = lc.perceptual_weight > 0.f && a.audio_energy && a.B <= TAIL_AV_MAX && G == 1;
    if (av_on) {
        for (int b = threadIdx.x; b < a.B; b += 64 * NW) {
            float m = 0.f;
            for (int k = 12; k < 32; ++k) m += a.out[(int64_t)b * 52 + k];
            av_g[b] = m * (1.0f / 20.0f);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float nm = 0.f, na = 0.f;
            for (int b = 0; b < a.B; ++b) { nm += av_g[b] * av_g[b]; na += a.audio_energy[b] * a.audio_energy[b]; }
            nm = fmaxf(sqrtf(nm), 1e-12f); na = fmaxf(sqrtf(na), 1e-12f);
            float dot = 0.f, w1 = 0.f, w2 = 0.f;
            for (int b = 0; b < a.B; ++b) {
                const float mh = av_g[b] / nm, ah = a.audio_energy[b] / na;
                dot += mh * ah; w1 += mh * mh; w2 += ah * ah;
            }
            const float corr = dot / sqrtf(fmaxf(w1 * w2, 1e-16f));
            av_c[0] = corr; av_c[1] = nm;
        }
        __syncthreads();
        {
            const float corr = av_c[0], nm = av_c[1];
            float na = 0.f;
            for (int b = 0; b < a.B; ++b) na += a.audio_energy[b] * a.audio_energy[b];
            na = fmaxf(sqrtf(na), 1e-12f);
            __syncthreads();                      // every thread has read av_c / will overwrite av_g with the gradient factor
            for (int b = threadIdx.x; b < a.B; b += 64 * NW)
                av_g[b] = -lc.perceptual_weight * 0.5f * (1.0f / 20.0f) * (a.audio_energy[b] / na - corr * (av_g[b] / nm)) / nm;
            __syncthreads();
        }
    }
    // ---- pass B: loss terms and dL/dy, row by row ----
    const bool have_prev = lc.prev_pred_dev && lc.prev_target_dev;
    const bool t_on = lc.temporal_weight > 0.f && have_prev, v_on = lc.velocity_weight > 0.f && have_prev;
    const bool lm_on = lc.landmark_weight > 0.f && lc.landmark_w_dev;
    const bool dsv_on = lc.ds_velocity_weight > 0.f && lc.ds_prev_pred_dev;
    // perceptual groups (losses.py:306-338): weight / group size
    float pg = 0.f;
    if (i < 52) pg = i < 12 ? 1.0f / 12.f : (i < 32 ? 2.0f / 20.f : (i < 44 ? 1.0f / 12.f : 1.5f / 8.f));
    float loss_acc = 0.f, dws = 0.f, dal = 0.f, db2 = 0.f;
    for (int b = wg + G * w; b < a.B; b += G * NW) {
        float y = 0.f, e = 0.f, dy = 0.f;
        if (i < 52) {
            y = a.out[(int64_t)b * 52 + i];
            e = y - a.target[(int64_t)b * 52 + i];
            loss_acc += (a.mse_w * e * e + a.l1_w * fabsf(e)) * inv_n;
            dy = (a.mse_w * 2.0f * e + a.l1_w * sgnf(e)) * inv_n;
            if (lc.perceptual_weight > 0.f) {
                const float wgt = lc.perceptual_weight * pg / (float)a.B;
                loss_acc += wgt * e * e;
                dy += wgt * 2.0f * e;
                if (av_on && i >= 12 && i < 32) dy += av_g[b];
            }
            if (t_on || v_on) {
                const float dd = (y - lc.prev_pred_dev[(int64_t)b * 52 + i]) -
                                 (a.target[(int64_t)b * 52 + i] - lc.prev_target_dev[(int64_t)b * 52 + i]);
                if (t_on) { loss_acc += lc.temporal_weight * dd * dd * inv_n; dy += lc.temporal_weight * 2.0f * dd * inv_n; }
                if (v_on) { loss_acc += lc.velocity_weight * fabsf(dd) * inv_n; dy += lc.velocity_weight * sgnf(dd) * inv_n; }
            }
            if (dsv_on) {   // DualStreamLoss velocity (train_dual_stream.py:489-495): both differences against the SAME previous prediction
                const float pp = lc.ds_prev_pred_dev[(int64_t)b * 52 + i];
                const float dd = (y - pp) - (a.target[(int64_t)b * 52 + i] - pp);
                loss_acc += lc.ds_velocity_weight * dd * dd * inv_n;
                dy += lc.ds_velocity_weight * 2.0f * dd * inv_n;
            }
            if (lc.sparsity_weight > 0.f) { loss_acc += lc.sparsity_weight * fabsf(y) * inv_n; dy += lc.sparsity_weight * sgnf(y) * inv_n; }
            if (lc.smoothness_weight > 0.f) {   // torch.diff along the 52 coefficients: 51 pairs per row
                const float wgt = lc.smoothness_weight / (float)(a.B * 51);
                if (i > 0) { const float dl = y - a.out[(int64_t)b * 52 + i - 1]; loss_acc += wgt * fabsf(dl); dy += wgt * sgnf(dl); }
                if (i < 51) { const float dr = a.out[(int64_t)b * 52 + i + 1] - y; dy -= wgt * sgnf(dr); }
            }
            e_s[w][i] = e;
        }
        if (lc.ds_separation_weight > 0.f) {   // DualStreamLoss separation (train_dual_stream.py:498-514), one value per window
            const bool mouth = i < 52 && tr_mouth_slot(i) >= 0;
            const float ms = wsum64(mouth ? y : 0.f), es = wsum64((i < 52 && !mouth) ? y : 0.f);
            const float diff = ms * (1.0f / 28.0f) - es * (1.0f / 24.0f);
            const float wgt = lc.ds_separation_weight / (float)a.B;
            if (i == 0) loss_acc += wgt * fabsf(diff);
            if (i < 52) dy += wgt * sgnf(diff) * (mouth ? 1.0f / 28.0f : -1.0f / 24.0f);
        }
        if (lm_on) {   // u = e W^T (136), loss = mean u^2, dL/de = 2/(B 136) u W; only this wave touches e_s[w], u_s[w]
            __builtin_amdgcn_wave_barrier();
            for (int k = i; k < 136; k += 64) {
                float u = 0.f;
                for (int jj = 0; jj < 52; ++jj) u += e_s[w][jj] * lc.landmark_w_dev[k * 52 + jj];
                u_s[w][k] = u;
            }
            __builtin_amdgcn_wave_barrier();
            const float wgt = lc.landmark_weight / (float)(a.B * 136);
            if (i < 52) {
                float gsum = 0.f;
                for (int k = 0; k < 136; ++k) gsum += u_s[w][k] * lc.landmark_w_dev[k * 52 + i];
                dy += wgt * 2.0f * gsum;
            }
            if (i == 0) { float q = 0.f; for (int k = 0; k < 136; ++k) q += u_s[w][k] * u_s[w][k]; loss_acc += wgt * q; }
            __builtin_amdgcn_wave_barrier();
        }
        if (i < 52) {
            if (ema_on) dal += dy * a.xp[(int64_t)b * 52 + i] * alpha * (1.0f - alpha);
            const float df = dy * a.fac[(int64_t)b * 52 + i];
            const float bs = a.bs[(int64_t)b * 52 + i];
            dws += df * bs;
            const float dzv = df * wsum_s[i] * bs * (1.0f - bs);
            a.dz[(int64_t)b * 52 + i] = dzv;
            db2 += dzv;
            if (a.grow) {
                const int slot = tr_mouth_slot(i);
                if (slot >= 0) a.grow[(int64_t)b * 28 + slot] = dzv;
                else a.grow[(int64_t)a.B * 28 + (int64_t)b * 24 + tr_expr_slot(i)] = dzv;
            }
        }
    }
    if (a.dh1 && a.grow) {      // every dL/dz of this workgroup's windows is in a.grow: the hidden layer's gradient, 4 units per thread
        __threadfence_block();
        __syncthreads();
        const int dh4 = a.DH >> 2, per_win = 52 * dh4;
        for (int wl = 0; wl < n_own; ++wl) {
            const int64_t b = wg + (int64_t)wl * G;
            for (int e = threadIdx.x; e < per_win; e += 64 * NW) {
                const int r = e / dh4, m4 = e - r * dh4;
                const int64_t row = r < 28 ? b * 28 + r : (int64_t)a.B * 28 + b * 24 + (r - 28);
                const float gs = a.grow[row], ks = a.keep_scale;
                const float* hp = r < 28 ? a.h1 + row * a.DH : a.he + (row - (int64_t)a.B * 28) * a.DH;
                float* dp = r < 28 ? a.dh1 + row * a.DH : a.dhe + (row - (int64_t)a.B * 28) * a.DH;
                const float4 h = reinterpret_cast<const float4*>(hp)[m4], w = reinterpret_cast<const float4*>(a.w2)[m4];
                reinterpret_cast<float4*>(dp)[m4] = make_float4(h.x > 0.f ? gs * w.x * ks : 0.f, h.y > 0.f ? gs * w.y * ks : 0.f,
                                                                 h.z > 0.f ? gs * w.z * ks : 0.f, h.w > 0.f ? gs * w.w * ks : 0.f);
            }
        }
    }
    // this workgroup's sums: waves in index order per coefficient, then (for the scalars) the butterfly over the coefficients
    __shared__ float tot_s[64];
    if (i < 52) dws_p[w][i] = dws;
    red[w][i] = i < 52 ? loss_acc : 0.f;
    __syncthreads();
    float loss_wg = 0.f, dal_wg = 0.f, db2_wg = 0.f, dws_wg = 0.f;
    if (w == 0) {
        float t = 0.f;
        for (int ww = 0; ww < NW; ++ww) t += red[ww][i];
        loss_wg = wsum64(t);
        if (i < 52)
            for (int ww = 0; ww < NW; ++ww) dws_wg += dws_p[ww][i];
    }
    __syncthreads();
    red[w][i] = i < 52 ? dal : 0.f;
    __syncthreads();
    if (w == 0) {
        float t = 0.f;
        for (int ww = 0; ww < NW; ++ww) t += red[ww][i];
        dal_wg = wsum64(t);
    }
    __syncthreads();
    red[w][i] = i < 52 ? db2 : 0.f;
    __syncthreads();
    if (w == 0) {
        float t = 0.f;
        for (int ww = 0; ww < NW; ++ww) t += red[ww][i];
        db2_wg = wsum64(t);
    }
    // all workgroups: leave the sums, the last one to arrive adds them up in workgroup order
    __shared__ int last_s;
    if (w == 0) {
        const float mine = i < 52 ? dws_wg : (i == 52 ? loss_wg : (i == 53 ? dal_wg : (i == 54 ? db2_wg : 0.f)));
        tot_s[i] = mine;
        if (G > 1) {
            a.part[(int64_t)wg * 64 + i] = mine;
            __threadfence();
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int last = 1;
        if (G > 1) {
            const unsigned arrived = atomicAdd(a.ctr, 1u);
            last = arrived == (unsigned)G - 1u;
            if (last) a.ctr[0] = 0u;
        }
        last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    if (w == 0) {
        float v = tot_s[i];
        if (G > 1) {
            __threadfence();
            v = 0.f;
            for (int g = 0; g < G; ++g) v += a.part[(int64_t)g * 64 + i];
        }
        const float loss_t = __shfl(v, 52), dal_t = __shfl(v, 53), db2_t = __shfl(v, 54);
        if (i == 0) {
            a.loss[0] = loss_t + (av_on ? lc.perceptual_weight * 0.5f * (1.0f - av_c[0]) : 0.f);
            a.d_alpha[0] = dal_t;
            if (a.d_b2) a.d_b2[0] = db2_t;
            if (a.drop_ctr) a.drop_ctr[0] += 1;           // the next step draws fresh dropout masks
        }
        // softmax backward for both stream-weight vectors: d w_j = (1/tau) w_j (g_j - sum_k g_k w_k), g = 0.5 dwsum
        const float dws_t = i < 52 ? v : 0.f;
        const float dm = wsum64(i < 52 ? 0.5f * dws_t * wm_s[i] : 0.f), de = wsum64(i < 52 ? 0.5f * dws_t * we_s[i] : 0.f);
        if (i < 52) {
            a.d_melw[i] = wm_s[i] * (0.5f * dws_t - dm) / a.temperature;
            a.d_emow[i] = we_s[i] * (0.5f * dws_t - de) / a.temperature;
        }
    }
}

// ---- one window per 256-thread workgroup (the phased step, round 4) ---------------------------------------------------------
// The same arithmetic as train_tail_dev, term by term and in the same order (the logits' 16-lane sums included), for a
// workgroup that owns exactly ONE window: its 52 hidden rows are staged in LDS once (they feed the logits AND the hidden layer's
// gradient), the prediction, the loss gradient and dL/dz stay in registers / LDS from the logits to dH, and every operand the
// window needs is requested at kernel entry.  train_tail_dev went to memory and back between its passes (logits -> pass A ->
// pass B -> dH: four dependent round trips and three fences; 17 us per 8-window step for a microsecond of arithmetic).
// Not for the audio-visual term (it couples the windows of the batch: train_tail_dev with one workgroup).
__device__ __forceinline__ void train_tail_window_dev(const TailArgs& a, float* smem) {
    const int tid = threadIdx.x, i = tid & 63, w = tid >> 6;
    const int G = (int)gridDim.x, b = (int)blockIdx.x, DH = a.DH, dh4 = DH >> 2;
    float* Hs = smem;                         // [52][DH]  post-ReLU hidden rows of this window (mouth rows 0..27, expression 28..51)
    float* w2s = Hs + 52 * DH;                // [DH]
    float* zs = w2s + DH;                     // [64] logits by ROW (mouth slot / 28 + expression slot)
    float* gs = zs + 64;                      // [64] dL/dz by row
    float* e_s = gs + 64;                     // [64] prediction error by coefficient (landmark term)
    float* u_s = e_s + 64;                    // [136]
    float* tot_s = u_s + 136;                 // [64]
    int* last_s = reinterpret_cast<int*>(tot_s + 64);
    auto wsum64 = [](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    auto wmax64 = [](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        return v;
    };
    const km_loss_config& lc = a.lc;
    const int64_t rm = (int64_t)a.B * 28;
    // ---- everything the window needs, requested up front ----
    float4 hv[7];                             // 52 dh4 float4 of hidden rows over 256 threads: at most 7 each at DH = 128
    const int nh4 = 52 * dh4;
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const int e = tid + 256 * u, r = e / dh4, m4 = e - r * dh4;
        hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < nh4) hv[u] = reinterpret_cast<const float4*>(r < 28 ? a.h1 + ((int64_t)b * 28 + r) * DH : a.he + ((int64_t)b * 24 + (r - 28)) * DH)[m4];
    }
    const float w2v = tid < DH ? a.w2[tid] : 0.f;
    const bool c52 = w == 0 && i < 52;
    const float mel_w_i = c52 ? a.mel_w[i] : 0.f, emo_w_i = c52 ? a.emo_w[i] : 0.f;
    const float alpha_raw = a.alpha_p[0], b2v = a.b2[0];
    const float tgt = c52 ? a.target[(int64_t)b * 52 + i] : 0.f;
    const bool ema_on = a.ema_state && !a.ema_first;
    const float prev_state = (c52 && ema_on) ? a.ema_state[(int64_t)b * 52 + i] : 0.f;
    const bool have_prev = lc.prev_pred_dev && lc.prev_target_dev;
    const bool t_on = lc.temporal_weight > 0.f && have_prev, v_on = lc.velocity_weight > 0.f && have_prev;
    const bool lm_on = lc.landmark_weight > 0.f && lc.landmark_w_dev;
    const bool dsv_on = lc.ds_velocity_weight > 0.f && lc.ds_prev_pred_dev;
    float pp_t = 0.f, pt_t = 0.f, pp_ds = 0.f;
    if (c52 && (t_on || v_on)) { pp_t = lc.prev_pred_dev[(int64_t)b * 52 + i]; pt_t = lc.prev_target_dev[(int64_t)b * 52 + i]; }
    if (c52 && dsv_on) pp_ds = lc.ds_prev_pred_dev[(int64_t)b * 52 + i];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const int e = tid + 256 * u;
        if (e < nh4) reinterpret_cast<float4*>(Hs)[e] = hv[u];
    }
    if (tid < DH) w2s[tid] = w2v;
    __syncthreads();
    // ---- decoder output layer: z[row] = H[row] . w2 + b2, 16 lanes per row striding the hidden units (train_tail_dev's order) ----
    {
        const int sub = tid >> 4, l16 = tid & 15;          // 16 rows per pass
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = 16 * pass + sub;
            float sacc = 0.f;
            if (r < 52)
                for (int k = l16; k < DH; k += 16) sacc = fmaf(Hs[r * DH + k], w2s[k], sacc);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
            if (l16 == 0 && r < 52) zs[r] = sacc + b2v;
        }
    }
    __syncthreads();
    float loss_acc = 0.f, dws = 0.f, dal = 0.f, db2 = 0.f, wm_i = 0.f, we_i = 0.f;
    if (w == 0) {
        // stream weight softmaxes (dual_stream_attention.py:252-253), lane k = coefficient k
        const float vm = i < 52 ? mel_w_i / a.temperature : -INFINITY, ve = i < 52 ? emo_w_i / a.temperature : -INFINITY;
        const float mm = wmax64(vm), me = wmax64(ve);
        const float em = i < 52 ? expf(vm - mm) : 0.f, ee = i < 52 ? expf(ve - me) : 0.f;
        const float sm = wsum64(em), se = wsum64(ee);
        wm_i = i < 52 ? em / sm : 0.f; we_i = i < 52 ? ee / se : 0.f;
        const float wsum_i = 0.5f * wm_i + 0.5f * we_i;
        const float alpha = 1.0f / (1.0f + expf(-alpha_raw));
        const float inv_n = 1.0f / (float)(a.B * 52);
        // ---- pass A: y = EMA(clamp(wsum * sigmoid(z))) ----
        float y = 0.f, bsig = 0.f, fac = 0.f, xp = 0.f;
        if (i < 52) {
            const int slot = tr_mouth_slot(i);
            const float z = slot >= 0 ? zs[slot] : zs[28 + tr_expr_slot(i)];
            bsig = 1.0f / (1.0f + expf(-z));
            const float f = wsum_i * bsig;
            const float x = fminf(fmaxf(f, 0.f), 1.f);
            y = x;
            float dy_dx = 1.f;
            if (a.ema_state) {
                if (!a.ema_first) { y = alpha * x + (1.0f - alpha) * prev_state; dy_dx = alpha; xp = x - prev_state; }
                a.ema_state[(int64_t)b * 52 + i] = y;
            }
            a.bs[(int64_t)b * 52 + i] = bsig;
            a.out[(int64_t)b * 52 + i] = y;
            if (a.out2) a.out2[(int64_t)b * 52 + i] = y;
            fac = dy_dx * ((f >= 0.f && f <= 1.f) ? 1.f : 0.f);
            a.fac[(int64_t)b * 52 + i] = fac;
            a.xp[(int64_t)b * 52 + i] = xp;
        }
        // ---- pass B: loss terms and dL/dy ----
        float pg = 0.f;
        if (i < 52) pg = i < 12 ? 1.0f / 12.f : (i < 32 ? 2.0f / 20.f : (i < 44 ? 1.0f / 12.f : 1.5f / 8.f));
        const float y_left = __shfl_up(y, 1), y_right = __shfl_down(y, 1);
        float e = 0.f, dy = 0.f;
        if (i < 52) {
            e = y - tgt;
            loss_acc += (a.mse_w * e * e + a.l1_w * fabsf(e)) * inv_n;
            dy = (a.mse_w * 2.0f * e + a.l1_w * sgnf(e)) * inv_n;
            if (lc.perceptual_weight > 0.f) {
                const float wgt = lc.perceptual_weight * pg / (float)a.B;
                loss_acc += wgt * e * e;
                dy += wgt * 2.0f * e;
            }
            if (t_on || v_on) {
                const float dd = (y - pp_t) - (tgt - pt_t);
                if (t_on) { loss_acc += lc.temporal_weight * dd * dd * inv_n; dy += lc.temporal_weight * 2.0f * dd * inv_n; }
                if (v_on) { loss_acc += lc.velocity_weight * fabsf(dd) * inv_n; dy += lc.velocity_weight * sgnf(dd) * inv_n; }
            }
            if (dsv_on) {
                const float dd = (y - pp_ds) - (tgt - pp_ds);
                loss_acc += lc.ds_velocity_weight * dd * dd * inv_n;
                dy += lc.ds_velocity_weight * 2.0f * dd * inv_n;
            }
            if (lc.sparsity_weight > 0.f) { loss_acc += lc.sparsity_weight * fabsf(y) * inv_n; dy += lc.sparsity_weight * sgnf(y) * inv_n; }
            if (lc.smoothness_weight > 0.f) {
                const float wgt = lc.smoothness_weight / (float)(a.B * 51);
                if (i > 0) { const float dl = y - y_left; loss_acc += wgt * fabsf(dl); dy += wgt * sgnf(dl); }
                if (i < 51) { const float dr = y_right - y; dy -= wgt * sgnf(dr); }
            }
            e_s[i] = e;
        }
        if (lc.ds_separation_weight > 0.f) {
            const bool mouth = i < 52 && tr_mouth_slot(i) >= 0;
            const float ms = wsum64(mouth ? y : 0.f), es = wsum64((i < 52 && !mouth) ? y : 0.f);
            const float diff = ms * (1.0f / 28.0f) - es * (1.0f / 24.0f);
            const float wgt = lc.ds_separation_weight / (float)a.B;
            if (i == 0) loss_acc += wgt * fabsf(diff);
            if (i < 52) dy += wgt * sgnf(diff) * (mouth ? 1.0f / 28.0f : -1.0f / 24.0f);
        }
        if (lm_on) {
            __builtin_amdgcn_wave_barrier();
            for (int k = i; k < 136; k += 64) {
                float u = 0.f;
                for (int jj = 0; jj < 52; ++jj) u += e_s[jj] * lc.landmark_w_dev[k * 52 + jj];
                u_s[k] = u;
            }
            __builtin_amdgcn_wave_barrier();
            const float wgt = lc.landmark_weight / (float)(a.B * 136);
            if (i < 52) {
                float gsum = 0.f;
                for (int k = 0; k < 136; ++k) gsum += u_s[k] * lc.landmark_w_dev[k * 52 + i];
                dy += wgt * 2.0f * gsum;
            }
            if (i == 0) { float q = 0.f; for (int k = 0; k < 136; ++k) q += u_s[k] * u_s[k]; loss_acc += wgt * q; }
            __builtin_amdgcn_wave_barrier();
        }
        if (i < 52) {
            if (ema_on) dal += dy * xp * alpha * (1.0f - alpha);
            const float df = dy * fac;
            dws += df * bsig;
            const float dzv = df * wsum_i * bsig * (1.0f - bsig);
            a.dz[(int64_t)b * 52 + i] = dzv;
            db2 += dzv;
            const int slot = tr_mouth_slot(i);
            const int row = slot >= 0 ? slot : 28 + tr_expr_slot(i);
            gs[row] = dzv;
            if (a.grow) {
                if (slot >= 0) a.grow[(int64_t)b * 28 + slot] = dzv;
                else a.grow[rm + (int64_t)b * 24 + tr_expr_slot(i)] = dzv;
            }
        }
    }
    __syncthreads();
    // ---- the hidden layer's gradient dH[r][m] = dL/dz[r] w2[m] keep_scale [H[r][m] > 0], from the LDS copies ----
    if (a.dh1) {
        const float ks = a.keep_scale;
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int e4 = tid + 256 * u;
            if (e4 < nh4) {
                const int r = e4 / dh4, m4 = e4 - r * dh4;
                const float g = gs[r];
                const float4 h = reinterpret_cast<const float4*>(Hs)[e4], wv = reinterpret_cast<const float4*>(w2s)[m4];
                float* dp = r < 28 ? a.dh1 + ((int64_t)b * 28 + r) * DH : a.dhe + ((int64_t)b * 24 + (r - 28)) * DH;
                reinterpret_cast<float4*>(dp)[m4] = make_float4(h.x > 0.f ? g * wv.x * ks : 0.f, h.y > 0.f ? g * wv.y * ks : 0.f,
                                                                 h.z > 0.f ? g * wv.z * ks : 0.f, h.w > 0.f ? g * wv.w * ks : 0.f);
            }
        }
    }
    // ---- this window's sums, then all windows': the workgroup that arrives last adds them up in window order ----
    if (w == 0) {
        const float loss_wg = wsum64(i < 52 ? loss_acc : 0.f), dal_wg = wsum64(i < 52 ? dal : 0.f), db2_wg = wsum64(i < 52 ? db2 : 0.f);
        const float mine = i < 52 ? dws : (i == 52 ? loss_wg : (i == 53 ? dal_wg : (i == 54 ? db2_wg : 0.f)));
        tot_s[i] = mine;
        if (G > 1) {
            // write-through store + drain instead of a release fence (the fence writes the L2 back; km_gridsync.h has the protocol):
            // the last workgroup acquires below
            kmsync::st_wt(a.part + (int64_t)b * 64 + i, mine);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (tid == 0) {
        int last = 1;
        if (G > 1) {
            const unsigned arrived = atomicAdd(a.ctr, 1u);
            last = arrived == (unsigned)G - 1u;
            if (last) a.ctr[0] = 0u;
        }
        *last_s = last;
    }
    __syncthreads();
    if (!*last_s) return;
    if (w == 0) {
        float v = tot_s[i];
        if (G > 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            v = 0.f;
            // the windows' sums in window order, eight loads in flight at a time (one dependent load per window made this loop
            // half of the tail at 64 windows: 25 us)
            for (int g0 = 0; g0 < G; g0 += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = a.part[(int64_t)(g0 + u < G ? g0 + u : g0) * 64 + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += g0 + u < G ? t[u] : 0.f;
            }
        }
        const float loss_t = __shfl(v, 52), dal_t = __shfl(v, 53), db2_t = __shfl(v, 54);
        if (i == 0) {
            a.loss[0] = loss_t;
            a.d_alpha[0] = dal_t;
            if (a.d_b2) a.d_b2[0] = db2_t;
            if (a.drop_ctr) a.drop_ctr[0] += 1;
        }
        const float dws_t = i < 52 ? v : 0.f;
        const float dm = wsum64(i < 52 ? 0.5f * dws_t * wm_i : 0.f), de = wsum64(i < 52 ? 0.5f * dws_t * we_i : 0.f);
        if (i < 52) {
            a.d_melw[i] = wm_i * (0.5f * dws_t - dm) / a.temperature;
            a.d_emow[i] = we_i * (0.5f * dws_t - de) / a.temperature;
        }
    }
}
__host__ __device__ constexpr int train_tail_window_lds_floats(int DH) { return 52 * DH + DH + 64 * 3 + 136 + 64 + 4; }
