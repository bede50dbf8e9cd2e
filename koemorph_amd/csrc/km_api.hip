// C-ABI entry points that touch the device (declared in include/koemorph.h).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstring>
#include <vector>

#include "km_context.h"
#include "km_device.h"
#include "km_gemm.h"

using namespace km;

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static int free_streams(Context* c);
static int free_train(Context* c);
static int free_pipeline(Context* c);

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int need_ready(Context* c) {
    if (!c) return fail(KM_ERR_INVALID_ARG, "NULL handle");
    if (!c->dev_finalized) return fail(KM_ERR_NOT_FINALIZED, "call km_finalize after loading the state dict");
    return KM_OK;
}

static int need_dual(Context* c) {
    if (int rc = need_ready(c)) return rc;
    if (c->kind != 0) return fail(KM_ERR_INVALID_ARG, "this entry point needs a dual-stream handle (km_create), not a legacy one");
    return KM_OK;
}

namespace km { LogParams plan_log_params(MelPlan* p); }

extern "C" {

int km_finalize(km_handle h, void* stream) {
    if (!h) return fail(KM_ERR_INVALID_ARG, "NULL handle");
    Context* c = h;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(KM_ERR_HIP, "no HIP device: libkoemorph_hip has no CPU fallback");
    HIP_TRY(hipGetDevice(&c->device));
    if (!c->host_finalized)
        if (int rc = finalize_host(c)) return rc;
    for (auto& kv : c->packed) {
        Packed& p = kv.second;
        if (!p.dev) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p.dev), p.host.size() * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(p.dev, p.host.data(), p.host.size() * sizeof(float), hipMemcpyHostToDevice,
                               (hipStream_t)stream));
    }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (MelPlan* p : c->mel_plans)
        if (int rc = upload_mel_plan(p)) return rc;
    if (c->kind != 2)                                   // the front end's chunk counters exist before any launch could be captured
        if (int rc = ensure_chunk_counters(c, 4096, stream)) return rc;
    c->dev_finalized = true;
    return KM_OK;
}

static int free_ws(Context* c) {
    if (c->ws_zemo) HIP_TRY(hipFree(c->ws_zemo));
    if (c->ws_zemo_win) HIP_TRY(hipFree(c->ws_zemo_win));
    if (c->ws_melpow) HIP_TRY(hipFree(c->ws_melpow));
    if (c->ws_melmax) HIP_TRY(hipFree(c->ws_melmax));
    if (c->ws_mel) HIP_TRY(hipFree(c->ws_mel));
    if (c->ws_short) HIP_TRY(hipFree(c->ws_short));
    if (c->ws_generic) HIP_TRY(hipFree(c->ws_generic));
    c->ws_zemo = c->ws_zemo_win = c->ws_melpow = c->ws_mel = c->ws_short = c->ws_generic = nullptr;
    c->ws_melmax = nullptr;
    c->ws_windows = c->ws_samples = c->ws_frames = 0;
    c->ws_mels = 0;
    return KM_OK;
}

int km_reserve(km_handle h, int64_t max_windows, int64_t max_samples) {
    if (!h || max_windows <= 0 || max_samples < 0) return fail(KM_ERR_INVALID_ARG, "km_reserve: bad argument");
    Context* c = h;
    if (c->kind == 2) return fail(KM_ERR_INVALID_ARG, "KoeMorphModel handles take no audio: use km_koemorph_reserve");
    // frames for the smallest hop among the registered plans (upper bound for all of them)
    int min_hop = c->cfg.mel.hop_length, max_mels = c->cfg.mel.n_mels;
    for (MelPlan* p : c->mel_plans) {
        if (p->cfg.hop_length < min_hop) min_hop = p->cfg.hop_length;
        if (p->cfg.n_mels > max_mels) max_mels = p->cfg.n_mels;
    }
    const int64_t frames = max_samples > 0 ? 1 + max_samples / min_hop : 0;
    const bool need_generic = c->host_finalized && !c->fused_ok;
    if (max_windows <= c->ws_windows && frames <= c->ws_frames && max_mels <= c->ws_mels && (c->ws_generic || !need_generic)) return KM_OK;
    const int64_t W = max_windows > c->ws_windows ? max_windows : c->ws_windows;
    const int64_t F = frames > c->ws_frames ? frames : c->ws_frames;
    const int64_t S = max_samples > c->ws_samples ? max_samples : c->ws_samples;
    if (int rc = free_ws(c)) return rc;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_zemo), (size_t)W * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_zemo_win), (size_t)W * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_melmax), (size_t)W * sizeof(unsigned)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_short), (size_t)W * 3 * max_mels * sizeof(float)));
    if (F > 0) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_melpow), (size_t)W * F * max_mels * sizeof(float)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_mel), (size_t)W * F * max_mels * sizeof(float)));
    }
    if (need_generic) {
        const int64_t per = c->kind == 1 ? legacy_ws_floats(c, F > 0 ? F : 1) : generic_ws_floats(c);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_generic), (size_t)W * per * sizeof(float)));
    }
    c->ws_windows = W; c->ws_frames = F; c->ws_samples = S; c->ws_mels = max_mels;
    c->melmax_dirty = true;
    return ensure_chunk_counters(c, W, nullptr);
}

int km_destroy(km_handle h) {
    if (!h) return KM_OK;
    Context* c = h;
    for (auto& kv : c->packed)
        if (kv.second.dev) (void)hipFree(kv.second.dev);
    (void)free_ws(c);
    if (c->ws_chunkctr) (void)hipFree(c->ws_chunkctr);
    (void)free_streams(c);
    (void)free_train(c);
    (void)free_pipeline(c);
    for (void* q : {(void*)c->seq_pow, (void*)c->seq_fmax, (void*)c->seq_edge, (void*)c->seq_emax})
        if (q) (void)hipFree(q);
    if (c->pipe_s1) {
        (void)hipStreamDestroy((hipStream_t)c->pipe_s1); (void)hipStreamDestroy((hipStream_t)c->pipe_s2);
        for (int i = 0; i < 2; ++i) {
            (void)hipEventDestroy((hipEvent_t)c->pipe_ev_in[i]); (void)hipEventDestroy((hipEvent_t)c->pipe_ev_mel[i]);
            (void)hipEventDestroy((hipEvent_t)c->pipe_ev_core[i]);
        }
    }
    for (MelPlan* p : c->mel_plans) free_mel_plan(p);
    for (void* e : c->stage_ev)
        if (e) (void)hipEventDestroy((hipEvent_t)e);
    delete h;
    return KM_OK;
}

int64_t km_mel_num_frames(km_handle h, int64_t L) {
    if (!h || L < 0) return -1;
    return 1 + L / h->cfg.mel.hop_length;    // librosa center=True: 1 + len(y) // hop
}

int km_mel_batch(km_handle h, const float* audio_dev, int64_t B, int64_t L, float* mel_long_dev,
                 float* mel_short_dev, void* stream) {
    if (int rc = need_ready(h)) return rc;
    if (!audio_dev || !mel_long_dev || B <= 0 || L <= 0) return fail(KM_ERR_INVALID_ARG, "km_mel_batch: bad argument");
    return launch_mel(h, h->mel_plans[0], audio_dev, B, L, 0, mel_long_dev, mel_short_dev, stream);
}

int km_mel_extract(km_handle h, const km_mel_config* cfg, const float* audio_dev, int64_t B, int64_t L,
                   int64_t out_frames, float* mel_dev, void* stream) {
    if (int rc = need_ready(h)) return rc;
    if (!cfg || !audio_dev || !mel_dev || B <= 0 || L <= 0 || out_frames < 0)
        return fail(KM_ERR_INVALID_ARG, "km_mel_extract: bad argument");
    if (cfg->n_fft != 512 && cfg->n_fft != 1024) return fail(KM_ERR_UNSUPPORTED, "n_fft must be 512 or 1024");
    if (cfg->hop_length <= 0 || cfg->n_mels <= 0 || cfg->n_mels > 128) return fail(KM_ERR_INVALID_ARG, "bad hop_length / n_mels");
    MelPlan* p = nullptr;
    for (MelPlan* q : h->mel_plans)
        if (std::memcmp(&q->cfg, cfg, sizeof(*cfg)) == 0) p = q;
    if (!p) {   // first use of this configuration: build + upload (allocates; not capturable)
        p = find_or_add_plan(h, *cfg);
        if (int rc = upload_mel_plan(p)) return rc;
        if (cfg->n_mels > h->ws_mels && h->ws_windows > 0) {
            // the workspace rows were sized for narrower plans: regrow it now (this first-use path allocates anyway)
            HIP_TRY(hipDeviceSynchronize());
            if (int rc = km_reserve(h, h->ws_windows, h->ws_samples)) return rc;
        }
    }
    return launch_mel(h, p, audio_dev, B, L, out_frames, mel_dev, nullptr, stream);
}

int km_core_forward(km_handle h, const float* mel_dev, int64_t B, int64_t T_in, const float* mel_short_dev,
                    const float* emotion_dev, float* out_dev, float* raw_dev, float* attn_mel_dev, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (!mel_dev || !mel_short_dev || !emotion_dev || !out_dev || B <= 0 || T_in <= 0)
        return fail(KM_ERR_INVALID_ARG, "km_core_forward: bad argument");
    if (!aligned16(mel_dev) || !aligned16(mel_short_dev))
        return fail(KM_ERR_INVALID_ARG, "km_core_forward: mel pointers must be 16-byte aligned");
    if (B > c->ws_windows) return fail(KM_ERR_WORKSPACE, "workspace holds %lld windows, need %lld: call km_reserve",
                                       (long long)c->ws_windows, (long long)B);
    if (int rc = launch_emotion(c, emotion_dev, B, c->ws_zemo, stream)) return rc;
    if (!c->fused_ok) {
        if (!c->ws_generic) return fail(KM_ERR_WORKSPACE, "generic workspace missing: call km_reserve after km_finalize");
        return launch_core_generic(c, mel_dev, B, T_in, mel_short_dev, c->ws_zemo, out_dev, raw_dev, attn_mel_dev, stream);
    }
    return launch_core_fused(c, mel_dev, B, T_in, mel_short_dev, c->ws_zemo, out_dev, raw_dev, attn_mel_dev,
                             nullptr, 1, stream);
}

int km_emotion_logit(km_handle h, const float* emotion_dev, int64_t B, float* z_dev, void* stream) {
    if (int rc = need_dual(h)) return rc;
    if (!emotion_dev || !z_dev || B <= 0) return fail(KM_ERR_INVALID_ARG, "km_emotion_logit: bad argument");
    return launch_emotion(h, emotion_dev, B, z_dev, stream);
}

int km_core_forward_z(km_handle h, const float* mel_dev, int64_t B, int64_t T_in, const float* mel_short_dev,
                      const float* z_dev, float* out_dev, float* raw_dev, float* attn_mel_dev, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (!mel_dev || !mel_short_dev || !z_dev || !out_dev || B <= 0 || T_in <= 0)
        return fail(KM_ERR_INVALID_ARG, "km_core_forward_z: bad argument");
    if (!aligned16(mel_dev) || !aligned16(mel_short_dev))
        return fail(KM_ERR_INVALID_ARG, "km_core_forward_z: mel pointers must be 16-byte aligned");
    if (!c->fused_ok) {
        if (B > c->ws_windows || !c->ws_generic) return fail(KM_ERR_WORKSPACE, "generic workspace missing or too small: call km_reserve after km_finalize");
        return launch_core_generic(c, mel_dev, B, T_in, mel_short_dev, z_dev, out_dev, raw_dev, attn_mel_dev, stream);
    }
    return launch_core_fused(c, mel_dev, B, T_in, mel_short_dev, z_dev, out_dev, raw_dev, attn_mel_dev, nullptr, 1, stream);
}

static int free_train(Context* c) {
    void* ptrs[] = {c->tr_params, c->tr_m, c->tr_v, c->tr_act, c->tr_q, c->tr_dq, c->tr_part, c->tr_gnorm, c->tr_loss, c->tr_red, c->tr_steps};
    for (void* p : ptrs)
        if (p) HIP_TRY(hipFree(p));
    c->tr_params = c->tr_m = c->tr_v = c->tr_act = c->tr_q = c->tr_dq = c->tr_part = c->tr_gnorm = c->tr_loss = c->tr_red = nullptr;
    c->tr_steps = nullptr;
    if (c->tr_red2) { HIP_TRY(hipFree(c->tr_red2)); c->tr_red2 = nullptr; }
    if (c->trp_act) { HIP_TRY(hipFree(c->trp_act)); c->trp_act = nullptr; c->trp_act_floats = 0; }
    if (c->trp_split) { HIP_TRY(hipFree(c->trp_split)); c->trp_split = nullptr; c->trp_split_floats = 0; }
    if (c->trp_wcep) { HIP_TRY(hipFree(c->trp_wcep)); c->trp_wcep = nullptr; }
    if (c->trp_tail_part) { HIP_TRY(hipFree(c->trp_tail_part)); c->trp_tail_part = nullptr; }
    if (c->trp_tail_ctr) { HIP_TRY(hipFree(c->trp_tail_ctr)); c->trp_tail_ctr = nullptr; }
    if (c->trp_masks) { HIP_TRY(hipFree(c->trp_masks)); c->trp_masks = nullptr; }
    if (c->trp_drop_ctr) { HIP_TRY(hipFree(c->trp_drop_ctr)); c->trp_drop_ctr = nullptr; }
    if (c->tr_s2) { (void)hipStreamDestroy((hipStream_t)c->tr_s2); c->tr_s2 = nullptr; }
    for (auto& e : c->tr_ev)
        if (e) { (void)hipEventDestroy((hipEvent_t)e); e = nullptr; }
    c->tr_windows = 0;
    return KM_OK;
}

static int upload_train_params(Context* c, void* stream) {
    std::vector<float> flat((size_t)c->tr_nparams);
    for (const auto& k : c->param_order) {
        const HostParam& hp = c->params.at(k);
        std::memcpy(flat.data() + c->tr_offset.at(k), hp.data.data(), hp.data.size() * sizeof(float));
    }
    HIP_TRY(hipMemcpyAsync(c->tr_params, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    if (int rc = train_refresh_padded_weights(c, stream)) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return KM_OK;
}

int km_train_init(km_handle h, int64_t max_windows, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (max_windows <= 0) return fail(KM_ERR_INVALID_ARG, "km_train_init: bad max_windows");
    if (int rc = free_train(c)) return rc;
    c->tr_offset.clear();
    // Flat layout: state-dict order within two groups; every offset a multiple of 4 floats (16 B).  The second group holds
    // the tensors whose gradients the backward pass finishes LAST (channel encoder, LayerNorm parameters, emotion encoder,
    // mouth queries: 17 % of the bucket), so the all-reduce of the first 83 % can start while they are still being
    // computed (km_train_grad_split / km_train_wait_early).
    auto late = [](const std::string& k) {
        return k == "mouth_queries" || k.rfind("mel_channel_encoder.", 0) == 0 || k.rfind("mel_norm.", 0) == 0 ||
               k.rfind("emotion_norm.", 0) == 0 || k.rfind("emotion_encoder.", 0) == 0;
    };
    int64_t off = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) c->tr_early = off;
        for (const auto& k : c->param_order) {
            if (late(k) != (pass == 1)) continue;
            c->tr_offset[k] = off;
            off += ((int64_t)c->params.at(k).data.size() + 3) / 4 * 4;
        }
    }
    c->tr_nparams = off;
    const size_t nb = (size_t)off * sizeof(float);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_params), nb));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_m), nb));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_v), nb));
    HIP_TRY(hipMemsetAsync(c->tr_params, 0, nb, (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(c->tr_m, 0, nb, (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(c->tr_v, 0, nb, (hipStream_t)stream));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_act), (size_t)max_windows * train_act_floats(c) * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_q), (size_t)28 * c->d * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_dq), (size_t)28 * c->d * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_part), 256 * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_gnorm), sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_loss), sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_red), (size_t)32 * 2 * 2 * c->d * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_red2), (size_t)32 * 2 * 2 * c->d * sizeof(float)));
    {
        hipStream_t s2;
        HIP_TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        c->tr_s2 = s2;
        for (auto& e : c->tr_ev) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e = ev;
        }
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tr_steps), 2 * sizeof(int)));
    HIP_TRY(hipMemsetAsync(c->tr_steps, 0, 2 * sizeof(int), (hipStream_t)stream));
    c->tr_windows = max_windows;
    {   // phased step (km_trainp.hip): packed input first, then a fixed part, then per-window activations; dropout masks
        int64_t fixed = 0;
        const int64_t per = trainp_act_floats(c, &fixed);
        const int64_t KP = trainp_kp(c);
        c->trp_act_floats = max_windows * (per + 2 * KP * c->NK) + fixed + 4096;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->trp_act), (size_t)c->trp_act_floats * sizeof(float)));
        // the packed input's columns beyond T + 3 are written by nobody when the front end packs: zeros (they meet zero weights)
        HIP_TRY(hipMemsetAsync(c->trp_act, 0, (size_t)c->trp_act_floats * sizeof(float), (hipStream_t)stream));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->trp_wcep), (size_t)c->d * KP * sizeof(float)));
        // split-K partials: up to 16 partial outputs of every weight / bias gradient that is a product over the rows of the batch
        c->trp_split_floats = 16 * (5 * (int64_t)c->d * c->d + 2 * (int64_t)c->DH * c->d + 16 * (int64_t)c->d + (int64_t)c->d * c->KT) + 1024;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->trp_split), (size_t)c->trp_split_floats * sizeof(float)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->trp_tail_part), (size_t)1024 * 64 * sizeof(float)));      // one row of sums per tail workgroup
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->trp_tail_ctr), sizeof(unsigned)));
        HIP_TRY(hipMemsetAsync(c->trp_tail_ctr, 0, sizeof(unsigned), (hipStream_t)stream));
        HIP_TRY(hipMalloc(&c->trp_masks, (size_t)trainp_mask_alloc_bytes(c)));
        HIP_TRY(hipMemsetAsync(c->trp_masks, 1, (size_t)trainp_mask_alloc_bytes(c), (hipStream_t)stream));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->trp_drop_ctr), sizeof(int)));
        HIP_TRY(hipMemsetAsync(c->trp_drop_ctr, 0, sizeof(int), (hipStream_t)stream));
    }
    return upload_train_params(c, stream);
}

int64_t km_train_num_params(km_handle h) { return h ? h->tr_nparams : -1; }

int64_t km_train_param_offset(km_handle h, const char* key) {
    if (!h || !key) return -1;
    auto it = h->tr_offset.find(key);
    return it == h->tr_offset.end() ? -1 : it->second;
}

static int need_train(Context* c, int64_t B) {
    if (int rc = need_dual(c)) return rc;
    if (!c->tr_params) return fail(KM_ERR_NOT_FINALIZED, "call km_train_init first");
    if (B <= 0 || B > c->tr_windows) return fail(KM_ERR_WORKSPACE, "km_train_init sized the step for %lld windows, got %lld",
                                                 (long long)c->tr_windows, (long long)B);
    return KM_OK;
}

int km_train_step(km_handle h, const float* mel_dev, int64_t B, int64_t T_in, const float* mel_short_dev,
                  const float* emotion_dev, const float* target_dev, float mse_weight, float l1_weight,
                  float* flat_grad_dev, float* loss_dev, float* out_dev, float* ema_state_dev, int32_t ema_first,
                  void* stream) {
    if (int rc = need_train(h, B)) return rc;
    if (!mel_dev || !mel_short_dev || !emotion_dev || !target_dev || !flat_grad_dev || !loss_dev || T_in <= 0)
        return fail(KM_ERR_INVALID_ARG, "km_train_step: bad argument");
    if (!h->opt.train_chain)
        return train_forward_backward_phased(h, mel_dev, B, T_in, mel_short_dev, nullptr, nullptr, emotion_dev, target_dev, mse_weight,
                                             l1_weight, flat_grad_dev, loss_dev, out_dev, ema_state_dev, ema_first, stream);
    if (h->tr_dropout_p > 0.f) return fail(KM_ERR_UNSUPPORTED, "the launch-per-op training chain has no dropout: unset train_chain");
    return train_forward_backward(h, mel_dev, B, T_in, mel_short_dev, emotion_dev, target_dev, mse_weight, l1_weight,
                                  flat_grad_dev, loss_dev, out_dev, ema_state_dev, ema_first, stream);
}

int km_train_step_audio(km_handle h, const float* audio_dev, int64_t B, int64_t L, const float* emotion_dev,
                        const float* target_dev, float mse_weight, float l1_weight, float* flat_grad_dev,
                        float* loss_dev, float* out_dev, float* ema_state_dev, int32_t ema_first, void* stream) {
    if (int rc = need_train(h, B)) return rc;
    Context* c = h;
    if (!audio_dev || !emotion_dev || !target_dev || !flat_grad_dev || !loss_dev || L <= 0)
        return fail(KM_ERR_INVALID_ARG, "km_train_step_audio: bad argument");
    const int64_t n_frames = 1 + L / c->cfg.mel.hop_length;
    if (B > c->ws_windows || n_frames > c->ws_frames)
        return fail(KM_ERR_WORKSPACE, "workspace too small for %lld windows x %lld samples: call km_reserve", (long long)B, (long long)L);
    if (!c->opt.train_chain) {
        // front end -> power-mel; phase 0 of the program converts and packs it into the encoder input (B, KP, n_mels) at the
        // head of the phased workspace
        // (round 4: the front end writes the packed dB input itself -- MelPack -- where it can; option train_no_fe_pack)
        const bool fe_packs = !c->opt.train_no_fe_pack && !c->opt.train_no_dma &&
                              mel_packs(c, c->mel_plans[0], n_frames, c->T) && c->mel_plans[0]->cfg.n_mels == c->NK;
        const MelPack pack{c->trp_act, (int)c->T, (int)trainp_kp(c)};
        if (int rc = launch_mel_power(c, c->mel_plans[0], audio_dev, B, L, stream, 0, 0, 0, 1, nullptr, nullptr, nullptr, nullptr, nullptr,
                                      fe_packs ? &pack : nullptr)) return rc;
        c->melmax_dirty = true;      // until phase 1 / 2 has re-zeroed the maxima
        const LogParams lp = plan_log_params(c->mel_plans[0]);
        const TrainAudioSrc asrc{c->ws_melpow, c->ws_melmax, (int)n_frames, &lp, fe_packs};
        return train_forward_backward_phased(c, nullptr, B, n_frames, nullptr, c->trp_act, &asrc, emotion_dev, target_dev, mse_weight,
                                             l1_weight, flat_grad_dev, loss_dev, out_dev, ema_state_dev, ema_first, stream);
    }
    if (c->tr_dropout_p > 0.f) return fail(KM_ERR_UNSUPPORTED, "the launch-per-op training chain has no dropout: unset train_chain");
    if (int rc = launch_mel(c, c->mel_plans[0], audio_dev, B, L, 0, c->ws_mel, c->ws_short, stream)) return rc;
    return train_forward_backward(c, c->ws_mel, B, n_frames, c->ws_short, emotion_dev, target_dev, mse_weight, l1_weight,
                                  flat_grad_dev, loss_dev, out_dev, ema_state_dev, ema_first, stream);
}

int km_linear(const float* x_dev, const float* w_dev, const float* b_dev, int64_t B, int64_t K, int64_t N, float* out_dev, void* stream) {
    if (!x_dev || !w_dev || !out_dev || B <= 0 || K <= 0 || N <= 0) return fail(KM_ERR_INVALID_ARG, "km_linear: bad argument");
    GemmArgs g{};
    g.alpha = 1.f; g.batch2 = 1; g.kb_count = 1;
    g.A = x_dev; g.a_rs = K; g.a_cs = 1; g.B = w_dev; g.b_rs = 1; g.b_cs = K; g.C = out_dev; g.c_rs = N;
    g.M = (int)B; g.N = (int)N; g.K = (int)K; g.bias = b_dev; g.bias_mode = b_dev ? 1 : 0;
    return launch_gemm(g, 1, stream);
}

int km_train_grad_split(km_handle h, int64_t* early_floats) {
    if (int rc = need_train(h, 1)) return rc;
    if (!early_floats) return fail(KM_ERR_INVALID_ARG, "km_train_grad_split: NULL argument");
    *early_floats = h->opt.train_chain ? h->tr_nparams : h->tr_early;      // the chain finishes everything at its last join
    return KM_OK;
}

int km_train_wait_early(km_handle h, void* stream) {
    if (int rc = need_train(h, 1)) return rc;
    if (!h->tr_early_recorded) return fail(KM_ERR_NOT_READY, "km_train_wait_early: no phased training step has run yet");
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)h->tr_ev[0], 0));
    return KM_OK;
}

int km_train_set_dropout(km_handle h, float p, uint64_t seed, int32_t external_masks) {
    if (int rc = need_train(h, 1)) return rc;
    if (!(p >= 0.f && p < 1.f)) return fail(KM_ERR_INVALID_ARG, "dropout probability has to be in [0, 1), but got %g", (double)p);
    h->tr_dropout_p = p; h->tr_dropout_seed = seed; h->tr_dropout_mode = external_masks ? 1 : 0;
    return KM_OK;
}

int km_train_get_dropout_step(km_handle h, int64_t* step) {
    if (int rc = need_train(h, 1)) return rc;
    if (!step) return fail(KM_ERR_INVALID_ARG, "km_train_get_dropout_step: NULL argument");
    int v = 0;
    if (h->trp_drop_ctr) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&v, h->trp_drop_ctr, sizeof(int), hipMemcpyDeviceToHost));
    }
    *step = v;
    return KM_OK;
}

int km_train_set_dropout_step(km_handle h, int64_t step) {
    if (int rc = need_train(h, 1)) return rc;
    if (step < 0 || step > 0x7fffffff) return fail(KM_ERR_INVALID_ARG, "km_train_set_dropout_step: step out of range");
    if (!h->trp_drop_ctr) return fail(KM_ERR_NOT_READY, "km_train_set_dropout_step: the phased training step is not initialised");
    const int v = (int)step;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->trp_drop_ctr, &v, sizeof(int), hipMemcpyHostToDevice));
    return KM_OK;
}

int km_train_get_dropout_masks(km_handle h, int64_t B, uint8_t* mel_host, uint8_t* emo_host, uint8_t* dec_host, void* stream) {
    if (int rc = need_train(h, B)) return rc;
    if (!mel_host || !emo_host || !dec_host) return fail(KM_ERR_INVALID_ARG, "km_train_get_dropout_masks: NULL argument");
    return trainp_copy_masks(h, B, mel_host, emo_host, dec_host, 0, stream);
}

int km_train_set_dropout_masks(km_handle h, int64_t B, const uint8_t* mel_host, const uint8_t* emo_host, const uint8_t* dec_host,
                               void* stream) {
    if (int rc = need_train(h, B)) return rc;
    if (!mel_host || !emo_host || !dec_host) return fail(KM_ERR_INVALID_ARG, "km_train_set_dropout_masks: NULL argument");
    return trainp_copy_masks(h, B, const_cast<uint8_t*>(mel_host), const_cast<uint8_t*>(emo_host), const_cast<uint8_t*>(dec_host), 1, stream);
}

int km_audio_energy(const float* features_dev, int64_t B, int64_t T, int64_t D, float* energy_dev, void* stream) {
    if (!features_dev || !energy_dev || B <= 0 || T <= 0 || D <= 0) return fail(KM_ERR_INVALID_ARG, "km_audio_energy: bad argument");
    return launch_audio_energy(features_dev, B, T, D, energy_dev, stream);
}

int km_train_set_loss(km_handle h, const km_loss_config* cfg) {
    if (int rc = need_train(h, 1)) return rc;
    Context* c = h;
    // the struct has grown since ABI version 1 and holds device pointers the loss tail dereferences: a caller built against
    // an older header must be refused, not read past
    if (cfg && cfg->abi_version != KM_ABI_VERSION)
        return fail(KM_ERR_INVALID_ARG, "km_loss_config.abi_version %d != %d", cfg->abi_version, KM_ABI_VERSION);
    if (cfg) c->tr_loss_cfg = *cfg; else c->tr_loss_cfg = km_loss_config{};
    return KM_OK;
}

int km_train_adamw(km_handle h, const float* flat_grad_dev, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float max_grad_norm, int64_t step, void* stream) {
    if (int rc = need_train(h, 1)) return rc;
    if (!flat_grad_dev || step < 1) return fail(KM_ERR_INVALID_ARG, "km_train_adamw: bad argument");
    return train_adamw(h, flat_grad_dev, lr, beta1, beta2, eps, weight_decay, max_grad_norm, step, stream);
}

int km_train_get_params(km_handle h, float* flat_host, int64_t n) {
    if (int rc = need_train(h, 1)) return rc;
    if (!flat_host || n != h->tr_nparams) return fail(KM_ERR_INVALID_ARG, "km_train_get_params: size mismatch");
    HIP_TRY(hipMemcpy(flat_host, h->tr_params, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return KM_OK;
}

int km_train_set_params(km_handle h, const float* flat_host, int64_t n) {
    if (int rc = need_train(h, 1)) return rc;
    if (!flat_host || n != h->tr_nparams) return fail(KM_ERR_INVALID_ARG, "km_train_set_params: size mismatch");
    HIP_TRY(hipMemcpy(h->tr_params, flat_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    if (int rc = train_refresh_padded_weights(h, nullptr)) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    return KM_OK;
}

int km_train_get_optimizer_state(km_handle h, float* exp_avg_host, float* exp_avg_sq_host, int64_t n, int32_t* steps2_host) {
    if (int rc = need_train(h, 1)) return rc;
    if (!exp_avg_host || !exp_avg_sq_host || !steps2_host || n != h->tr_nparams)
        return fail(KM_ERR_INVALID_ARG, "km_train_get_optimizer_state: bad argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(exp_avg_host, h->tr_m, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(exp_avg_sq_host, h->tr_v, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(steps2_host, h->tr_steps, 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return KM_OK;
}

int km_train_set_optimizer_state(km_handle h, const float* exp_avg_host, const float* exp_avg_sq_host, int64_t n,
                                 const int32_t* steps2_host) {
    if (int rc = need_train(h, 1)) return rc;
    if (!exp_avg_host || !exp_avg_sq_host || !steps2_host || n != h->tr_nparams)
        return fail(KM_ERR_INVALID_ARG, "km_train_set_optimizer_state: bad argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->tr_m, exp_avg_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->tr_v, exp_avg_sq_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->tr_steps, steps2_host, 2 * sizeof(int32_t), hipMemcpyHostToDevice));
    return KM_OK;
}

int km_train_sync(km_handle h, void* stream) {
    if (int rc = need_train(h, 1)) return rc;
    Context* c = h;
    std::vector<float> flat((size_t)c->tr_nparams);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(flat.data(), c->tr_params, flat.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (const auto& k : c->param_order) {
        HostParam& hp = c->params.at(k);
        std::memcpy(hp.data.data(), flat.data() + c->tr_offset.at(k), hp.data.size() * sizeof(float));
    }
    c->host_finalized = false;
    return km_finalize(h, stream);
}

int km_legacy_forward_mel(km_handle h, const float* mel_dev, int64_t B, int64_t T_mel, float* out_dev, void* stream) {
    if (int rc = need_ready(h)) return rc;
    Context* c = h;
    if (c->kind != 1) return fail(KM_ERR_INVALID_ARG, "not a legacy handle (km_legacy_create)");
    if (!mel_dev || !out_dev || B <= 0 || T_mel <= 0) return fail(KM_ERR_INVALID_ARG, "km_legacy_forward_mel: bad argument");
    if (B > c->ws_windows || T_mel > c->ws_frames || !c->ws_generic)
        return fail(KM_ERR_WORKSPACE, "workspace too small for %lld windows x %lld frames: call km_reserve",
                    (long long)B, (long long)T_mel);
    return launch_legacy(c, mel_dev, B, T_mel, out_dev, stream);
}

int km_koemorph_reserve(km_handle h, int64_t max_batch, int64_t max_frames) {
    if (int rc = need_ready(h)) return rc;
    Context* c = h;
    if (c->kind != 2) return fail(KM_ERR_INVALID_ARG, "not a KoeMorphModel handle (km_koemorph_create)");
    if (max_batch <= 0 || max_frames <= 0) return fail(KM_ERR_INVALID_ARG, "km_koemorph_reserve: bad argument");
    if (max_batch <= c->kmm_batch && max_frames <= c->kmm_frames) return KM_OK;
    const int64_t Bm = max_batch > c->kmm_batch ? max_batch : c->kmm_batch, Tm = max_frames > c->kmm_frames ? max_frames : c->kmm_frames;
    if (c->ws_generic) { HIP_TRY(hipFree(c->ws_generic)); c->ws_generic = nullptr; c->kmm_batch = c->kmm_frames = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_generic), (size_t)(Bm * koemorph_ws_floats(c, Tm)) * sizeof(float)));
    c->kmm_batch = Bm; c->kmm_frames = Tm;
    return KM_OK;
}

int km_koemorph_forward(km_handle h, const float* mel_dev, const float* emotion_dev, int64_t B, int64_t T,
                        const uint8_t* audio_mask_dev, const float* prev_dev, float* smoother_state_dev,
                        int32_t apply_constraints, float* out_dev, float* raw_dev, float* attn_dev, void* stream) {
    if (int rc = need_ready(h)) return rc;
    Context* c = h;
    if (c->kind != 2) return fail(KM_ERR_INVALID_ARG, "not a KoeMorphModel handle (km_koemorph_create)");
    if (!mel_dev || !emotion_dev || !out_dev || B <= 0 || T <= 0) return fail(KM_ERR_INVALID_ARG, "km_koemorph_forward: bad argument");
    // the workspace is carved per call for (B, T): B * ws(T) floats must fit what km_koemorph_reserve allocated
    if (!c->ws_generic || B * koemorph_ws_floats(c, T) > c->kmm_batch * koemorph_ws_floats(c, c->kmm_frames))
        return fail(KM_ERR_WORKSPACE, "workspace too small for %lld x %lld frames: call km_koemorph_reserve", (long long)B, (long long)T);
    return launch_koemorph(c, mel_dev, emotion_dev, B, T, audio_mask_dev, prev_dev, smoother_state_dev, apply_constraints, out_dev, raw_dev,
                           attn_dev, stream);
}

int km_legacy_forward(km_handle h, const float* audio_dev, int64_t B, int64_t L, float* out_dev, void* stream) {
    if (int rc = need_ready(h)) return rc;
    Context* c = h;
    if (c->kind != 1) return fail(KM_ERR_INVALID_ARG, "not a legacy handle (km_legacy_create)");
    if (!audio_dev || !out_dev || B <= 0 || L <= 0) return fail(KM_ERR_INVALID_ARG, "km_legacy_forward: bad argument");
    const int64_t n_frames = 1 + L / c->cfg.mel.hop_length;
    if (B > c->ws_windows || n_frames > c->ws_frames || !c->ws_generic)
        return fail(KM_ERR_WORKSPACE, "workspace too small for %lld windows x %lld samples: call km_reserve",
                    (long long)B, (long long)L);
    if (legacy_pow_ok(c)) {
        // front end -> power-mel + window maxima; the fused encoder converts to dB while it stages its rows and the attention
        // kernel re-zeroes the maxima: no mel_log_kernel launch, no memset (round 4; 18 us of the 460 per 256 windows)
        if (int rc = launch_mel_power(c, c->mel_plans[0], audio_dev, B, L, stream)) return rc;
        const LogParams lp = plan_log_params(c->mel_plans[0]);
        const LegacyPowSrc src{c->ws_melpow, c->ws_melmax, &lp};
        return launch_legacy(c, nullptr, B, n_frames, out_dev, stream, &src);
    }
    if (int rc = launch_mel(c, c->mel_plans[0], audio_dev, B, L, 0, c->ws_mel, nullptr, stream)) return rc;
    return launch_legacy(c, c->ws_mel, B, n_frames, out_dev, stream);
}

int km_ema_scan(km_handle h, float* x_dev, int64_t B, int64_t N, void* stream) {
    if (int rc = need_ready(h)) return rc;
    if (h->kind != 0) return fail(KM_ERR_INVALID_ARG, "this entry point needs a dual-stream handle (km_create)");
    if (!x_dev || B <= 0 || N <= 0) return fail(KM_ERR_INVALID_ARG, "km_ema_scan: bad argument");
    return launch_ema_scan(h, x_dev, B, N, stream);
}

int km_smooth(km_handle h, float* x_dev, float* state_dev, int64_t B, int32_t first, void* stream) {
    if (int rc = need_ready(h)) return rc;
    if (!x_dev || !state_dev || B <= 0) return fail(KM_ERR_INVALID_ARG, "km_smooth: bad argument");
    return launch_smooth(h, x_dev, state_dev, B, first, stream);
}

int km_forward_audio(km_handle h, const float* audio_dev, int64_t B, int64_t L, const float* emotion_dev,
                     float* out_dev, float* state_dev, int32_t first, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (!audio_dev || !emotion_dev || !out_dev || B <= 0 || L <= 0)
        return fail(KM_ERR_INVALID_ARG, "km_forward_audio: bad argument");
    const int64_t n_frames = 1 + L / c->cfg.mel.hop_length;
    if (B > c->ws_windows || n_frames > c->ws_frames)
        return fail(KM_ERR_WORKSPACE, "workspace too small for %lld windows x %lld samples: call km_reserve",
                    (long long)B, (long long)L);
    if (!c->fused_ok) {     // generic shapes: staged front end, GEMM-chain core, stand-alone EMA
        if (!c->ws_generic) return fail(KM_ERR_WORKSPACE, "generic workspace missing: call km_reserve after km_finalize");
        const bool power_path = generic_core_takes_power(c) && !c->opt.generic_staged;
        const bool fuse_emo = power_path && mel_fuses_emotion(c, c->mel_plans[0]);   // emotion logits inside the front-end kernel
        if (!fuse_emo)
            if (int rc = launch_emotion(c, emotion_dev, B, c->ws_zemo, stream)) return rc;
        if (power_path) {
            // power-mel -> dB conversion inside the encoder's tile staging: no log-mel image at all
            if (int rc = launch_mel_power(c, c->mel_plans[0], audio_dev, B, L, stream, 0, 0, 0, 1, nullptr, nullptr,
                                          fuse_emo ? emotion_dev : nullptr, fuse_emo ? c->ws_zemo : nullptr)) return rc;
            if (int rc = launch_core_generic_power(c, c->mel_plans[0], B, n_frames, c->ws_zemo, out_dev, nullptr,
                                                   nullptr, stream)) return rc;
        } else if (c->NK == 80 && !c->opt.generic_staged) {
            // log-mel written straight into the packed encoder input; long + short-term rows in one contraction
            float* xp = generic_packed_x(c, B);
            if (int rc = launch_mel_packed(c, c->mel_plans[0], audio_dev, B, L, xp, c->T, (c->KT + 15) / 16 * 16, stream)) return rc;
            if (int rc = launch_core_generic_packed(c, xp, B, c->ws_zemo, out_dev, nullptr, nullptr, stream)) return rc;
        } else {
            if (int rc = launch_mel(c, c->mel_plans[0], audio_dev, B, L, 0, c->ws_mel, c->ws_short, stream)) return rc;
            if (int rc = launch_core_generic(c, c->ws_mel, B, n_frames, c->ws_short, c->ws_zemo, out_dev, nullptr, nullptr, stream)) return rc;
        }
        if (state_dev) return launch_smooth(c, out_dev, state_dev, B, first, stream);
        return KM_OK;
    }
    // three launches: emotion logits, power-mel + window maxima, fused core (dB conversion on load)
    hipStream_t st = (hipStream_t)stream;
    const bool tm = c->stage_timing;
    const bool fuse_emo = mel_fuses_emotion(c, c->mel_plans[0]);   // emotion logits computed inside the front-end kernel
    if (tm) HIP_TRY(hipEventRecord((hipEvent_t)c->stage_ev[0], st));
    if (!fuse_emo)
        if (int rc = launch_emotion(c, emotion_dev, B, c->ws_zemo, stream)) return rc;
    if (tm) HIP_TRY(hipEventRecord((hipEvent_t)c->stage_ev[1], st));
    if (int rc = launch_mel_power(c, c->mel_plans[0], audio_dev, B, L, stream, 0, 0, 0, 1, nullptr, nullptr,
                                  fuse_emo ? emotion_dev : nullptr, fuse_emo ? c->ws_zemo : nullptr)) return rc;
    if (tm) { HIP_TRY(hipEventRecord((hipEvent_t)c->stage_ev[2], st)); HIP_TRY(hipEventRecord((hipEvent_t)c->stage_ev[3], st)); }
    if (int rc = launch_core_fused_db(c, c->mel_plans[0], B, n_frames, c->ws_zemo, out_dev, state_dev, first, stream)) return rc;
    if (tm) HIP_TRY(hipEventRecord((hipEvent_t)c->stage_ev[4], st));
    return KM_OK;
}

static int free_streams(Context* c) {
    void* ptrs[] = {c->ring, c->ring_wptr, c->ring_frames, c->ring_ready, c->ring_started, c->ring_state};
    for (void* p : ptrs)
        if (p) HIP_TRY(hipFree(p));
    c->ring = nullptr; c->ring_wptr = nullptr; c->ring_frames = nullptr; c->ring_ready = nullptr;
    c->ring_started = nullptr; c->ring_state = nullptr; c->n_streams = 0;
    return KM_OK;
}

int km_stream_reset(km_handle h, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (c->n_streams <= 0) return fail(KM_ERR_INVALID_ARG, "km_stream_reset: no streams (km_stream_create first)");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(c->ring, 0, (size_t)c->n_streams * c->ring_len * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(c->ring_wptr, 0, (size_t)c->n_streams * sizeof(int), st));
    HIP_TRY(hipMemsetAsync(c->ring_frames, 0, (size_t)c->n_streams * sizeof(int), st));
    HIP_TRY(hipMemsetAsync(c->ring_ready, 0, (size_t)c->n_streams, st));
    HIP_TRY(hipMemsetAsync(c->ring_started, 0, (size_t)c->n_streams, st));
    HIP_TRY(hipMemsetAsync(c->ring_state, 0, (size_t)c->n_streams * c->NB * sizeof(float), st));
    return KM_OK;
}

int km_stream_create(km_handle h, int64_t n_streams, double context_window_s, double update_interval_s,
                     const km_mel_config* mel_cfg) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (n_streams <= 0 || !(context_window_s > 0) || !(update_interval_s > 0) || !mel_cfg)
        return fail(KM_ERR_INVALID_ARG, "km_stream_create: bad argument");
    if (mel_cfg->n_fft != 512 && mel_cfg->n_fft != 1024) return fail(KM_ERR_UNSUPPORTED, "n_fft must be 512 or 1024");
    if (mel_cfg->n_mels != c->NK) return fail(KM_ERR_INVALID_ARG, "stream front end must produce %d mel channels", c->NK);
    if (int rc = free_streams(c)) return rc;
    const int sr = mel_cfg->sample_rate;
    c->ring_len = (int64_t)(context_window_s * sr);                                // mel_sliding_window.py:46
    const double target_fps = 1.0 / update_interval_s;
    c->ring_hop = (int)((double)sr / target_fps);                                   // :49-50 (532 for 0.0333)
    c->stream_out_frames = (int64_t)(context_window_s / update_interval_s);         // :300
    if (c->ring_hop <= 0 || c->ring_len <= mel_cfg->n_fft / 2) return fail(KM_ERR_INVALID_ARG, "context window too short");
    c->n_streams = n_streams;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ring), (size_t)n_streams * c->ring_len * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ring_wptr), (size_t)n_streams * sizeof(int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ring_frames), (size_t)n_streams * sizeof(int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ring_ready), (size_t)n_streams));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ring_started), (size_t)n_streams));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ring_state), (size_t)n_streams * c->NB * sizeof(float)));
    c->stream_plan = find_or_add_plan(c, *mel_cfg);
    if (int rc = upload_mel_plan(c->stream_plan)) return rc;
    if (int rc = km_reserve(h, n_streams, c->ring_len)) return rc;
    if (int rc = km_stream_reset(h, nullptr)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return KM_OK;
}

int km_stream_push(km_handle h, const float* samples_dev, int64_t n_per_stream, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (c->n_streams <= 0) return fail(KM_ERR_INVALID_ARG, "km_stream_push: no streams (km_stream_create first)");
    if (!samples_dev) return fail(KM_ERR_INVALID_ARG, "km_stream_push: NULL samples");
    const int64_t diff = n_per_stream - c->ring_hop;
    if (diff > 1 || diff < -1)                                                      // :80-82
        return fail(KM_ERR_INVALID_ARG, "Frame size mismatch: expected ~%d, got %lld", c->ring_hop, (long long)n_per_stream);
    return launch_ring_push(c, samples_dev, n_per_stream, stream);
}

int km_stream_tick(km_handle h, const float* emotion_dev, float* out_dev, uint8_t* ready_dev, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (c->n_streams <= 0) return fail(KM_ERR_INVALID_ARG, "km_stream_tick: no streams (km_stream_create first)");
    if (!emotion_dev || !out_dev) return fail(KM_ERR_INVALID_ARG, "km_stream_tick: NULL argument");
    if (!c->fused_ok)
        return fail(KM_ERR_UNSUPPORTED, "no kernel for d_model=%d, mel_sequence_length=%d, heads=%d", c->d, c->T, c->H);
    const int64_t S = c->n_streams, L = c->ring_len;
    const int64_t n_frames = 1 + L / c->stream_plan->cfg.hop_length;
    if (S > c->ws_windows || n_frames > c->ws_frames) return fail(KM_ERR_WORKSPACE, "stream workspace too small");
    const bool fuse_emo = mel_fuses_emotion(c, c->stream_plan);
    if (!fuse_emo)
        if (int rc = launch_emotion(c, emotion_dev, S, c->ws_zemo, stream)) return rc;
    if (int rc = launch_mel_power(c, c->stream_plan, c->ring, S, L, stream, L, 0, 0, 1, c->ring_wptr, c->ring_ready,
                                  fuse_emo ? emotion_dev : nullptr, fuse_emo ? c->ws_zemo : nullptr)) return rc;
    if (int rc = launch_core_fused_db(c, c->stream_plan, S, n_frames, c->ws_zemo, out_dev, c->ring_state, 0, stream, 0, 1,
                                      c->stream_out_frames, c->ring_ready, c->ring_started)) return rc;
    if (ready_dev)
        HIP_TRY(hipMemcpyAsync(ready_dev, c->ring_ready, (size_t)S, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return KM_OK;
}

int km_enable_stage_timing(km_handle h, int32_t enable) {
    if (!h) return fail(KM_ERR_INVALID_ARG, "NULL handle");
    Context* c = h;
    if (enable && !c->stage_ev[0])
        for (int i = 0; i < 5; ++i) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            c->stage_ev[i] = e;
        }
    c->stage_timing = enable != 0;
    return KM_OK;
}

int km_stage_times(km_handle h, float* ms3) {
    if (!h || !ms3) return fail(KM_ERR_INVALID_ARG, "km_stage_times: NULL argument");
    Context* c = h;
    if (!c->stage_ev[0]) return fail(KM_ERR_INVALID_ARG, "stage timing was never enabled");
    HIP_TRY(hipEventSynchronize((hipEvent_t)c->stage_ev[4]));
    HIP_TRY(hipEventElapsedTime(&ms3[0], (hipEvent_t)c->stage_ev[0], (hipEvent_t)c->stage_ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms3[1], (hipEvent_t)c->stage_ev[1], (hipEvent_t)c->stage_ev[2]));
    HIP_TRY(hipEventElapsedTime(&ms3[2], (hipEvent_t)c->stage_ev[3], (hipEvent_t)c->stage_ev[4]));
    return KM_OK;
}

static int free_pipeline(Context* c) {
    for (int i = 0; i < 2; ++i) {
        if (c->pipe_melpow[i]) HIP_TRY(hipFree(c->pipe_melpow[i]));
        if (c->pipe_melmax[i]) HIP_TRY(hipFree(c->pipe_melmax[i]));
        if (c->pipe_zemo[i]) HIP_TRY(hipFree(c->pipe_zemo[i]));
        c->pipe_melpow[i] = nullptr; c->pipe_melmax[i] = nullptr; c->pipe_zemo[i] = nullptr; c->pipe_dirty[i] = true;
    }
    c->pipe_windows = c->pipe_frames = 0;
    return KM_OK;
}

static int ensure_pipeline(Context* c, int64_t B, int64_t n_frames) {
    if (!c->pipe_s1) {
        hipStream_t s;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); c->pipe_s1 = s;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); c->pipe_s2 = s;
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->pipe_ev_in[i] = e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->pipe_ev_mel[i] = e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->pipe_ev_core[i] = e;
        }
    }
    if (B <= c->pipe_windows && n_frames <= c->pipe_frames) return KM_OK;
    HIP_TRY(hipDeviceSynchronize());                       // growing the slots: nothing may be in flight
    if (int rc = free_pipeline(c)) return rc;
    const int64_t W = B, F = n_frames;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->pipe_melpow[i]), (size_t)W * F * c->NK * sizeof(float)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->pipe_melmax[i]), (size_t)W * sizeof(unsigned)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->pipe_zemo[i]), (size_t)W * sizeof(float)));
    }
    c->pipe_windows = W; c->pipe_frames = F; c->pipe_seq = 0;
    return KM_OK;
}

int km_forward_audio_pipelined(km_handle h, const float* audio_dev, int64_t B, int64_t L, const float* emotion_dev,
                               float* out_dev, float* state_dev, int32_t first, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (!audio_dev || !emotion_dev || !out_dev || B <= 0 || L <= 0)
        return fail(KM_ERR_INVALID_ARG, "km_forward_audio_pipelined: bad argument");
    if (!c->fused_ok) return fail(KM_ERR_UNSUPPORTED, "the pipelined mode needs the fused kernel shape (d_model 256, window 256, 8 heads)");
    const int64_t n_frames = 1 + L / c->cfg.mel.hop_length;
    if (B > c->ws_windows || n_frames > c->ws_frames)
        return fail(KM_ERR_WORKSPACE, "workspace too small for %lld windows x %lld samples: call km_reserve", (long long)B, (long long)L);
    if (int rc = ensure_pipeline(c, B > c->ws_windows ? B : c->ws_windows, n_frames > c->ws_frames ? n_frames : c->ws_frames)) return rc;
    const int slot = (int)(c->pipe_seq & 1);
    hipStream_t caller = (hipStream_t)stream, s1 = (hipStream_t)c->pipe_s1, s2 = (hipStream_t)c->pipe_s2;
    const bool tm = c->stage_timing;
    // redirect the workspace pointers the launchers read to this call's slot
    float* keep_pow = c->ws_melpow; unsigned* keep_max = c->ws_melmax; float* keep_z = c->ws_zemo; const bool keep_dirty = c->melmax_dirty;
    c->ws_melpow = c->pipe_melpow[slot]; c->ws_melmax = c->pipe_melmax[slot]; c->ws_zemo = c->pipe_zemo[slot];
    c->melmax_dirty = c->pipe_dirty[slot];
    int rc = KM_OK;
    do {
        if (hipEventRecord((hipEvent_t)c->pipe_ev_in[slot], caller) != hipSuccess) { rc = fail(KM_ERR_HIP, "hipEventRecord failed"); break; }
        if (hipStreamWaitEvent(s1, (hipEvent_t)c->pipe_ev_in[slot], 0) != hipSuccess) { rc = fail(KM_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        if (tm) (void)hipEventRecord((hipEvent_t)c->stage_ev[0], s1);
        const bool fuse_emo = mel_fuses_emotion(c, c->mel_plans[0]);
        if (!fuse_emo && (rc = launch_emotion(c, emotion_dev, B, c->ws_zemo, s1))) break;
        if (tm) (void)hipEventRecord((hipEvent_t)c->stage_ev[1], s1);
        if ((rc = launch_mel_power(c, c->mel_plans[0], audio_dev, B, L, s1, 0, 0, 0, 1, nullptr, nullptr,
                                   fuse_emo ? emotion_dev : nullptr, fuse_emo ? c->ws_zemo : nullptr))) break;
        if (tm) (void)hipEventRecord((hipEvent_t)c->stage_ev[2], s1);
        (void)hipEventRecord((hipEvent_t)c->pipe_ev_mel[slot], s1);
        (void)hipStreamWaitEvent(s2, (hipEvent_t)c->pipe_ev_mel[slot], 0);
        if (tm) (void)hipEventRecord((hipEvent_t)c->stage_ev[3], s2);
        if ((rc = launch_core_fused_db(c, c->mel_plans[0], B, n_frames, c->ws_zemo, out_dev, state_dev, first, s2))) break;
        if (tm) (void)hipEventRecord((hipEvent_t)c->stage_ev[4], s2);
        (void)hipEventRecord((hipEvent_t)c->pipe_ev_core[slot], s2);
        // stream-order visibility of the PREVIOUS call's result (and release of its slot for the next call)
        if (c->pipe_seq > 0) (void)hipStreamWaitEvent(caller, (hipEvent_t)c->pipe_ev_core[slot ^ 1], 0);
    } while (0);
    c->pipe_dirty[slot] = c->melmax_dirty;
    c->ws_melpow = keep_pow; c->ws_melmax = keep_max; c->ws_zemo = keep_z; c->melmax_dirty = keep_dirty;
    if (rc == KM_OK) c->pipe_seq += 1;
    return rc;
}

int km_pipeline_flush(km_handle h, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (c->pipe_seq > 0 && c->pipe_s1)
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)c->pipe_ev_core[(c->pipe_seq - 1) & 1], 0));
    return KM_OK;
}

int64_t km_sequence_num_outputs(km_handle h, int64_t L, int32_t stride_frames) {
    if (!h || L < 0 || stride_frames <= 0) return -1;
    const int64_t num_frames = L / h->cfg.mel.hop_length;                       // sequential_dual_stream_model.py:84
    const int64_t n = (num_frames - h->T) / stride_frames + 1;                   // :96 (floor division)
    const int64_t nn = (num_frames - h->T) >= 0 ? n : ((num_frames - h->T - (stride_frames - 1)) / stride_frames + 1);
    return nn > 1 ? nn : 1;
}

int km_sequence_forward(km_handle h, const float* audio_dev, int64_t B, int64_t L, const float* emotion_dev,
                        int32_t stride_frames, int32_t smooth, float* out_dev, void* stream) {
    if (int rc = need_dual(h)) return rc;
    Context* c = h;
    if (!audio_dev || !emotion_dev || !out_dev || B <= 0 || L <= 0 || stride_frames <= 0)
        return fail(KM_ERR_INVALID_ARG, "km_sequence_forward: bad argument");
    if (!c->fused_ok && !c->ws_generic)
        return fail(KM_ERR_WORKSPACE, "generic workspace missing: call km_reserve after km_finalize");
    const int hop = c->cfg.mel.hop_length;
    const int64_t N = km_sequence_num_outputs(h, L, stride_frames);
    const int64_t W = (int64_t)c->T * hop;                 // window_samples (sequential_dual_stream_model.py:54)
    const int64_t step = (int64_t)stride_frames * hop;     // stride_samples (:55)
    const int64_t n_frames = 1 + W / hop;
    if (c->ws_windows < B || c->ws_windows < 1 || n_frames > c->ws_frames)
        return fail(KM_ERR_WORKSPACE, "workspace too small for %lld clips / %lld-frame windows: call km_reserve",
                    (long long)B, (long long)n_frames);
    if (N > 0x7fffffff) return fail(KM_ERR_INVALID_ARG, "too many output frames");
    // emotion features ONCE for the entire audio (:88): one logit per clip
    if (int rc = launch_emotion(c, emotion_dev, B, c->ws_zemo, stream)) return rc;
    const int64_t total = B * N, tile = c->ws_windows;
    // Shared-frame path.  Windows start at multiples of the hop (:101-117), so frame f of window i IS clip frame
    // i*stride + f for f = 1 .. T-1; only frame 0 and frame T see the zero padding at the window boundary.  The STFT
    // of the clip is computed once (N-1)*stride + T + 1 frames instead of N * (T+1)), the two edge frames per window
    // separately, and the core reads its rows from both images.  Results are bit-identical to the per-window path.
    const bool dedup = !c->opt.seq_per_window;       // km_set_option: tests compare both paths
    if (c->fused_ok && dedup && c->cfg.mel.n_fft == 1024 && !c->opt.mel_two_frame && N < (1 << 24)) {
        hipStream_t st = (hipStream_t)stream;
        const int64_t nfc = (N - 1) * stride_frames + n_frames;                 // clip frames any window touches
        if (B * nfc > c->seq_pow_cap) {                                           // grow-only; may allocate (not capturable)
            if (c->seq_pow) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(c->seq_pow)); HIP_TRY(hipFree(c->seq_fmax)); }
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->seq_pow), (size_t)B * nfc * c->NK * sizeof(float)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->seq_fmax), (size_t)B * nfc * sizeof(unsigned)));
            c->seq_pow_cap = B * nfc;
        }
        if (total > c->seq_edge_cap) {
            if (c->seq_edge) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(c->seq_edge)); HIP_TRY(hipFree(c->seq_emax)); }
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->seq_edge), (size_t)total * 2 * c->NK * sizeof(float)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->seq_emax), (size_t)total * 2 * sizeof(unsigned)));
            c->seq_edge_cap = total;
        }
        HIP_TRY(hipMemsetAsync(c->seq_fmax, 0, (size_t)B * nfc * sizeof(unsigned), st));
        HIP_TRY(hipMemsetAsync(c->seq_emax, 0, (size_t)total * 2 * sizeof(unsigned), st));
        // (1) every clip as one long "window" of nfc frames, zero beyond the clip end
        SeqFrames clip_img{c->seq_pow, c->seq_fmax, nfc, 1};
        if (int rc = launch_mel_power(c, c->mel_plans[0], audio_dev, B, (nfc - 1) * hop, stream, L, 0, 0, 1, nullptr, nullptr,
                                      nullptr, nullptr, &clip_img)) return rc;
        // (2) the first and last frame of every window (rows 0 and 1 = frames 0 and T of the window)
        SeqFrames edge_img{c->seq_edge, c->seq_emax, 2, (int)(n_frames - 1)};
        if (int rc = launch_mel_power(c, c->mel_plans[0], audio_dev, total, W, stream, L, step, 0, (int)N, nullptr, nullptr,
                                      nullptr, nullptr, &edge_img)) return rc;
        // (3) per tile: window maxima, then the fused core reading rows from both images
        SeqCore sc{c->seq_pow, c->seq_edge, (int)nfc, (int)stride_frames, (int)N};
        for (int64_t w0 = 0; w0 < total; w0 += tile) {
            const int64_t nw = (total - w0) < tile ? (total - w0) : tile;
            if (int rc = launch_seq_window_max(c, c->seq_fmax, c->seq_emax, nw, w0, (int)nfc, (int)stride_frames, (int)N,
                                               (int)n_frames, stream)) return rc;
            if (int rc = launch_core_fused_db(c, c->mel_plans[0], nw, n_frames, c->ws_zemo, out_dev + w0 * c->NB, nullptr, 1,
                                              stream, w0, (int)N, 0, nullptr, nullptr, &sc)) return rc;
        }
        if (smooth) return launch_ema_scan(c, out_dev, B, N, stream);
        return KM_OK;
    }
    for (int64_t w0 = 0; w0 < total; w0 += tile) {
        const int64_t nw = (total - w0) < tile ? (total - w0) : tile;
        if (c->fused_ok) {
            if (int rc = launch_mel_power(c, c->mel_plans[0], audio_dev, nw, W, stream, L, step, w0, (int)N)) return rc;
            if (int rc = launch_core_fused_db(c, c->mel_plans[0], nw, n_frames, c->ws_zemo, out_dev + w0 * c->NB, nullptr, 1,
                                              stream, w0, (int)N)) return rc;
        } else {
            // generic shapes: staged log-mel, per-window logits gathered from the per-clip ones, GEMM-chain core
            if (int rc = launch_mel(c, c->mel_plans[0], audio_dev, nw, W, 0, c->ws_mel, c->ws_short, stream, L, step, w0, (int)N)) return rc;
            if (int rc = launch_gather_clip_logits(c, c->ws_zemo, c->ws_zemo_win, nw, w0, (int)N, stream)) return rc;
            if (int rc = launch_core_generic(c, c->ws_mel, nw, n_frames, c->ws_short, c->ws_zemo_win, out_dev + w0 * c->NB,
                                             nullptr, nullptr, stream)) return rc;
        }
    }
    if (smooth) return launch_ema_scan(c, out_dev, B, N, stream);
    return KM_OK;
}

}  // extern "C"
