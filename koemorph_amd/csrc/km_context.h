// Internal definitions shared by the translation units of libkoemorph_hip.so.
// Not part of the public ABI (that is include/koemorph.h).
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "koemorph.h"

namespace km {

constexpr int kNumMouth = 28;   // dual_stream_attention.py:14-45 (MOUTH_INDICES)
constexpr int kNumExpr = 24;    // EXPRESSION_INDICES
extern const int kMouthIdx[kNumMouth];
extern const int kExprIdx[kNumExpr];

struct HostParam {
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool loaded = false;
};

// A named host buffer produced by folding/packing, mirrored 1:1 on the device.
struct Packed {
    std::vector<float> host;
    float* dev = nullptr;
};

constexpr int kMelRpWaves = 8;      // waves per workgroup of mel_power_rp_kernel
constexpr int kMelRpGroups = 4;     // filter groups a wave can hold (8 x 4 x 4 = up to 128 filters)
constexpr int kMelRpRow = 580;      // power-row stride in dwords (4 mod 64: see the mel stage), >= fbg_extent

// Sparse triangular mel filterbank + window for one front-end configuration.
struct MelPlan {
    km_mel_config cfg{};
    int n_freq = 0;                 // n_fft/2 + 1
    std::vector<float> window;      // n_fft, periodic Hann (optionally / sqrt(sum w^2))
    std::vector<float> twiddle;     // n_fft complex (cos, -sin) pairs: W_N^p
    std::vector<int32_t> fb_start;  // n_mels: first FFT bin with non-zero weight
    std::vector<int32_t> fb_count;  // n_mels: number of bins
    std::vector<int32_t> fb_offset; // n_mels: offset into fb_weight
    std::vector<float> fb_weight;   // concatenated non-zero weights
    // the same filters for mel_power_rp_kernel (km_host.cpp build_mel_plan): groups of four consecutive filters dealt to
    // the workgroup's waves (balanced by length); a lane owns one (frame, filter) pair and walks the filter four bins a step
    std::vector<int32_t> fbg_gid;   // [wave][slot]: group (filters 4 gid .. 4 gid + 3) or -1
    std::vector<int32_t> fbg_desc;  // [wave][slot][filter in group]: first bin / 4 | steps << 8 | tap offset (float4) << 16
    std::vector<float> fbg_weight;  // four taps x 1/4 per (filter, step), zero padded to the group's step count
    int fbg_extent = 0;             // 1 + the highest bin a step reads (the power rows are padded with zeros up to it)
    // device mirrors
    float* d_window = nullptr;
    float* d_twiddle = nullptr;
    int32_t* d_fb_start = nullptr;
    int32_t* d_fb_count = nullptr;
    int32_t* d_fb_offset = nullptr;
    float* d_fb_weight = nullptr;
    int32_t* d_fbg_gid = nullptr;
    int32_t* d_fbg_desc = nullptr;
    float* d_fbg_weight = nullptr;
    bool uploaded = false;
};

// Run-time switches (km_set_option).  Seeded ONCE per handle, at creation, from the KM_* environment variables of
// the same name in upper case (tools/ab_*.sh); nothing on a launch path reads the environment.
struct Options {
    int core_split = 0;            // 3 / 6: split-bf16 variant of the fused core (experimental, never the default)
    int seq_per_window = 0;        // 1: sequence mode recomputes every window's STFT (the reference's schedule)
    int generic_staged = 0;        // 1: generic core from a staged log-mel image instead of the power-mel workspace
    int mel_two_frame = 0;         // 1: two-frames-per-wave front end (A/B baseline)
    int emotion_separate = 0;      // 1: emotion logits in their own kernel
    int no_ln_fusion = 0, no_db_fusion = 0, no_score_fusion = 0, no_out_fusion = 0, no_v_fusion = 0;   // generic chain A/B
    int legacy_no_merge = 0;       // 1: the legacy model's attention and tail as two launches instead of one (legacy_attn_tail_kernel)
    int no_core_merge = 0;         // 1: the d_model 512 core as its three launches (encoder + LayerNorm, scores, output) instead of one workgroup per window walking the three stages (core512_kernel)
    int legacy_no_tail_fusion = 0; // 1: SimplifiedKoeMorphModel out_proj + decoder as four GEMM launches + a row kernel (A/B, tests)
    int legacy_no_enc_fusion = 0;  // 1: SimplifiedKoeMorphModel audio encoder + key / value projections as four GEMM launches (A/B, tests)
    int legacy_no_attn_fusion = 0; // 1: SimplifiedKoeMorphModel attention as two batched strided products + a row softmax (A/B, tests)
    int kmm_no_fuse = 0;           // 1: KoeMorphModel as the launch-per-step chain even at the fused kernels' width (A/B, tests)
    int train_chain = 0;           // 1: training step as the round-1 launch-per-op chain (A/B reference; no dropout)
    int train_no_split = 0;        // 1: no split-K of the long gradient products of the phased training step (A/B)
    int train_dwce_parts = 0;      // > 0: partial sums of the channel-encoder gradient per step (a divisor of the batch; A/B)
    int train_split_min_k = 0;     // > 0: split gradient products longer than this many rows into chains of about this length (A/B; default 1024 / 640)
    int train_alone_max = 0;       // > 0: a phase of up to this many workgroups gives its 32-row LDS-DMA tiles the 8-stage ring (default 256)
    int train_no_dy_split = 0;     // 1: dY = dKV Wkv as one product at every batch size (A/B of the two K halves summed by the LayerNorm backward); 2: two halves at every batch size (measured: 16 windows 0.1401 -> 0.1422 ms, 64: 0.2299 -> 0.2327)
    int train_ln_fuse_rows = 0;    // > 0: LayerNorm by the reader up to this many key rows per step (default 3200 = 40 windows of 80 channels)
    int train_no_ln_fuse = 0;      // 1: the training program keeps its LayerNorm phase (P2) instead of normalising in the readers of Y0 / E0 (LnXform)
    int train_colsum_gemm = 0;     // 1: column sums of the training program as products with a ones vector on the matrix pipe (rounds 2-4a) instead of OP_COLSUM
    int train_no_fe_pack = 0;      // 1: km_train_step_audio converts and packs the power-mel in phase 0 of the program (round 3/4 form) instead of inside the front-end launch
    int train_attn_regs = 0;       // 1: the attention blocks of the training program as the register-staged blocks of round 3 (A/B of the LDS-DMA blocks)
    int train_no_dma = 0;          // 1: the products of the training program run on the register-staged tile only (A/B of the LDS-DMA tile, km_gemm_dma_dev.h)
    int train_op_per_launch = 0;   // 1: every operation of the training program is its own launch (timing aid: rocprofv3 then shows each operation)
    int train_bm32_below = 0;      // > 0: products with fewer 64-row tiles than this run on 32-row tiles (A/B; default 192)
    int train_tail_groups = 0;     // > 0: workgroups of the loss tail (A/B; default: one per 4 windows, at most 32)
};
void options_from_env(Options& o);
int set_option(struct Context* c, const char* name, long long value);

// hipFuncSetAttribute applies to the CURRENT device: one bit per device and call site, not one per process
struct PerDeviceOnce {
    unsigned long long seen[4] = {0, 0, 0, 0};
    bool first(int device) {
        if (device < 0 || device >= 256) return true;
        const unsigned long long bit = 1ull << (device & 63);
        if (seen[device >> 6] & bit) return false;
        seen[device >> 6] |= bit;
        return true;
    }
};

struct Context {
    km_config cfg{};
    Options opt{};
    int kind = 0;                                // 0 dual-stream production model, 1 legacy SimplifiedKoeMorphModel, 2 legacy KoeMorphModel
    int legacy_hidden = 128;
    km_koemorph_config kmm{};                    // kind 2
    int64_t kmm_batch = 0, kmm_frames = 0;       // kind 2: reserved workspace (ws_generic)
    bool legacy_tail_fused = false;              // kind 1: the blob of legacy_tail_kernel exists as well (decoder 128 wide, 52 queries)
    bool legacy_fused = false;                   // kind 1: the blob of legacy_encoder_kernel exists (d_model 256, 80 mel bins, 8 heads)
    bool kmm_fused = false;                      // kind 2: the model has the width of the fused kernels (km_kmmf.hip) and their blobs exist
    int d = 0, H = 0, hd = 0, T = 0, KT = 0, ED = 0, DH = 0, NB = 0, NK = 0;
    std::map<std::string, HostParam> params;     // reference state-dict tensors (fp32 masters)
    std::vector<std::string> param_order;
    std::map<std::string, Packed> packed;        // folded / packed buffers
    bool host_finalized = false;
    bool dev_finalized = false;
    bool fused_ok = false;                       // (d,T,H) has the fused gfx950 kernel
    float alpha = 0.0f;                          // sigmoid(smoothing_alpha)
    int device = -1;

    std::vector<MelPlan*> mel_plans;             // [0] = cfg.mel

    // workspace (device)
    int64_t ws_windows = 0, ws_samples = 0;
    float* ws_zemo = nullptr;      // (windows)             emotion-stream logit
    float* ws_zemo_win = nullptr;  // (windows)             per-window copy of per-clip logits (generic sequence mode)
    float* ws_melpow = nullptr;    // (windows, frames, 80) power-mel
    unsigned* ws_melmax = nullptr; // (windows)             max power (float bits)
    unsigned* ws_chunkctr = nullptr;   // (ws_chunkctr_cap) chunk requests per window of mel_power_rp_kernel: zero between launches
    int64_t ws_chunkctr_cap = 0;
    float* ws_mel = nullptr;       // (windows, frames, 80) log-mel
    float* ws_short = nullptr;     // (windows, 3, 80)
    float* ws_generic = nullptr;   // generic (non-fused) core intermediates, generic_ws_floats() per window
    int64_t ws_frames = 0;
    int ws_mels = 0;               // mel bins per frame the power-mel / log-mel / short-term workspaces were sized for
    // streaming state (km_stream_*): device-resident per-stream audio rings (MelAudioBuffer semantics)
    int64_t n_streams = 0, ring_len = 0;
    int ring_hop = 0;
    float* ring = nullptr;              // (n_streams, ring_len)
    int* ring_wptr = nullptr;           // (n_streams) write pointer == chronological start once full
    int* ring_frames = nullptr;         // (n_streams) frames added
    unsigned char* ring_ready = nullptr;    // (n_streams) is_full
    unsigned char* ring_started = nullptr;  // (n_streams) EMA state valid
    float* ring_state = nullptr;        // (n_streams, 52) EMA state
    MelPlan* stream_plan = nullptr;
    int64_t stream_out_frames = 0;
    // training state (km_train_*): flat fp32 master parameters + AdamW moments on the device, in state-dict order
    int64_t tr_nparams = 0, tr_windows = 0;
    int64_t tr_early = 0;            // floats [0, tr_early) of the gradient bucket are final when tr_ev[0] fires (phased step)
    bool tr_early_recorded = false;
    std::map<std::string, int64_t> tr_offset;
    float* tr_params = nullptr; float* tr_m = nullptr; float* tr_v = nullptr;
    float* trp_wcep = nullptr;     // (d, KP) the channel encoder weight with rows padded to KP floats (zeros): written with tr_params (upload, AdamW)
    float* tr_act = nullptr; float* tr_q = nullptr; float* tr_dq = nullptr; float* tr_part = nullptr; float* tr_gnorm = nullptr;
    float* tr_loss = nullptr;
    float* tr_red = nullptr;         // split-reduction partials
    int* tr_steps = nullptr;         // device-side AdamW step counters (graph replay safe)
    bool tr_alpha_live = false;      // smoothing_alpha was in the last step's graph (see adamw_kernel)
    km_loss_config tr_loss_cfg{};    // extra KoeMorphLoss terms (all weights 0 = off)
    // phased training step (km_trainp.hip): its own activation workspace, dropout masks and per-step mask counter
    float* trp_act = nullptr; int64_t trp_act_floats = 0;
    float* trp_split = nullptr; int64_t trp_split_floats = 0;   // partial outputs of the split-K gradient products of one step
    float* trp_tail_part = nullptr; unsigned* trp_tail_ctr = nullptr;   // per-workgroup sums of the loss tail + its arrival counter
    void* trp_masks = nullptr;       // bytes: mel (W,H,28,NK) | emo (W,H,24) | dec (W,52,DH), W = tr_windows
    int* trp_drop_ctr = nullptr;     // device-side step counter of the mask generator (graph-replay safe)
    float tr_dropout_p = 0.f;        // training-mode dropout probability (0 = eval-mode arithmetic)
    int tr_dropout_mode = 0;         // 0: masks drawn per step (Philox), 1: masks supplied by the caller (km_train_set_dropout_masks)
    unsigned long long tr_dropout_seed = 0;
    // side stream of the training step (emotion stream + decoder weight gradients run beside the mel chain)
    void* tr_s2 = nullptr; void* tr_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; float* tr_red2 = nullptr;
    // shared-frame sequence mode buffers (grow-only, allocated by km_sequence_forward)
    float* seq_pow = nullptr; unsigned* seq_fmax = nullptr; float* seq_edge = nullptr; unsigned* seq_emax = nullptr;
    int64_t seq_pow_cap = 0, seq_edge_cap = 0;
    int64_t tr_alpha_steps = 0;
    bool stage_timing = false;
    void* stage_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // hipEvent_t: emo b/e, mel e, core b/e
    // two-deep pipeline across calls (km_forward_audio_pipelined)
    void* pipe_s1 = nullptr; void* pipe_s2 = nullptr;      // hipStream_t: front end / core
    void* pipe_ev_in[2] = {nullptr, nullptr}; void* pipe_ev_mel[2] = {nullptr, nullptr}; void* pipe_ev_core[2] = {nullptr, nullptr};
    float* pipe_melpow[2] = {nullptr, nullptr}; unsigned* pipe_melmax[2] = {nullptr, nullptr}; float* pipe_zemo[2] = {nullptr, nullptr};
    bool pipe_dirty[2] = {true, true};
    int64_t pipe_seq = 0, pipe_windows = 0, pipe_frames = 0;
    bool melmax_dirty = true;      // ws_melmax may hold stale maxima (see launch_mel_power)
};

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

// km_host.cpp
int finalize_host(Context* c);
MelPlan* build_mel_plan(const km_mel_config& cfg);
MelPlan* find_or_add_plan(Context* c, const km_mel_config& cfg);

// km_core.hip
int launch_emotion(Context* c, const float* emo, int64_t B, float* zemo, void* stream);
int launch_core_fused(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short,
                      const float* zemo, float* out, float* raw, float* attn, float* state, int first,
                      void* stream);
// fused variant: reads the workspace power-mel + window maxima, applies the log/dB conversion on load
// Shared-frame sequence mode, core side (see CoreArgs in km_core.hip)
struct SeqCore {
    const float* pow;     // (clips, nfc, 80) clip-level power-mel
    const float* edge;    // (clips * n_per_clip, 2, 80) first / last frame of every window
    int nfc, stride, n_per_clip;
};
int launch_core_fused_db(Context* c, MelPlan* p, int64_t B, int64_t n_frames, const float* zemo, float* out,
                         float* state, int first, void* stream, int64_t win0 = 0, int zemo_div = 1, int64_t n_use = 0,
                         const unsigned char* ready = nullptr, unsigned char* started = nullptr,
                         const struct SeqCore* seq = nullptr);
int launch_seq_window_max(Context* c, const unsigned* fmax, const unsigned* emax, int64_t nw, int64_t win0, int nfc, int stride,
                          int n_per_clip, int n_frames, void* stream);
int launch_ema_scan(Context* c, float* x, int64_t B, int64_t N, void* stream);
int launch_smooth(Context* c, float* x, float* state, int64_t B, int first, void* stream);

// km_train.hip
int64_t train_act_floats(Context* c);
int train_forward_backward(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short, const float* emo,
                           const float* target, float mse_w, float l1_w, float* flat_grad, float* loss_dev, float* out_dev,
                           float* ema_state, int ema_first, void* stream);
int train_refresh_padded_weights(Context* c, void* stream);
int train_adamw(Context* c, const float* flat_grad, float lr, float b1, float b2, float eps, float wd, float max_norm,
                int64_t step, void* stream);

// km_trainp.hip
int64_t trainp_act_floats(Context* c, int64_t* fixed);
int64_t trainp_kp(Context* c);
int64_t trainp_mask_alloc_bytes(Context* c);
int trainp_mask_sizes(Context* c, int64_t B, int64_t* mel, int64_t* emo, int64_t* dec);
int trainp_copy_masks(Context* c, int64_t B, unsigned char* mel, unsigned char* emo, unsigned char* dec, int to_device, void* stream);
// from-audio training step: the front end's power-mel + window maxima, converted and packed by phase 0 of the program
struct LogParams;
// packed: the front end wrote 10 log10(power) into the packed input itself (MelPack); the readers finish the dB conversion
struct TrainAudioSrc { const float* melpow; const unsigned* melmax; int n_frames; const LogParams* lp; bool packed; };
int train_forward_backward_phased(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short, const float* xp_dev,
                                  const TrainAudioSrc* asrc,
                                  const float* emo, const float* target, float mse_w, float l1_w, float* flat_grad, float* loss_dev,
                                  float* out_dev, float* ema_state, int ema_first, void* stream);
int launch_audio_energy(const float* feats, int64_t B, int64_t T, int64_t D, float* out, void* stream);

// km_generic.hip
int64_t generic_ws_floats(Context* c);
int64_t legacy_ws_floats(Context* c, int64_t frames);
int finalize_host_legacy(Context* c);
int finalize_host_koemorph(Context* c);
int64_t koemorph_ws_floats(Context* c, int64_t T);
int launch_koemorph(Context* c, const float* mel, const float* emo, int64_t B, int64_t T, const unsigned char* kvalid, const float* prev,
                    float* state, int apply_constraints, float* out, float* raw, float* attn, void* stream);
struct LegacyPowSrc { const float* melpow; unsigned* melmax; const LogParams* lp; };      // the front end's power-mel as the legacy model's input
bool legacy_pow_ok(Context* c);
int launch_legacy(Context* c, const float* mel, int64_t B, int64_t T_mel, float* out, void* stream, const LegacyPowSrc* pow_src = nullptr);
int launch_gather_clip_logits(Context* c, const float* zclip, float* zwin, int64_t nw, int64_t w0, int wins_per_clip, void* stream);
int launch_core_generic_packed(Context* c, const float* xp, int64_t B, const float* zemo, float* out, float* raw, float* attn,
                               void* stream);
float* generic_packed_x(Context* c, int64_t B);
bool generic_core_takes_power(Context* c);
int launch_core_generic_power(Context* c, MelPlan* plan, int64_t B, int64_t n_frames, const float* zemo, float* out,
                              float* raw, float* attn, void* stream);
int launch_core_generic(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short, const float* zemo,
                        float* out, float* raw, float* attn, void* stream);

// km_mel.hip
// Shared-frame sequence mode: the front end writes n_rows rows per window (row r = STFT frame r * frame_mul) into
// `pow` and the per-row maxima (float bits, zero-initialised by the caller) into `fmax`, instead of the workspace.
struct SeqFrames {
    float* pow;        // (B, n_rows, n_mels)
    unsigned* fmax;    // (B, n_rows)
    int64_t n_rows;
    int frame_mul;
};
// packed training input written by the front end itself (MelArgs::pack_*): xt (B, n_mels, KP), T long frames per row
struct MelPack { float* xt; int T, KP; };
int launch_mel_power(Context* c, MelPlan* p, const float* audio, int64_t B, int64_t L, void* stream,
                     int64_t clip_len = 0, int64_t win_step = 0, int64_t win0 = 0, int wins_per_clip = 1,
                     const int* ring_start = nullptr, const unsigned char* ready = nullptr,
                     const float* emotion = nullptr, float* zemo = nullptr, const SeqFrames* seq = nullptr,
                     const MelPack* pack = nullptr);
bool mel_packs(Context* c, MelPlan* p, int64_t n_frames, int64_t T);
bool mel_fuses_emotion(Context* c, MelPlan* p);
int ensure_chunk_counters(Context* c, int64_t windows, void* stream);
int launch_ring_push(Context* c, const float* samples, int64_t n_per_stream, void* stream);
int launch_mel(Context* c, MelPlan* p, const float* audio, int64_t B, int64_t L, int64_t out_frames,
               float* mel_long, float* mel_short, void* stream, int64_t clip_len = 0, int64_t win_step = 0,
               int64_t win0 = 0, int wins_per_clip = 1);
int launch_mel_packed(Context* c, MelPlan* p, const float* audio, int64_t B, int64_t L, float* xp, int T, int KP, void* stream);
int upload_mel_plan(MelPlan* p);
void free_mel_plan(MelPlan* p);

}  // namespace km

// the opaque handle type of the public header
struct km_context : public km::Context {};
