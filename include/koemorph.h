/*
 * koemorph.h -- C-ABI of libkoemorph_hip.so, the MI355X (gfx950) implementation of the
 * KoeMorph hot path.
 *
 * The reference (atsuki-ichikawa/KoeMorph) is 100 % Python and has no FFI / plugin
 * interface; its boundary for this path is a set of torch.nn.Module.forward methods
 * (SURVEY.md section 8b).  This header is what a maintainer would bind from those methods
 * (ctypes stub in INTEGRATION.md).  Every entry point cites the reference interface it
 * replaces as path:line relative to the reference repository root.
 *
 * Conventions
 *   - plain C: pointers, sizes, int status codes.  No torch / C++ types.
 *   - every function returns KM_OK (0) or a negative km_status; km_last_error() gives a
 *     thread-local message for the last failure on the calling thread.
 *   - all `const float*` / `float*` arguments named *_dev are DEVICE pointers owned by the
 *     caller (e.g. torch tensor.data_ptr()), fp32, contiguous, 16-byte aligned.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Launch functions
 *     (everything taking a stream except km_finalize / km_reserve) neither allocate nor
 *     synchronise, so they can be captured into a hipGraph.
 *   - the library owns a packed copy of the weights and a workspace; km_reserve sizes the
 *     workspace up front, launch functions fail with KM_ERR_WORKSPACE if it is too small.
 *   - not re-entrant per handle (the reference modules are not either: stateful EMA,
 *     src/model/simplified_dual_stream_model.py:164); use one handle per thread/stream.
 */
#ifndef KOEMORPH_H
#define KOEMORPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: km_koemorph_config grew (output_activation, smoothing_method, smoothing_window); km_loss_config grew (ds_*) and now
 * starts with its own abi_version -- a caller built against the version-1 header is refused instead of being read past. */
#define KM_ABI_VERSION 2

typedef enum km_status {
    KM_OK = 0,
    KM_ERR_INVALID_ARG = -1,   /* bad shape / NULL / misaligned pointer (reference: ValueError, src/features/stft.py:115-116) */
    KM_ERR_UNSUPPORTED = -2,   /* configuration has no kernel */
    KM_ERR_NOT_FINALIZED = -3, /* forward before km_finalize, or a parameter is missing */
    KM_ERR_WORKSPACE = -4,     /* km_reserve was not called with a large enough batch */
    KM_ERR_HIP = -5,           /* HIP runtime error; message in km_last_error() */
    KM_ERR_NOT_READY = -6      /* streaming: ring not full yet (reference returns None, src/features/mel_sliding_window.py:126-127) */
} km_status;

typedef struct km_context* km_handle;

/* Mel front-end variants (the reference has three inconsistent ones). */
typedef enum km_mel_scale { KM_MEL_SLANEY = 0, KM_MEL_HTK = 1 } km_mel_scale;
typedef enum km_pad_mode { KM_PAD_CONSTANT = 0, KM_PAD_REFLECT = 1 } km_pad_mode;
typedef enum km_log_mode {
    KM_LOG_DB_MAX = 0, /* librosa.power_to_db(ref=np.max, amin, top_db) then (x + db_add) * db_scale */
    KM_LOG_LN_EPS = 1  /* log(mel + log_eps)  (src/features/stft.py:123) */
} km_log_mode;

typedef struct km_mel_config {
    int32_t sample_rate;   /* 16000 */
    int32_t n_fft;         /* 512 or 1024 */
    int32_t hop_length;    /* int(sample_rate / target_fps): 533 @30 fps, 266 @60 fps */
    int32_t n_mels;        /* 80 */
    float f_min;           /* 80 */
    float f_max;           /* 8000 */
    int32_t mel_scale;     /* km_mel_scale */
    int32_t slaney_norm;   /* 1: area-normalised triangles (librosa norm='slaney') */
    int32_t pad_mode;      /* km_pad_mode; frames are always centred (center=True) */
    int32_t window_norm;   /* 1: divide the STFT by sqrt(sum(w^2)) (torchaudio normalized=True) */
    int32_t log_mode;      /* km_log_mode */
    float amin;            /* 1e-10 */
    float top_db;          /* 80 */
    float db_add;          /* 80  -> (x + 80) / 80, simplified_dual_stream_model.py:200; 0 for none */
    float db_scale;        /* 1/80;  1 for none */
    float log_eps;         /* 1e-8 */
} km_mel_config;

/* Mirrors DualStreamCrossAttention.__init__ (src/model/dual_stream_attention.py:57-70)
 * plus the front end and smoothing of SimplifiedDualStreamModel
 * (src/model/simplified_dual_stream_model.py:28-55,163). */
typedef struct km_config {
    int32_t abi_version;          /* KM_ABI_VERSION */
    int32_t d_model;              /* 256 */
    int32_t num_heads;            /* 8 */
    int32_t num_mel_channels;     /* 80 */
    int32_t mel_sequence_length;  /* 256 */
    int32_t mel_temporal_frames;  /* 3 */
    int32_t emotion_dim;          /* 256 */
    int32_t num_blendshapes;      /* 52 */
    float temperature;            /* 1.0 */
    km_mel_config mel;            /* batch front end of the model */
} km_config;

/* ---- lifecycle ---------------------------------------------------------------------- */
int km_abi_version(void);
const char* km_last_error(void);

/* Construct for the current HIP device.  Replaces DualStreamCrossAttention.__init__ /
 * SimplifiedDualStreamModel.__init__.  Works without a GPU (host-side state only) so that
 * parameter folding can be exercised on the CPU; device memory is touched from km_finalize on. */
int km_create(const km_config* cfg, km_handle* out);
int km_destroy(km_handle h);

/* Replaces nn.Module.load_state_dict for one tensor.  `key` is the reference state-dict key
 * relative to DualStreamCrossAttention (e.g. "mel_attention.in_proj_weight"; the full-model
 * prefix "dual_stream_attention." is accepted and stripped) or "smoothing_alpha"
 * (simplified_dual_stream_model.py:163).  `data` is a HOST pointer to fp32. */
int km_load_param(km_handle h, const char* key, const float* data, const int64_t* shape, int32_t ndim);
/* Copy a parameter back to the host (state_dict()); n = number of floats in `out`. */
int km_get_param(km_handle h, const char* key, float* out, int64_t n);
/* Number of parameters the configuration expects / has received so far. */
int km_param_count(km_handle h, int32_t* expected, int32_t* loaded);

/* Fold + pack the weights for the kernels (host, double precision) and upload them.
 * Must follow the last km_load_param and precede any forward.  Allocates and synchronises. */
int km_finalize(km_handle h, void* stream);
/* Host half of km_finalize only (no GPU needed): used by the CPU tests. */
int km_finalize_host(km_handle h);
/* Size the workspace for batches up to max_windows windows of up to max_samples audio samples
 * (0 = core only).  Allocates; call outside any graph capture. */
int km_reserve(km_handle h, int64_t max_windows, int64_t max_samples);

/* ---- forward path ---------------------------------------------------------------------- */

/* Batch log-mel front end.  Replaces SimplifiedDualStreamModel.extract_mel_features
 * (src/model/simplified_dual_stream_model.py:166-229) with the handle's km_config.mel:
 * audio_dev (B, L) -> mel_long_dev (B, n_frames, n_mels), mel_short_dev (B, 3, n_mels) = the
 * last three frames, n_frames = 1 + L / hop.  mel_short_dev may be NULL. */
int km_mel_batch(km_handle h, const float* audio_dev, int64_t B, int64_t L,
                 float* mel_long_dev, float* mel_short_dev, void* stream);
int64_t km_mel_num_frames(km_handle h, int64_t L);

/* Stand-alone front end with an explicit configuration and output frame policy.  Replaces
 * MelSpectrogramExtractor.forward (src/features/stft.py:101-142) and
 * MelSlidingWindowExtractor.process_audio_batch / the per-tick extraction
 * (src/features/mel_sliding_window.py:280-307,326-365).  out_frames > 0 truncates, or pads by
 * repeating the last frame, to exactly out_frames rows (stft.py:130-140). */
int km_mel_extract(km_handle h, const km_mel_config* cfg, const float* audio_dev, int64_t B, int64_t L,
                   int64_t out_frames, float* mel_dev, void* stream);

/* Attention core.  Replaces DualStreamCrossAttention.forward
 * (src/model/dual_stream_attention.py:162-280), eval mode:
 *   mel_dev (B, T_in, 80)  zero-padded / truncated to mel_sequence_length (:193-202)
 *   mel_short_dev (B, 3, 80), emotion_dev (B, emotion_dim)
 *   out_dev (B, 52)                      'blendshapes'
 *   raw_dev (B, 52) or NULL              sigmoid outputs before the stream weights; the host
 *                                        derives 'mel_blendshapes' / 'emotion_blendshapes' (:257-262)
 *   attn_mel_dev (B, 28, 80) or NULL     'mel_attention_weights' (head-averaged)
 * ('emotion_attention_weights' is identically 1: softmax over a single key, :234-239.) */
int km_core_forward(km_handle h, const float* mel_dev, int64_t B, int64_t T_in,
                    const float* mel_short_dev, const float* emotion_dev,
                    float* out_dev, float* raw_dev, float* attn_mel_dev, void* stream);

/* The two halves of km_core_forward as separate launches (same arithmetic; bench.py times the
 * dominant kernel alone through km_core_forward_z):
 *   km_emotion_logit   emotion stream, z_dev (B) = the decoder logit shared by the 24 expression rows
 *                      (src/model/dual_stream_attention.py:216-218, :234-240, :248)
 *   km_core_forward_z  mel stream + decoder + stream weights with the logits supplied by the caller. */
int km_emotion_logit(km_handle h, const float* emotion_dev, int64_t B, float* z_dev, void* stream);
int km_core_forward_z(km_handle h, const float* mel_dev, int64_t B, int64_t T_in,
                      const float* mel_short_dev, const float* z_dev,
                      float* out_dev, float* raw_dev, float* attn_mel_dev, void* stream);

/* Temporal smoothing.  Replaces SimplifiedDualStreamModel.apply_temporal_smoothing
 * (src/model/simplified_dual_stream_model.py:341-368): alpha = sigmoid(smoothing_alpha);
 * first != 0 stores x into state and leaves x unchanged, otherwise x = alpha*x + (1-alpha)*state
 * and state = x.  x_dev, state_dev: (B, 52). */
int km_smooth(km_handle h, float* x_dev, float* state_dev, int64_t B, int32_t first, void* stream);

/* Whole-model forward.  Replaces SimplifiedDualStreamModel.forward
 * (src/model/simplified_dual_stream_model.py:370-415) with the emotion vector supplied by the
 * caller (openSMILE / emotion2vec are host-side providers, out of scope):
 * audio_dev (B, L) -> out_dev (B, 52).  state_dev (B,52) or NULL disables smoothing. */
int km_forward_audio(km_handle h, const float* audio_dev, int64_t B, int64_t L,
                     const float* emotion_dev, float* out_dev, float* state_dev, int32_t first,
                     void* stream);

/* Throughput mode of km_forward_audio: a two-deep software pipeline ACROSS calls.  The front end (VALU/LDS bound
 * FFT) of call i runs on an internal stream concurrently with the fused core (fp32-MFMA bound) of call i-1, which
 * the hardware co-schedules on the same CUs (separate matrix and vector pipes); the power-mel workspace is double
 * buffered.  Differences from km_forward_audio:
 *   - out_dev (and state_dev) of call i are complete only after the NEXT pipelined call returns and the caller's
 *     stream has advanced past it, or after km_pipeline_flush(h, stream);
 *   - audio_dev / emotion_dev of call i must stay untouched until then as well.
 * EMA state is applied in call order.  Mixing with the non-pipelined entry points requires a flush in between. */
int km_forward_audio_pipelined(km_handle h, const float* audio_dev, int64_t B, int64_t L,
                               const float* emotion_dev, float* out_dev, float* state_dev, int32_t first,
                               void* stream);
int km_pipeline_flush(km_handle h, void* stream);

/* Sequence forward.  Replaces SequentialDualStreamModel.forward
 * (src/model/sequential_dual_stream_model.py:63-167): slides a mel_sequence_length-frame
 * window with stride_frames over each clip, one output frame per position, EMA reset at the
 * clip start.  audio_dev (B, L) -> out_dev (B, N, 52), N = km_sequence_num_outputs().
 * Windows are addressed in place inside the clip (no (B*N, window) copy is materialised) and processed
 * in tiles of the reserved workspace size.  The reference recomputes every window's STFT; here, because windows
 * start at multiples of the hop, the clip's STFT is computed ONCE ((N-1)*stride + T+1 frames) plus the two
 * zero-padded boundary frames of each window, and the core reads rows from both images -- bit-identical to the
 * per-window evaluation (KM_SEQ_PER_WINDOW=1 selects that for comparison).  This entry point may allocate (grow-only
 * clip-level buffers sized B*L) on first use, so capture it in a hipGraph only after a warm-up call. */
int64_t km_sequence_num_outputs(km_handle h, int64_t L, int32_t stride_frames);
/* The temporal smoothing of a whole sequence, in place: x_dev (B, N, 52), y[0] = x[0], y[n] = alpha x[n] + (1 - alpha) y[n - 1]
 * along the frame axis with alpha = sigmoid(smoothing_alpha) -- what SequentialDualStreamModel.forward's per-position calls of
 * apply_temporal_smoothing add up to (src/model/sequential_dual_stream_model.py:99-151,
 * simplified_dual_stream_model.py:341-368).  It is the last step of km_sequence_forward(smooth = 1); exported so that a clip
 * whose output frames were computed in chunks on several GPUs (km_sequence_forward with smooth = 0 on each chunk, SURVEY.md
 * section 8e) is smoothed once, over the gathered frames. */
int km_ema_scan(km_handle h, float* x_dev, int64_t B, int64_t N, void* stream);
int km_sequence_forward(km_handle h, const float* audio_dev, int64_t B, int64_t L,
                        const float* emotion_dev, int32_t stride_frames, int32_t smooth,
                        float* out_dev, void* stream);

/* ---- training step (data-parallel ready) -------------------------------------------------------------
 * Replaces the body of SequentialTrainer.train_epoch (src/train_sequential.py:158-181: forward, loss,
 * backward, clip_grad_norm_(1.0), AdamW.step) for the 28 tensors of DualStreamCrossAttention + smoothing_alpha.
 * Parameters, AdamW moments and gradients are FLAT fp32 vectors in state-dict order (km_train_param_offset);
 * the gradient bucket is caller-owned so that the data-parallel build can all-reduce it over RCCL between
 * km_train_step* and km_train_adamw.  Eval-mode arithmetic (dropout = 0); loss = mse_weight * MSE +
 * l1_weight * L1 of the (optionally EMA-smoothed) prediction against target (src/model/losses.py:112-121).
 *   km_train_init        allocate the training state for up to max_windows windows per step and upload the
 *                        currently loaded parameters (allocates; call after km_finalize)
 *   km_train_step        mel_dev (B,T_in,80), mel_short_dev (B,3,80), emotion_dev (B,ED), target_dev (B,52) ->
 *                        flat_grad_dev (n_params), loss_dev (1), out_dev (B,52) or NULL; ema_state_dev (B,52) or
 *                        NULL applies the model's temporal smoothing inside the forward as the reference does
 *   km_train_step_audio  same from audio_dev (B, L): runs the log-mel front end first (no gradient flows into
 *                        it in the reference either: NumPy round trip, simplified_dual_stream_model.py:184-229)
 *   km_train_adamw       grad-norm clipping (max_grad_norm <= 0 disables) + AdamW on the flat vectors; `step`
 *                        is the 1-based optimizer step for the bias correction
 *   km_train_get_params / km_train_set_params   flat master copy <-> host
 *   km_train_sync        make the inference kernels see the trained weights (device master -> host store ->
 *                        fold + pack + upload) */
int km_train_init(km_handle h, int64_t max_windows, void* stream);
int64_t km_train_num_params(km_handle h);
int64_t km_train_param_offset(km_handle h, const char* key);
int km_train_step(km_handle h, const float* mel_dev, int64_t B, int64_t T_in, const float* mel_short_dev,
                  const float* emotion_dev, const float* target_dev, float mse_weight, float l1_weight,
                  float* flat_grad_dev, float* loss_dev, float* out_dev, float* ema_state_dev, int32_t ema_first,
                  void* stream);
int km_train_step_audio(km_handle h, const float* audio_dev, int64_t B, int64_t L, const float* emotion_dev,
                        const float* target_dev, float mse_weight, float l1_weight, float* flat_grad_dev,
                        float* loss_dev, float* out_dev, float* ema_state_dev, int32_t ema_first, void* stream);
int km_train_adamw(km_handle h, const float* flat_grad_dev, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float max_grad_norm, int64_t step, void* stream);
int km_train_get_params(km_handle h, float* flat_host, int64_t n);
int km_train_set_params(km_handle h, const float* flat_host, int64_t n);
int km_train_sync(km_handle h, void* stream);
/* AdamW state for checkpoints (the optimizer_state_dict of src/train_sequential.py:303-339): first and second moments as
 * flat vectors laid out like the parameters, and the two device-side step counters (all parameters, smoothing_alpha).
 * Both calls synchronise the device. */
int km_train_get_optimizer_state(km_handle h, float* exp_avg_host, float* exp_avg_sq_host, int64_t n, int32_t* steps2_host);
int km_train_set_optimizer_state(km_handle h, const float* exp_avg_host, const float* exp_avg_sq_host, int64_t n,
                                 const int32_t* steps2_host);

/* The remaining terms of KoeMorphLoss (src/model/losses.py:29-178), ADDED to the mse/l1 terms of km_train_step*:
 *   perceptual  four group-weighted MSEs: mouth cols 12..31 x2.0, eye 0..11 x1.0, brow 32..43 x1.0, jaw 44..51 x1.5
 *               (losses.py:306-338; the optional audio-visual term needs audio_features and is not computed, as when
 *               the reference is called with audio_features=None)
 *   temporal    MSE of (pred - prev_pred) vs (target - prev_target)           (losses.py:185-200)
 *   velocity    L1 of the same differences                                    (losses.py:202-217)
 *   sparsity    mean |pred|                                                   (losses.py:219-224)
 *   smoothness  mean |pred[:, j+1] - pred[:, j]| over the 51 adjacent pairs    (losses.py:226-234)
 *   landmark    MSE of pred W^T vs target W^T, W = the loss module's fixed (136, 52) matrix (losses.py:380-412)
 * prev_pred_dev / prev_target_dev (B,52) are constants (no gradient); temporal and velocity are skipped when either
 * is NULL, landmark when landmark_w_dev is NULL -- as the reference skips them.  Pointers must stay valid for every
 * later km_train_step* call (they are read on the step's stream).  cfg = NULL switches the extra terms off. */
typedef struct km_loss_config {
    int32_t abi_version;             /* KM_ABI_VERSION: km_train_set_loss refuses anything else (the struct has grown) */
    float perceptual_weight, temporal_weight, sparsity_weight, smoothness_weight, landmark_weight, velocity_weight;
    const float* prev_pred_dev;
    const float* prev_target_dev;
    const float* landmark_w_dev;
    const float* audio_energy_dev;   /* (B) per-window audio energy (km_audio_energy) or NULL: adds the audio-visual term
                                        0.5 * (1 - cos(mean mouth activation, audio energy)) to the perceptual loss
                                        (PerceptualBlendshapeLoss.forward with audio_features, losses.py:340-378) */
    /* DualStreamLoss (src/train_dual_stream.py:434-516).  Its L1 / L2 terms are l1_weight / mse_weight of km_train_step*
     * (defaults there 1.0 / 0.1); the two others:
     *   ds_velocity    weight x MSE(pred - prev_predictions, target - prev_predictions)   (:489-495), prev_predictions =
     *                  ds_prev_pred_dev (B, 52), a constant (no gradient); skipped when NULL, as the reference skips it
     *   ds_separation  weight x mean over the batch of | mean(pred[:, MOUTH]) - mean(pred[:, EXPRESSION]) |  (:498-514;
     *                  index sets of src/model/dual_stream_attention.py:14-45).  The reference computes it only when both
     *                  attention maps are passed in; here weight > 0 switches it on */
    float ds_velocity_weight, ds_separation_weight;
    const float* ds_prev_pred_dev;
} km_loss_config;
int km_train_set_loss(km_handle h, const km_loss_config* cfg);
/* (B, T, D) or (B, 1, D) audio features -> (B) energies for km_loss_config.audio_energy_dev: mean over T of the L2 norm
 * over D, as PerceptualBlendshapeLoss._compute_audiovisual_loss reduces its audio_features (losses.py:352-358). */
int km_audio_energy(const float* features_dev, int64_t B, int64_t T, int64_t D, float* energy_dev, void* stream);

/* Training-mode dropout: the reference trains under model.train() (src/train_sequential.py:118) with dropout p = 0.1 on
 * the attention weights of both nn.MultiheadAttention modules and on the decoder's hidden layer
 * (src/model/dual_stream_attention.py:106, :115, :153).  p = 0 (the state after km_train_init) is eval-mode arithmetic.
 *   km_train_set_dropout        p in [0, 1); masks are drawn per step by a counter-based Philox4x32-10 generator keyed by
 *                               `seed` and a device-side step counter (a captured step draws fresh masks on every replay);
 *                               external_masks != 0: the step uses the masks last given to km_train_set_dropout_masks
 *                               (parity tests replay the masks of reference-generated fixtures)
 *   km_train_get_dropout_masks  keep flags (1 = kept) of the most recent step, for B windows: mel (B, H, 28, n_mels),
 *                               emo (B, H, 24), dec (B, 52, d_model / 2) bytes on the host -- the layout of
 *                               oracle.core.core_forward(drop_masks=...)
 *   km_train_set_dropout_masks  the reverse (host -> device) */
int km_train_set_dropout(km_handle h, float p, uint64_t seed, int32_t external_masks);
/* The device-side step counter of the mask generator (part of a training checkpoint: a resumed run draws the masks the
 * uninterrupted run would have drawn). */
int km_train_get_dropout_step(km_handle h, int64_t* step);
int km_train_set_dropout_step(km_handle h, int64_t step);

/* Overlapping the data-parallel gradient all-reduce with the end of the backward pass (new construction: the reference is
 * single-process).  The flat bucket is laid out so that the tensors the backward pass finishes last come last:
 *   km_train_grad_split   *early_floats = E: floats [0, E) of flat_grad (83 % at d_model 256) are final after phase P11 of
 *                         km_train_step*'s program (the last but one or two of its launches), the rest when the call's work completes
 *   km_train_wait_early   make `stream` wait (hipStreamWaitEvent) for that point of the most recent km_train_step*: a
 *                         side stream can then all-reduce flat_grad[0:E] while the launch stream still computes the tail */
int km_train_grad_split(km_handle h, int64_t* early_floats);
int km_train_wait_early(km_handle h, void* stream);
int km_train_get_dropout_masks(km_handle h, int64_t B, uint8_t* mel_host, uint8_t* emo_host, uint8_t* dec_host, void* stream);
int km_train_set_dropout_masks(km_handle h, int64_t B, const uint8_t* mel_host, const uint8_t* emo_host, const uint8_t* dec_host,
                               void* stream);

/* ---- legacy single-stream variant --------------------------------------------------------------------
 * SimplifiedKoeMorphModel (src/model/simplified_model.py:12-156), used by the reference's src/train.py,
 * scripts/rt_simplified.py and scripts/test_model.py: same librosa-style log-mel front end, a two-layer
 * per-frame audio encoder, ONE nn.MultiheadAttention with 52 learnable blendshape queries over the T_mel
 * encoded frames (the literal "52 x 256" attention of the north star), a 3-layer decoder + sigmoid per query
 * row and a mean over the 52 rows.  State-dict keys: audio_encoder.{0,3}.*, attention.*, decoder.{0,3,6}.*,
 * blendshape_queries.  The handle is used with km_load_param / km_finalize / km_reserve / km_destroy as usual.
 *   km_legacy_forward      audio_dev (B, L)            -> out_dev (B, 52)   (forward, :114-149)
 *   km_legacy_forward_mel  mel_dev (B, T_mel, 80)      -> out_dev (B, 52)   (everything after extract_mel_features) */
typedef struct km_legacy_config {
    int32_t abi_version;       /* KM_ABI_VERSION */
    int32_t d_model;           /* 256 */
    int32_t num_heads;         /* 8 */
    int32_t decoder_hidden;    /* 128 */
    int32_t num_blendshapes;   /* 52 */
    km_mel_config mel;
} km_legacy_config;
int km_legacy_create(const km_legacy_config* cfg, km_handle* out);
int km_legacy_forward(km_handle h, const float* audio_dev, int64_t B, int64_t L, float* out_dev, void* stream);
int km_legacy_forward_mel(km_handle h, const float* mel_dev, int64_t B, int64_t T_mel, float* out_dev, void* stream);

/* ---- legacy multi-layer model: KoeMorphModel (src/model/gaussian_face.py:29-268) -----------------------
 * The model behind create_koemorph_model / scripts/rt.py:283-304, eval mode: DualStreamEncoder on both feature streams
 * (Linear + ReLU + LayerNorm, then num_encoder_layers post-norm nn.TransformerEncoderLayer with 8 heads, 4 d feed-forward,
 * exact GELU; src/model/dual_stream_attention.py:296-390), their average, BlendshapeQueryEmbedding conditioned on the
 * previous frame (src/model/attention.py:481-514), num_attention_layers x [MultiHeadCrossAttention with the causal and
 * window masks of attention.py:208-246, residual, LayerNorm], BlendshapeDecoder (diagonal of output_proj, sigmoid,
 * 0.9 / 0.1 mix with the previous frame; src/model/decoder.py:108-177), learnable TemporalSmoother (exponential,
 * gaussian or median: decoder.py:278-340) and BlendshapeConstraints (decoder.py:434-466).  d_query must equal d_model (the reference's
 * residual `attn_out + attention_output` needs it; its default d_query = 128 does not run).  State-dict keys are the
 * reference's; the buffers of the smoother / constraints are NOT parameters here: the smoother state is the caller's
 * (B, 52) device array (zero it for reset_temporal_state).  A query row whose keys are all masked yields NaN, as in
 * the reference.
 *   km_koemorph_reserve  workspace for max_batch x max_frames
 *   km_koemorph_forward  mel_dev (B, T, mel_dim), emotion_dev (B, T, emotion_dim), audio_mask_dev (B, T) bytes, 1 = valid
 *                        frame, or NULL (padded frames are masked as keys in the encoder and in every cross-attention
 *                        layer, gaussian_face.py:180,204,224; what the encoder leaves AT padded positions is unspecified,
 *                        as in torch, and never reaches the output), prev_dev (B, 52) or NULL,
 *                        smoother_state_dev (B, 52) in/out or NULL (= apply_smoothing False), apply_constraints,
 *                        -> out_dev (B, 52), raw_dev (B, 52) or NULL, attn_dev (layers, B, H, 52, T) or NULL */
typedef struct km_koemorph_config {
    int32_t abi_version;            /* KM_ABI_VERSION */
    int32_t mel_dim;                /* 80 */
    int32_t emotion_dim;            /* 256 */
    int32_t d_model;                /* 256, a multiple of 8 and of num_heads */
    int32_t num_heads;              /* 8 */
    int32_t num_encoder_layers;     /* 2 */
    int32_t num_attention_layers;   /* 4 */
    int32_t decoder_hidden_dim;     /* 128 */
    int32_t decoder_layers;         /* 2 */
    int32_t decoder_activation;     /* 0 relu, 1 gelu (default), 2 swish = nn.SiLU, 3 leaky_relu(0.1)  (decoder.py:68-75) */
    int32_t causal;                 /* 1 */
    int32_t window_size;            /* 30; < 0 = None */
    int32_t use_temporal_smoothing; /* 1: temporal_smoother.alpha is a parameter */
    int32_t use_constraints;        /* 1 */
    int32_t num_blendshapes;        /* 52 */
    int32_t output_activation;      /* 0 sigmoid (default), 1 tanh, 2 none  (decoder.py:162-167) */
    int32_t smoothing_method;       /* TemporalSmoother (decoder.py:179-340), learnable=True as KoeMorphModel builds it:
                                       0 exponential: parameter temporal_smoother.alpha, state (B, 52);
                                       1 gaussian: parameter temporal_smoother.gaussian_weights (window), y = sum_k softmax(w)_k *
                                         ring[k] over the SLOTS of the history ring (:299-317: the weights go with the slot, not
                                         with the age of its content);
                                       2 median: torch.median over the ring's slots (:319-331; the lower middle value for an
                                         even window, NaN if a slot holds NaN).
                                       1 and 2 keep smoother_state_dev as (B, smoothing_window * 52 + 1) floats per call: each
                                       batch element's ring (window, 52) followed by its slot pointer; all zero = reset */
    int32_t smoothing_window;       /* 5 (TemporalSmoother's default; KoeMorphModel never passes another), 1..16 */
} km_koemorph_config;
int km_koemorph_create(const km_koemorph_config* cfg, km_handle* out);
int km_koemorph_reserve(km_handle h, int64_t max_batch, int64_t max_frames);
int km_koemorph_forward(km_handle h, const float* mel_dev, const float* emotion_dev, int64_t B, int64_t T,
                        const uint8_t* audio_mask_dev, const float* prev_dev, float* smoother_state_dev,
                        int32_t apply_constraints, float* out_dev, float* raw_dev, float* attn_dev, void* stream);

/* ---- streaming: many concurrent speaker streams, state resident on the device -------------------------
 * Replaces, for all streams of this GPU at once, MelAudioBuffer.add_audio_frame / get_current_audio
 * (src/features/mel_sliding_window.py:70-140), MelSlidingWindowExtractor.process_audio_frame (:252-324) and
 * SimplifiedDualStreamModel.process_audio_frame_realtime (src/model/simplified_dual_stream_model.py:452-498).
 *   km_stream_create  n_streams rings of int(context_window_s * sr) samples; the ring hop is
 *                     int(sr / (1 / update_interval_s)) (= 532 for 0.0333 s, reference quirk); mel_cfg is the
 *                     sliding-window front end (n_fft 1024 / hop 533 / reflect / dB when built by the model,
 *                     simplified_dual_stream_model.py:122-131), out_frames = int(context_window / update_interval).
 *                     Allocates; call once, outside any graph capture.
 *   km_stream_push    samples_dev (n_streams, n_per_stream): one frame per stream; n_per_stream must be within
 *                     +/-1 of the ring hop (reference rejects others, :80-82) and is padded / truncated to it.
 *   km_stream_tick    emotion_dev (n_streams, emotion_dim) -> out_dev (n_streams, 52) for every stream whose ring
 *                     is full; ready_dev (n_streams) u8 mirrors MelAudioBuffer.is_full (rows of not-ready streams
 *                     are left untouched).  Per-stream EMA state lives in the handle.  The short-term rows are
 *                     the last three kept frames.  push + tick never allocate or synchronise: capture them in a
 *                     hipGraph and replay it per 33 ms tick.
 *   km_stream_reset   clear rings, readiness and EMA state (MelSlidingWindowExtractor.reset :373-383). */
int km_stream_create(km_handle h, int64_t n_streams, double context_window_s, double update_interval_s,
                     const km_mel_config* mel_cfg);
int km_stream_push(km_handle h, const float* samples_dev, int64_t n_per_stream, void* stream);
int km_stream_tick(km_handle h, const float* emotion_dev, float* out_dev, uint8_t* ready_dev, void* stream);
int km_stream_reset(km_handle h, void* stream);

/* ---- measurement aid used by bench.py ------------------------------------------------------------
 * With stage timing enabled, km_forward_audio / km_forward_audio_pipelined record HIP events on the stream each
 * kernel is launched on, right before and after it (emotion logits, power-mel front end, fused core);
 * km_stage_times synchronises on the last event and returns the elapsed milliseconds of the most recent call:
 * ms[0] emotion, ms[1] front end, ms[2] core -- in pipelined mode these are the durations WHILE the other
 * call's kernel shares the chip, i.e. what rocprofv3 reports per dispatch.  Event records are not
 * graph-capturable: keep it off in production. */
int km_enable_stage_timing(km_handle h, int32_t enable);
int km_stage_times(km_handle h, float* ms3);

/* ---- eGeMAPSv02 functionals: the long-context emotion stream's feature extractor --------------------------------
 * Replaces the openSMILE call of OpenSMILEeGeMAPSExtractor._extract_features_from_audio
 * (src/features/opensmile_extractor.py:427-439): peak normalisation to [-1, 1] (:431-433), then the 88 eGeMAPSv02
 * functionals of the window (`opensmile.Smile(FeatureSet.eGeMAPSv02, FeatureLevel.Functionals).process_signal`, :227-235,
 * :439), for B windows of L samples at 16 kHz in one call.  openSMILE is a third-party package the reference neither
 * vendors nor pins: the arithmetic is the restatement of the published parameter set in oracle/egemaps.py (parity
 * unpinned; output order = openSMILE's feature order).
 *   km_egemaps_plan_create / _destroy   constant tables (windows, filterbank, log-frequency axis, ...) on the device
 *   km_egemaps_num_frames               10 ms frames of a window of L samples (60 ms frames, left aligned); at most 2048
 *   km_egemaps_workspace_floats         floats of caller-owned scratch km_egemaps_functionals needs for (B, L)
 *   km_egemaps_functionals              audio_dev (B, L) -> out_dev (B, 88); launches on `stream`, no allocation, no sync
 *   km_egemaps_records                  per-frame low-level descriptors of the last call on that workspace, (B, frames, 36)
 *                                       floats to the host (tests) */
int km_egemaps_plan_create(void** plan_out);
int km_egemaps_plan_destroy(void* plan);
int64_t km_egemaps_num_frames(int64_t L);
int64_t km_egemaps_workspace_floats(int64_t B, int64_t L);
int km_egemaps_functionals(void* plan, const float* audio_dev, int64_t B, int64_t L, int32_t normalize, float* work_dev,
                           int64_t work_floats, float* out_dev, void* stream);
int km_egemaps_records(const float* work_dev, int64_t B, int64_t L, float* rec_host, void* stream);
/* out (B, N) = x (B, K) w^T + b with w stored (N, K) like nn.Linear: the 264 -> 256 compression of the three concatenated
 * eGeMAPS windows (OpenSMILEeGeMAPSExtractor.get_concatenated_features, opensmile_extractor.py:575-590).  b may be NULL. */
int km_linear(const float* x_dev, const float* w_dev, const float* b_dev, int64_t B, int64_t K, int64_t N, float* out_dev, void* stream);

/* ---- run-time switches ---------------------------------------------------------------------------
 * Replaces what would be module attributes / environment switches on the reference side (the reference has none on
 * this path: every switch selects between two implementations of the SAME arithmetic, for A/B timing and for the
 * tests that pin one path against the other).  Names: "core_split" (0 | 3 | 6, experimental split-bf16 core),
 * "seq_per_window", "generic_staged", "mel_two_frame", "emotion_separate", "no_ln_fusion", "no_db_fusion",
 * "no_score_fusion", "no_out_fusion", "no_v_fusion", "train_chain" (training step as the launch-per-op chain),
 * "kmm_no_fuse" (km_koemorph_forward as the launch-per-step GEMM chain even at the width of the two fused kernels),
 * "legacy_no_enc_fusion" / "legacy_no_attn_fusion" / "legacy_no_tail_fusion" (km_legacy_forward's three fused kernels back to GEMM-chain launches), "train_op_per_launch" (timing aid:
 * every operation of the training program as its own launch), "train_bm32_below", "train_tail_groups", "train_split_min_k",
 * "train_no_dma" (the products of the training program on the register-staged tile instead of the LDS-DMA tile),
 * "train_attn_regs" (its attention blocks register-staged as in round 3), "train_no_fe_pack" (km_train_step_audio converts and
 * packs the front end's power-mel in phase 0 of the program instead of inside the front-end launch + on the channel encoder's
 * operand fragments), "train_colsum_gemm" (its column sums as ones-vector products on the matrix pipe), "train_no_ln_fuse" (a LayerNorm phase
 * instead of LayerNorm in the readers of the channel encoder's output) / "train_ln_fuse_rows" (the batch size, in key rows, up to which the readers do it).  A handle's switches start from the environment
 * variables KM_<NAME> read ONCE in km_create; no launch path reads the environment.  Unknown name: KM_ERR_INVALID_ARG. */
int km_set_option(km_handle h, const char* name, int64_t value);

/* ---- introspection used by the tests ------------------------------------------------- */
/* Copy a named host-side folded/packed buffer (after km_finalize_host) into out; returns its
 * length in floats via *n when out == NULL. */
int km_debug_buffer(km_handle h, const char* name, float* out, int64_t* n);

/* ---- window producer for sequence training -------------------------------------------------------------
 * Replaces the per-window host slicing + H2D copies of KoeMorphSequentialDataset (src/data/sequential_dataset.py:
 * 136-209) for clips and labels resident in HBM.
 *   km_resample_labels  labels recorded at another frame rate -> target rate, exactly as _resample_blendshapes
 *                       (:136-154): dst[t,k] = float32(np.interp(np.linspace(0, n_src-1, n_dst)[t], arange(n_src),
 *                       src[:,k])) evaluated in float64 like numpy (bit-identical); n_dst = int(n_src * target/source)
 *                       is the caller's (host) computation
 *   km_gather_windows   window b starts at frame start_frames[b]: audio_out (B, window_samples) =
 *                       clip[start*hop : +window_samples] (zero beyond the clip), labels_out (B, window_frames, dims) =
 *                       labels[start : +window_frames] (:181-188), target_out (B, dims) = the label of the window's
 *                       last frame; any of the three outputs may be NULL */
int km_resample_labels(const float* src_dev, int64_t n_src, int32_t dims, int64_t n_dst, float* dst_dev, void* stream);
int km_gather_windows(const float* clip_dev, int64_t clip_len, const int32_t* start_frames_dev, int64_t B, int32_t hop,
                      int64_t window_samples, float* audio_out_dev, const float* labels_dev, int64_t n_label_frames,
                      int32_t window_frames, int32_t dims, float* labels_out_dev, float* target_out_dev, void* stream);

/* ---- wire encoding (host only, no GPU) -----------------------------------------------------------------
 * Byte-identical replacement for the per-frame json.dumps of the reference's output side (scripts/rt.py:209-231:
 * UDP datagrams and JSONL lines; src/data/io.py:119-131: dataset labels):
 *     {"timestamp": <float>, "blendshapes": [<n_values floats>]}
 * frames_host (n_frames, n_values) fp32 on the host, timestamps (n_frames) as time.time() doubles; floats are printed as
 * CPython prints ndarray.tolist() values (shortest round-trip repr of the float32 widened to double, 'Infinity' /
 * 'NaN' as json.dumps spells them).  Frame f occupies out[offsets[f] .. offsets[f+1]) (offsets has n_frames + 1 entries,
 * may be NULL); newline != 0 appends '\n' to every frame (JSONL).  Returns the number of bytes written, or a negative
 * number -(n) when `capacity` is too small (n bytes are certainly enough), or KM_ERR_INVALID_ARG. */
int64_t km_format_frames(const float* frames_host, int64_t n_frames, int32_t n_values, const double* timestamps,
                         int32_t newline, char* out, int64_t capacity, int64_t* offsets);

#ifdef __cplusplus
}
#endif
#endif /* KOEMORPH_H */
