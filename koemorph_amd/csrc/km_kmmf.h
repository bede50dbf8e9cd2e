// Fused form of the legacy KoeMorphModel forward (src/model/gaussian_face.py:175-268) at its default width: the layout of the
// weight blobs km_host.cpp builds at km_finalize and km_kmmf.hip reads.  Shared by host and device code.
//
// Every nn.Linear weight W (N x K, row-major, N and K multiples of 16) is stored FRAGMENT-PACKED:
//     image[(t * K/16 + kb) * 64 + lane][e] = W[16 t + (lane & 15)][16 kb + 4 (lane >> 4) + e],   e = 0..3
// i.e. the 16-byte operand a lane feeds to the four v_mfma_f32_16x16x4_f32 of one 16-wide k block (lane group g contracts
// k = 4 g + s in MFMA s), for output tile t: one wave-wide load is 1 KB contiguous, a tile's k blocks are consecutive.
// The same image serves W as the B operand (x W^T, tile = 16 output columns) and as the A operand (W x^T, tile = 16 output rows).
#pragma once

#include <cstdint>

namespace kmmf {

constexpr int D = 256;        // d_model
constexpr int HEADS = 8;      // heads of the encoder layers (fixed by the reference) and of the cross-attention layers
constexpr int HD = 32;        // head width
constexpr int FF = 1024;      // encoder feed-forward width (4 d)
constexpr int NQ = 52;        // blendshape queries
constexpr int HID = 128;      // decoder hidden width
constexpr int TMAX = 32;      // frames (keys) per window

// ---- encoder blob "kmf_enc": [stream 0 = mel, 1 = emotion] x (head, layers...) ----
constexpr int64_t ENC_W0 = 0;                               // input projection, packed as (256 x 256): columns >= in_dim are zero
constexpr int64_t ENC_B0 = ENC_W0 + (int64_t)D * D;
constexpr int64_t ENC_LNG = ENC_B0 + D;
constexpr int64_t ENC_LNB = ENC_LNG + D;
constexpr int64_t ENC_HEAD = ENC_LNB + D;
constexpr int64_t EL_WIN = 0;                               // self_attn.in_proj_weight (768 x 256)
constexpr int64_t EL_BIN = EL_WIN + (int64_t)3 * D * D;
constexpr int64_t EL_WO = EL_BIN + 3 * D;                   // self_attn.out_proj
constexpr int64_t EL_BO = EL_WO + (int64_t)D * D;
constexpr int64_t EL_N1G = EL_BO + D;
constexpr int64_t EL_N1B = EL_N1G + D;
constexpr int64_t EL_W1 = EL_N1B + D;                       // linear1 (1024 x 256)
constexpr int64_t EL_B1 = EL_W1 + (int64_t)FF * D;
constexpr int64_t EL_W2 = EL_B1 + FF;                       // linear2 (256 x 1024)
constexpr int64_t EL_B2 = EL_W2 + (int64_t)D * FF;
constexpr int64_t EL_N2G = EL_B2 + D;
constexpr int64_t EL_N2B = EL_N2G + D;
constexpr int64_t ENC_LAYER = EL_N2B + D;
inline constexpr int64_t enc_stream_floats(int layers) { return ENC_HEAD + (int64_t)layers * ENC_LAYER; }

// ---- cross-attention blob "kmf_cross": per layer ----
constexpr int64_t CL_WQ = 0;
constexpr int64_t CL_BQ = CL_WQ + (int64_t)D * D;
constexpr int64_t CL_WK = CL_BQ + D;
constexpr int64_t CL_BK = CL_WK + (int64_t)D * D;
constexpr int64_t CL_WV = CL_BK + D;
constexpr int64_t CL_BV = CL_WV + (int64_t)D * D;
constexpr int64_t CL_WO = CL_BV + D;
constexpr int64_t CL_BO = CL_WO + (int64_t)D * D;
constexpr int64_t CL_LNG = CL_BO + D;
constexpr int64_t CL_LNB = CL_LNG + D;
constexpr int64_t CROSS_LAYER = CL_LNB + D;

// ---- decoder blob "kmf_dec": input projection, then per hidden layer ----
constexpr int64_t DC_WI = 0;                                // decoder.input_proj (128 x 256)
constexpr int64_t DC_BI = DC_WI + (int64_t)HID * D;
constexpr int64_t DC_HEAD = DC_BI + HID;
constexpr int64_t DL_W = 0;                                 // hidden_layers.i (128 x 128)
constexpr int64_t DL_B = DL_W + (int64_t)HID * HID;
constexpr int64_t DL_LNG = DL_B + HID;
constexpr int64_t DL_LNB = DL_LNG + HID;
constexpr int64_t DEC_LAYER = DL_LNB + HID;

// ---- legacy SimplifiedKoeMorphModel blob "lgf_enc" (d_model 256, 80 mel bins): audio_encoder + key / value projections ----
constexpr int LG_MEL = 80;
constexpr int64_t LG_W0 = 0;                                // audio_encoder.0 (256 x 80), 5 k blocks per tile
constexpr int64_t LG_B0 = LG_W0 + (int64_t)D * LG_MEL;
constexpr int64_t LG_W3 = LG_B0 + D;                        // audio_encoder.3 (256 x 256)
constexpr int64_t LG_B3 = LG_W3 + (int64_t)D * D;
constexpr int64_t LG_WK = LG_B3 + D;                        // attention in_proj rows [d, 2 d)
constexpr int64_t LG_BK = LG_WK + (int64_t)D * D;
constexpr int64_t LG_WV = LG_BK + D;                        // attention in_proj rows [2 d, 3 d)
constexpr int64_t LG_BV = LG_WV + (int64_t)D * D;
constexpr int64_t LG_FLOATS = LG_BV + D;
// blob "lgf_tail": attention out_proj and the decoder (Linear(256, 128) ReLU Linear(128, 128) ReLU Linear(128, 52) Sigmoid); the last
// weight padded to 64 rows (rows 52 .. 63 zero)
constexpr int64_t LT_WO = 0;
constexpr int64_t LT_BO = LT_WO + (int64_t)D * D;
constexpr int64_t LT_W0 = LT_BO + D;
constexpr int64_t LT_B0 = LT_W0 + (int64_t)HID * D;
constexpr int64_t LT_W3 = LT_B0 + HID;
constexpr int64_t LT_B3 = LT_W3 + (int64_t)HID * HID;
constexpr int64_t LT_W6 = LT_B3 + HID;
constexpr int64_t LT_B6 = LT_W6 + (int64_t)64 * HID;
constexpr int64_t LT_FLOATS = LT_B6 + 64;

}  // namespace kmmf
