"""Oracle: DualStreamCrossAttention.forward restated with explicit torch-CPU math.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every step cites the reference line it
follows in /root/reference/src/model/dual_stream_attention.py.  ``nn.MultiheadAttention``
is written out (packed in-projection, per-head scaled dot product, softmax, out
projection, head-averaged weights) after torch.nn.functional.multi_head_attention_forward,
which is what the reference's two ``nn.MultiheadAttention`` modules execute (:104-109,
:113-118, batch_first=True, eval mode => dropout is the identity).

All tensors are float32 by default (the reference's dtype); pass dtype=torch.float64 for
a high-precision version used to bound the fp32 rounding noise of both sides.
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

MOUTH_INDICES = list(range(14, 41)) + [51]          # dual_stream_attention.py:14-45
EXPRESSION_INDICES = list(range(0, 14)) + list(range(41, 51))


def _t(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def mha_forward(query, key_value, in_w, in_b, out_w, out_b, num_heads: int,
                need_weights: bool, dropout_p: float = 0.0,
                drop_mask: Optional[torch.Tensor] = None):
    """nn.MultiheadAttention(embed_dim, num_heads, batch_first=True)(q, kv, kv).

    query (B,Lq,d), key_value (B,Lk,d).  Returns (out (B,Lq,d), weights (B,Lq,Lk) or None)
    where weights are averaged over heads (average_attn_weights=True, the default the
    reference relies on at :225-230).
    """
    B, Lq, d = query.shape
    Lk = key_value.shape[1]
    hd = d // num_heads
    wq, wk, wv = in_w[:d], in_w[d:2 * d], in_w[2 * d:]
    bq, bk, bv = in_b[:d], in_b[d:2 * d], in_b[2 * d:]
    q = F.linear(query, wq, bq).view(B, Lq, num_heads, hd).transpose(1, 2)      # (B,H,Lq,hd)
    k = F.linear(key_value, wk, bk).view(B, Lk, num_heads, hd).transpose(1, 2)
    v = F.linear(key_value, wv, bv).view(B, Lk, num_heads, hd).transpose(1, 2)
    q = q * math.sqrt(1.0 / float(hd))                 # torch scales q before the product
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)                          # (B,H,Lq,Lk)
    p_used = p
    if drop_mask is not None:                          # training-mode hook for the tests
        p_used = p * drop_mask / (1.0 - dropout_p)
    o = (p_used @ v).transpose(1, 2).reshape(B, Lq, d)
    o = F.linear(o, out_w, out_b)
    return o, (p.mean(dim=1) if need_weights else None)


def core_forward(params: Dict[str, "np.ndarray | torch.Tensor"],
                 mel_features, mel_temporal_features, emotion_features,
                 num_heads: int = 8, mel_sequence_length: int = 256,
                 temperature: float = 1.0, return_attention: bool = False,
                 dtype=torch.float32, return_intermediates: bool = False,
                 dropout_p: float = 0.0, drop_masks: Optional[Dict[str, "np.ndarray | torch.Tensor"]] = None
                 ) -> Dict[str, torch.Tensor]:
    """DualStreamCrossAttention.forward (dual_stream_attention.py:162-280).

    Training mode: the module holds three dropouts (p = ``dropout``, :106, :115, :153) -- on the attention weights of
    both nn.MultiheadAttention modules (after the softmax, before P V) and on the decoder's hidden layer (after the
    ReLU).  ``drop_masks`` = {"mel": (B,H,28,80), "emo": (B,H,24,1), "dec": (B,52,d/2)} of 0/1 keep flags makes them
    explicit; kept values are scaled by 1/(1-p) as torch.nn.functional.dropout does."""
    P = {k: _t(v, dtype) for k, v in params.items()}
    mel = _t(mel_features, dtype)
    short = _t(mel_temporal_features, dtype)
    emo = _t(emotion_features, dtype)
    B = mel.shape[0]
    d = P["mel_channel_encoder.weight"].shape[0]

    # :189-202  (B,T,80) -> (B,80,T); zero-pad or truncate the time axis to T_seq
    x = mel.transpose(1, 2)
    T = x.shape[2]
    if T < mel_sequence_length:
        x = torch.cat([x, torch.zeros(B, x.shape[1], mel_sequence_length - T, dtype=dtype)], dim=2)
    elif T > mel_sequence_length:
        x = x[:, :, :mel_sequence_length]
    # :205-208  append the 3 short-term frames -> (B,80,T_seq+3)
    x = torch.cat([x, short.transpose(1, 2)], dim=2)
    # :211-212  per-channel encoder + LayerNorm(eps=1e-5)
    y = F.linear(x, P["mel_channel_encoder.weight"], P["mel_channel_encoder.bias"])
    y = F.layer_norm(y, (d,), P["mel_norm.weight"], P["mel_norm.bias"], 1e-5)
    # :216-218  emotion vector -> one token
    e = F.linear(emo, P["emotion_encoder.weight"], P["emotion_encoder.bias"]).unsqueeze(1)
    e = F.layer_norm(e, (d,), P["emotion_norm.weight"], P["emotion_norm.bias"], 1e-5)
    # :221-222
    qm = P["mouth_queries"].unsqueeze(0).expand(B, -1, -1)
    qe = P["expression_queries"].unsqueeze(0).expand(B, -1, -1)
    # :225-231
    dm = {k: _t(v, dtype) for k, v in (drop_masks or {}).items()}
    mo, mw = mha_forward(qm, y, P["mel_attention.in_proj_weight"], P["mel_attention.in_proj_bias"],
                         P["mel_attention.out_proj.weight"], P["mel_attention.out_proj.bias"],
                         num_heads, return_attention, dropout_p, dm.get("mel"))
    mo = F.linear(mo, P["mel_output_proj.weight"], P["mel_output_proj.bias"])
    # :234-240
    eo, ew = mha_forward(qe, e, P["emotion_attention.in_proj_weight"], P["emotion_attention.in_proj_bias"],
                         P["emotion_attention.out_proj.weight"], P["emotion_attention.out_proj.bias"],
                         num_heads, return_attention, dropout_p, dm.get("emo"))
    eo = F.linear(eo, P["emotion_output_proj.weight"], P["emotion_output_proj.bias"])
    # :243-245
    nb = P["mel_weights"].shape[0]
    comb = torch.zeros(B, nb, d, dtype=dtype)
    comb[:, MOUTH_INDICES] = mo
    comb[:, EXPRESSION_INDICES] = eo
    # :248 (decoder :150-156; Dropout is identity in eval)
    h = torch.relu(F.linear(comb, P["blendshape_decoder.0.weight"], P["blendshape_decoder.0.bias"]))
    if "dec" in dm:                                    # nn.Dropout(dropout) of the decoder (:153), training mode
        h = h * dm["dec"] / (1.0 - dropout_p)
    z = F.linear(h, P["blendshape_decoder.3.weight"], P["blendshape_decoder.3.bias"]).squeeze(-1)
    bs = torch.sigmoid(z)
    # :252-253
    wm = torch.softmax(P["mel_weights"] / temperature, dim=0)
    we = torch.softmax(P["emotion_weights"] / temperature, dim=0)
    # :264-270
    final = torch.clamp(wm * bs * 0.5 + we * bs * 0.5, 0, 1)
    out = {"blendshapes": final}
    if return_attention:                               # :273-278
        mb = torch.zeros_like(bs)
        eb = torch.zeros_like(bs)
        mb[:, MOUTH_INDICES] = bs[:, MOUTH_INDICES]
        eb[:, EXPRESSION_INDICES] = bs[:, EXPRESSION_INDICES]
        out["mel_attention_weights"] = mw
        out["emotion_attention_weights"] = ew
        out["mel_blendshapes"] = mb
        out["emotion_blendshapes"] = eb
    if return_intermediates:
        out["_y"] = y
        out["_z"] = z
        out["_bs"] = bs
    return out


def core_forward_np(params, mel, short, emo, **kw) -> Dict[str, np.ndarray]:
    """numpy-in / numpy-out convenience wrapper."""
    with torch.no_grad():
        o = core_forward(params, mel, short, emo, **kw)
    return {k: v.detach().cpu().numpy() for k, v in o.items()}


# perceptual groups and weights of PerceptualBlendshapeLoss (src/model/losses.py:306-330)
PERCEPTUAL_GROUPS = (("mouth", range(12, 32), 2.0), ("eye", range(0, 12), 1.0), ("brow", range(32, 44), 1.0),
                     ("jaw", range(44, 52), 1.5))


def koemorph_loss(pred, target, mse_weight=1.0, l1_weight=0.1, perceptual_weight=0.5, temporal_weight=0.2,
                  sparsity_weight=0.01, smoothness_weight=0.1, landmark_weight=0.3, velocity_weight=0.05,
                  prev_pred=None, prev_target=None, landmark_w=None, audio_features=None):
    """KoeMorphLoss.forward restated (src/model/losses.py:89-178) for audio_features=None: the weighted sum of
    mse (:113-117), l1 (:119-123), perceptual = group-weighted MSEs (:326-338), temporal (:185-200), velocity
    (:202-217), sparsity (:219-224), smoothness = total variation along the 52 coefficients (:226-234) and landmark
    consistency through the fixed (136,52) matrix (:397-412).  Terms whose inputs are missing are skipped, as there."""
    total = pred.new_zeros(())
    if mse_weight > 0:
        total = total + mse_weight * F.mse_loss(pred, target)
    if l1_weight > 0:
        total = total + l1_weight * F.l1_loss(pred, target)
    if perceptual_weight > 0:
        per = pred.new_zeros(())
        for _, idx, w in PERCEPTUAL_GROUPS:
            idx = list(idx)
            per = per + w * F.mse_loss(pred[:, idx], target[:, idx])
        if audio_features is not None:                  # audio-visual consistency (:340-378): 1 - cos(mouth activation, audio energy)
            mouth = pred[:, list(range(12, 32))].mean(dim=1)
            energy = audio_features.norm(dim=2).mean(dim=1) if audio_features.dim() == 3 else audio_features.norm(dim=1)
            mouth_n = F.normalize(mouth.unsqueeze(0), dim=1).squeeze(0)
            audio_n = F.normalize(energy.unsqueeze(0), dim=1).squeeze(0)
            corr = F.cosine_similarity(mouth_n.unsqueeze(0), audio_n.unsqueeze(0))
            per = per + 0.5 * (1 - corr.mean())
        total = total + perceptual_weight * per
    if prev_pred is not None and prev_target is not None:
        if temporal_weight > 0:
            total = total + temporal_weight * F.mse_loss(pred - prev_pred, target - prev_target)
        if velocity_weight > 0:
            total = total + velocity_weight * F.l1_loss(pred - prev_pred, target - prev_target)
    if sparsity_weight > 0:
        total = total + sparsity_weight * pred.abs().mean()
    if smoothness_weight > 0:
        total = total + smoothness_weight * torch.diff(pred, dim=1).abs().mean()
    if landmark_weight > 0 and landmark_w is not None:
        total = total + landmark_weight * F.mse_loss(pred @ landmark_w.T, target @ landmark_w.T)
    return total


def dual_stream_loss(pred, target, l1_weight=1.0, l2_weight=0.1, velocity_weight=0.05, stream_separation_weight=0.01,
                     prev_predictions=None, with_attention: bool = True):
    """DualStreamLoss.forward restated (src/train_dual_stream.py:434-516): l1_weight * L1 + l2_weight * MSE (:478-483),
    + velocity_weight * MSE(pred - prev, target - prev) when prev_predictions is given (:489-495), + stream_separation_weight
    * mean | mean(pred[:, MOUTH]) - mean(pred[:, EXPRESSION]) | when both attention maps are passed (:498-514;
    ``with_attention`` stands for that condition).  PARITY UNPINNED: the module imports hydra (absent here), and its
    separation branch cannot run in the reference either (``from .dual_stream_attention import ...`` inside ``src/`` resolves
    to a module that does not exist) -- restated from the source text."""
    total = l1_weight * F.l1_loss(pred, target) + l2_weight * F.mse_loss(pred, target)
    if prev_predictions is not None and velocity_weight > 0:
        total = total + velocity_weight * F.mse_loss(pred - prev_predictions, target - prev_predictions)
    if with_attention and stream_separation_weight > 0:
        sep = (pred[:, MOUTH_INDICES].mean(dim=1) - pred[:, EXPRESSION_INDICES].mean(dim=1)).abs().mean()
        total = total + stream_separation_weight * sep
    return total


def core_dual_stream_loss_and_grads(params, mel, short, emo, target, prev_predictions, weights=None, num_heads=8,
                                    mel_sequence_length=256, dtype=torch.float32):
    """DualStreamLoss of the core's prediction and d loss / d param through torch.autograd on the restated forward."""
    P = {k: _t(v, dtype).clone().requires_grad_(True) for k, v in params.items()}
    out = core_forward(P, mel, short, emo, num_heads=num_heads, mel_sequence_length=mel_sequence_length, dtype=dtype)
    loss = dual_stream_loss(out["blendshapes"], _t(target, dtype),
                            prev_predictions=None if prev_predictions is None else _t(prev_predictions, dtype), **(weights or {}))
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy() for k, v in P.items()}
    return float(loss.detach()), grads, out["blendshapes"].detach().numpy()


def core_full_loss_and_grads(params, mel, short, emo, target, prev_pred, prev_target, landmark_w, weights=None,
                             num_heads=8, mel_sequence_length=256, dtype=torch.float32, audio_features=None,
                             dropout_p=0.0, drop_masks=None):
    """Full KoeMorphLoss (defaults of losses.py:36-47 unless `weights` overrides) of the core's prediction and
    d loss / d param through torch.autograd on the restated forward (eval mode)."""
    P = {k: _t(v, dtype).clone().requires_grad_(True) for k, v in params.items()}
    out = core_forward(P, mel, short, emo, num_heads=num_heads, mel_sequence_length=mel_sequence_length, dtype=dtype,
                       dropout_p=dropout_p, drop_masks=drop_masks)
    loss = koemorph_loss(out["blendshapes"], _t(target, dtype), prev_pred=_t(prev_pred, dtype),
                         prev_target=_t(prev_target, dtype), landmark_w=_t(landmark_w, dtype),
                         audio_features=None if audio_features is None else _t(audio_features, dtype), **(weights or {}))
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy() for k, v in P.items()}
    return float(loss.detach()), grads, out["blendshapes"].detach().numpy()


def core_loss_and_grads(params, mel, short, emo, target, num_heads=8,
                        mel_sequence_length=256, dtype=torch.float32, dropout_p=0.0, drop_masks=None):
    """MSE(blendshapes, target) and d loss / d param for every state-dict tensor, through
    torch.autograd on the restated forward (eval mode, dropout off, unless masks are given) -- golden G7."""
    P = {k: _t(v, dtype).clone().requires_grad_(True) for k, v in params.items()}
    out = core_forward(P, mel, short, emo, num_heads=num_heads,
                       mel_sequence_length=mel_sequence_length, dtype=dtype, dropout_p=dropout_p, drop_masks=drop_masks)
    loss = F.mse_loss(out["blendshapes"], _t(target, dtype))
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy()
             for k, v in P.items()}
    return float(loss.detach()), grads, out["blendshapes"].detach().numpy()
