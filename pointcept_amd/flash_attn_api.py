"""Mirror of the one flash_attn entry point the reference calls
(pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:208-214; also m2/m3, LitePT):
    flash_attn.flash_attn_varlen_qkvpacked_func(qkv[T,3,H,D] bf16, cu_seqlens int32[S+1], max_seqlen,
                                                dropout_p=0.0, softmax_scale=None, causal=False) -> [T,H,D]
head_dim 16 and max_seqlen <= 1024 (every PT-v3m1 / m2 configuration) run on the gfx950 MFMA window-attention kernels
(attention.hip); head_dim 17..64 (PT-v3m3 / LitePT use 18: a multiple of 3 for their 3-D RoPE) on their multi-slab form
(attention_hd.h; windows up to 1024 keys for head_dim <= 32, 672 for <= 48, 512 for <= 64); head_dim 1..15 is zero-padded to 16
here (exact).  Anything else (head_dim > 64, longer windows) raises PtcoreError: there is no library (SDPA) backend behind this mirror (round 4).  fp16 qkv (LitePT's call site,
head_dim 18) runs on f16-operand instances of the multi-slab kernels (f16 MFMAs, P / dS rounded to f16, fp32 accumulation).  dropout_p > 0 (head_dim 16): attention dropout with flash-attn's semantics and the engine's own counter-based
mask (csrc/attention_drop.h; seed from torch's CPU generator).  causal / alibi / softcap / local windows raise.
"""
from __future__ import annotations

import torch

from . import functional as PF
from . import ops
from ._lib import PtcoreError


def _require_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise PtcoreError("flash_attn_varlen_qkvpacked_func: qkv must live on a GPU (there is no CPU fallback)")


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                     window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                                     return_attn_probs=False):
    dropout_p = float(dropout_p)
    if dropout_p != 0.0 and qkv.dim() == 4 and int(qkv.shape[3]) > 16:
        raise PtcoreError("flash_attn_varlen_qkvpacked_func: dropout_p > 0 is implemented for head_dim <= 16 (csrc/attention_drop.h)")
    if causal or alibi_slopes is not None or softcap != 0.0 or tuple(window_size) != (-1, -1) or return_attn_probs:
        raise PtcoreError("flash_attn_varlen_qkvpacked_func: only plain non-causal attention is implemented")
    if qkv.dim() != 4 or qkv.shape[1] != 3:
        raise PtcoreError(f"flash_attn_varlen_qkvpacked_func: qkv must be [T,3,H,D], got {tuple(qkv.shape)}")
    _require_gpu(qkv)
    d = int(qkv.shape[3])
    if 1 <= d < 16:
        # head_dim below one MFMA k-step (16): zero channels appended up to 16 and cut off the result -- exact (the appended products
        # are zeros in q.k; the appended output channels are sums of zeros), softmax_scale stays that of the caller's head_dim.  The pad
        # and the slice are two torch copies of [T, 3, H, 16] / [T, H, d]: no reference model has such heads, the mirror serves them
        # for flash-attn's other callers instead of raising.
        scale = d ** -0.5 if softmax_scale is None else softmax_scale
        out = flash_attn_varlen_qkvpacked_func(torch.nn.functional.pad(qkv, (0, 16 - d)), cu_seqlens, max_seqlen, dropout_p, scale)
        return out[..., :d]
    if not ops.attn_hd_supported(int(qkv.shape[3]), int(max_seqlen)):
        # no library (SDPA) backend behind this mirror: outside the kernels' range the call fails loudly
        raise PtcoreError(f"flash_attn_varlen_qkvpacked_func: head_dim={int(qkv.shape[3])} with max_seqlen={int(max_seqlen)} is outside the "
                          "window-attention kernels' range (head_dim 16..32: 1024 keys, ..48: 672, ..64: 512) -- PTC_EUNSUPPORTED")
    if qkv.dtype == torch.float16 and int(qkv.shape[3]) == 16 and dropout_p != 0.0:
        # fp16 operands at head_dim 16: the head_dim-16 kernels compute in bf16 (what every PT-v3 call site asks for by casting first);
        # an fp16 caller of THIS shape gets its operands re-rounded to bf16 (three mantissa bits) and fp16 back -- stated deviation.
        # Without dropout the f16-I/O instances do both roundings in their load / store paths (same bits, no cast passes: below); the
        # dropout kernels have bf16 I/O only.  head_dim 17..64 (LitePT, litept_v1.py:259-265) runs f16-operand instances: no re-rounding.
        return PF.attn_varlen_qkvpacked(qkv.to(torch.bfloat16), cu_seqlens, max_seqlen, softmax_scale, dropout_p).to(torch.float16)
    return PF.attn_varlen_qkvpacked(qkv, cu_seqlens, max_seqlen, softmax_scale, dropout_p)
