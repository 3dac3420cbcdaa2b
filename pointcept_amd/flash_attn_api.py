"""Mirror of the one flash_attn entry point the reference calls
(pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:208-214; also m2/m3, LitePT):
    flash_attn.flash_attn_varlen_qkvpacked_func(qkv[T,3,H,D] bf16, cu_seqlens int32[S+1], max_seqlen,
                                                dropout_p=0.0, softmax_scale=None, causal=False) -> [T,H,D]
backed by the gfx950 MFMA window-attention kernels (attention.hip).  D must be 16, max_seqlen <= 1024,
dropout_p must be 0 (all reference PTv3 configs) -- anything else raises.
"""
from __future__ import annotations

from . import functional as PF
from ._lib import PtcoreError


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                     window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                                     return_attn_probs=False):
    if dropout_p not in (0, 0.0):
        raise PtcoreError("flash_attn_varlen_qkvpacked_func: dropout_p > 0 is not implemented")
    if causal or alibi_slopes is not None or softcap != 0.0 or tuple(window_size) != (-1, -1) or return_attn_probs:
        raise PtcoreError("flash_attn_varlen_qkvpacked_func: only plain non-causal attention is implemented")
    return PF.attn_varlen_qkvpacked(qkv, cu_seqlens, max_seqlen, softmax_scale)
