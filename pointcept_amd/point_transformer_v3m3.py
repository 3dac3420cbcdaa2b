"""PT-v3m3 (Utonia backbone) on the engine: module-level drop-in for
pointcept/models/point_transformer_v3/point_transformer_v3m3_utonia.py (registry name "PT-v3m3", ctor kwargs :688-723,
forward :907-917, same state-dict keys / shapes incl. the `attn.rope.inv_freq` buffers).  SURVEY 8(f) rank 2.

m3 = m2 (Linear stem, GridPooling / GridUnpooling with re-serialization, LayerScale: point_transformer_v3m2.py) plus a 3-axis rotary
embedding of q and k from the CONTINUOUS point coordinates inside every attention (`Point3DRoPE`, :43-102; call site :274-331):

    rope_coord = coord[order]  (+ in training: one random shift / per-axis jitter / global rescale per call, :276-300)
    q, k       = rope(q, k, rope_coord)            fp32 arithmetic on the Linear's output
    flash-attn reads stack([q, k, v]).to(bf16)

Engine mapping: the qkv GEMM already writes the padded, serialized rows (gather table folded in, as m1); the rotation runs on that
packed [n, 3, H, D] buffer and emits the bf16 operand of the window-attention kernels (`PF.rope_xyz_qkvpacked`: ptc_rope3d_xyz); head_dim % 6 == 0 is what the
rotation needs (18 in the reference's Utonia configs -> the slab kernels of csrc/attention_hd.h).  The augmentation draws use the same
torch calls in the same order as the reference, so a seeded run consumes the device RNG identically.
`dec_rope_enable=False` builds decoder blocks with rope_base=None (:862-887): plain m2 attention.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as PF
from .point_transformer_v3 import SerializedAttention as _AttnM1
from .point_transformer_v3m2 import Block as _BlockM2
from .point_transformer_v3m2 import Embedding, GridPooling, GridUnpooling, LayerScale  # noqa: F401  (same classes in m3)
from .point_transformer_v3m2 import PointTransformerV3 as _ModelM2


class Point3DRoPE(nn.Module):
    """point_transformer_v3m3_utonia.py:43-102.  A head is three chunks of head_dim / 3 (x, y, z); inside a chunk element i of the
    first half rotates with element i of the second half by the angle coord[axis] * inv_freq[i]."""

    def __init__(self, head_dim, base=10000):
        super().__init__()
        assert head_dim % 3 == 0, f"Head dimension must be divisible by 3 for 3D RoPE, {head_dim}"
        self.head_dim, self.chunk_dim, self.base = head_dim, head_dim // 3, base
        inv_freq = 1.0 / (self.base ** (torch.arange(0, self.chunk_dim, 2).float() / self.chunk_dim))      # :53-55
        self.register_buffer("inv_freq", inv_freq)

    def forward(self, q, k, xyz):
        """q, k [n, H, D] -> rotated (fp32 for 16-bit inputs, like the reference's `q * cos` promotion)."""
        packed = torch.stack((q, k, k), dim=1)
        out = self._rotate_packed(packed, xyz)
        return out[:, 0], out[:, 1]

    def _rotate_packed(self, qkv, xyz):
        n, _, H, D = qkv.shape
        Q = D // 6
        emb = xyz[:, :, None] * self.inv_freq[None, None, :]
        cos, sin = emb.cos()[:, None, None, :, None, :], emb.sin()[:, None, None, :, None, :]
        t = qkv[:, :2].float().reshape(n, 2, H, 3, 2, Q)
        u, v = t[..., 0:1, :], t[..., 1:2, :]
        rot = torch.cat((u * cos + (-v) * sin, v * cos + u * sin), dim=-2).reshape(n, 2, H, D)
        return torch.cat((rot, qkv[:, 2:].float()), dim=1)


class RopeAttention(_AttnM1):
    """m1's serialized attention with a rotation of q and k between the qkv GEMM and the window-attention kernels.  A derived class
    says whether the rotation is on (`_rope_on`) and supplies the rows' positions and frequencies (`_rope_inputs`)."""

    def _rope_on(self) -> bool:
        raise NotImplementedError

    def _rope_inputs(self, point, order):
        """-> (xyz [n, 3] fp32 positions of the padded, serialized rows; inv_freq [head_dim / 6] fp32)"""
        raise NotImplementedError

    def _operand_dtype(self, qkv_dtype):
        """dtype of the tensor the call site hands to flash-attn: PT-v3m3 casts to bf16 itself (utonia.py:353)"""
        return torch.bfloat16

    def _rotate_for_dense(self, qkv, point, order):
        """[n, 3, H, D] -> fp32 rotated q, k (dense branch)"""
        xyz, inv_freq = self._rope_inputs(point, order)
        n, _, H, D = qkv.shape
        Q = D // 6
        emb = xyz[:, :, None] * inv_freq[None, None, :]
        cos, sin = emb.cos()[:, None, None, :, None, :], emb.sin()[:, None, None, :, None, :]
        t = qkv[:, :2].float().reshape(n, 2, H, 3, 2, Q)
        u, v = t[..., 0:1, :], t[..., 1:2, :]
        rot = torch.cat((u * cos + (-v) * sin, v * cos + u * sin), dim=-2).reshape(n, 2, H, D)
        return rot[:, 0], rot[:, 1]

    def _forward_dense_rope(self, point):
        """enable_flash=False with the rotation (utonia.py:303-318): rotation on [n, H, D], then the dense [P, H, K, K] formulation."""
        H, K, C = self.num_heads, self.patch_size, self.channels
        pad, unpad, _ = self.get_padding_and_inverse(point)
        order = point.serialized_order[self.order_index][pad]
        inverse = unpad[point.serialized_inverse[self.order_index]]
        qkv = self.qkv(point.feat)[order].reshape(-1, 3, H, C // H)
        q, k = self._rotate_for_dense(qkv, point, order)
        q, k = (t.reshape(-1, K, H, C // H).permute(0, 2, 1, 3) for t in (q, k))
        v = qkv[:, 2].reshape(-1, K, H, C // H).permute(0, 2, 1, 3)
        if self.upcast_attention:
            q, k = q.float(), k.float()
        attn = (q * self.scale) @ k.transpose(-2, -1)
        if self.rpe is not None:
            attn = attn + self.rpe(self.get_rel_pos(point, order))
        if self.upcast_softmax:
            attn = attn.float()
        attn = torch.softmax(attn, dim=-1)
        attn = self.attn_drop(attn).to(qkv.dtype)
        feat = (attn @ v).transpose(1, 2).reshape(-1, C)[inverse]
        point.feat = self.proj_drop(self.proj(feat))
        return point

    def forward(self, point):
        if not self._rope_on():
            return super().forward(point)
        if not self.enable_flash:
            _, offset_host = point._host_facts()
            smallest = min(b - a for a, b in zip([0] + list(offset_host[:-1]), offset_host))
            self.patch_size = min(int(smallest), self.patch_size_max)      # utonia.py:258-261
            if not self._kernel_ok():
                return self._forward_dense_rope(point)
        H, K, C = self.num_heads, self.patch_size, self.channels
        pad, _, cu_seqlens = self.get_padding_and_inverse(point)
        gidx, inv, dup_of_point, gidx_primary, tabs = self._index_maps(point)
        order = point.serialized_order[self.order_index][pad] if gidx is None else gidx
        xyz, inv_freq = self._rope_inputs(point, order)
        if tabs is not None:
            qkv_s = self.qkv(point.feat, tabs[0], tabs[1])                                     # padded, serialized rows (:272)
            # rotation (:303-305,319-321) + window attention as one operator where the kernels fuse them (head_dim 18)
            out = PF.attn_rope_qkvpacked(qkv_s.reshape(-1, 3, H, C // H), xyz, inv_freq, cu_seqlens, K, self.scale, self._operand_dtype(qkv_s.dtype))
            feat = self.proj(out.reshape(-1, C).to(qkv_s.dtype), tabs[2], tabs[3])
        else:
            qkv = self.qkv(point.feat)
            qkv_s = PF.gather_rows(qkv, gidx, inv, dup_of_point)
            out = PF.attn_rope_qkvpacked(qkv_s.reshape(-1, 3, H, C // H), xyz, inv_freq, cu_seqlens, K, self.scale, self._operand_dtype(qkv_s.dtype))
            feat = self.proj(PF.gather_rows(out.reshape(-1, C), inv, gidx_primary).to(qkv.dtype))
        point.feat = self.proj_drop(feat)
        return point


class SerializedAttention(RopeAttention):
    """:125-367.  rope_base falsy = m1 / m2 attention."""

    def __init__(self, channels, num_heads, patch_size, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0, order_index=0,
                 enable_rpe=False, enable_flash=True, upcast_attention=True, upcast_softmax=True, rope_base=10000, shift_coords=None,
                 jitter_coords=None, rescale_coords=None):
        super().__init__(channels, num_heads, patch_size, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop,
                         order_index=order_index, enable_rpe=enable_rpe, enable_flash=enable_flash, upcast_attention=upcast_attention,
                         upcast_softmax=upcast_softmax)
        self.rope_base = rope_base
        if rope_base:
            head_dim = channels // num_heads
            if head_dim % 6 != 0:
                # the reference asserts % 3 (:46-48) and then splits each chunk in two halves: an odd chunk fails there at run time
                raise PF.PtcoreError(f"Point3DRoPE needs head_dim % 6 == 0 (three chunks of two halves), got {head_dim}")
            self.rope = Point3DRoPE(head_dim=head_dim, base=rope_base)
            self.shift_coords, self.jitter_coords, self.rescale_coords = shift_coords, jitter_coords, rescale_coords

    # ---- the rows' coordinates, with the training-time augmentation of :276-300 -------------------------------------------
    def _rope_coord(self, point, order):
        rope_coord = point.coord[order].clone()
        if self.training:
            dd = {"device": rope_coord.device, "dtype": rope_coord.dtype}
            if self.shift_coords is not None and self.shift_coords > 0:
                rope_coord += torch.empty(3, **dd).uniform_(-self.shift_coords, self.shift_coords).unsqueeze(0)
            if self.jitter_coords is not None and self.jitter_coords > 1.0:
                jitter_max = math.log(self.jitter_coords)
                rope_coord *= torch.empty(3, **dd).uniform_(-jitter_max, jitter_max).exp().unsqueeze(0)
            if self.rescale_coords is not None and self.rescale_coords > 1.0:
                rescale_max = math.log(self.rescale_coords)
                rope_coord *= torch.empty(1, **dd).uniform_(-rescale_max, rescale_max).exp()
        return rope_coord

    def _rope_on(self) -> bool:
        return bool(self.rope_base)

    def _rope_inputs(self, point, order):
        return self._rope_coord(point, order), self.rope.inv_freq


class Block(_BlockM2):
    """:396-506: m2's block with the RoPE attention."""

    def __init__(self, channels, num_heads, patch_size=48, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0, drop_path=0.0, layer_scale=None, norm_layer=nn.LayerNorm, act_layer=nn.GELU, pre_norm=True,
                 order_index=0, cpe_indice_key=None, enable_rpe=False, enable_flash=True, upcast_attention=True,
                 upcast_softmax=True, rope_base=10000, shift_coords=None, jitter_coords=None, rescale_coords=None):
        self._rope_kw = dict(rope_base=rope_base, shift_coords=shift_coords, jitter_coords=jitter_coords, rescale_coords=rescale_coords)
        super().__init__(channels, num_heads, patch_size=patch_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                         attn_drop=attn_drop, proj_drop=proj_drop, drop_path=drop_path, layer_scale=layer_scale, norm_layer=norm_layer,
                         act_layer=act_layer, pre_norm=pre_norm, order_index=order_index, cpe_indice_key=cpe_indice_key,
                         enable_rpe=enable_rpe, enable_flash=enable_flash, upcast_attention=upcast_attention,
                         upcast_softmax=upcast_softmax)

    def _build_attn(self, **kw):
        return SerializedAttention(**kw, **self._rope_kw)


class PointTransformerV3(_ModelM2):
    """registry name "PT-v3m3" (:686)"""
    block_cls = Block

    def __init__(self, in_channels=6, order=("z", "z-trans"), stride=(2, 2, 2, 2), enc_depths=(2, 2, 2, 6, 2),
                 enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32), enc_patch_size=(48, 48, 48, 48, 48),
                 dec_depths=(2, 2, 2, 2), dec_channels=(64, 64, 128, 256), dec_num_head=(4, 4, 8, 16), dec_patch_size=(48, 48, 48, 48),
                 dec_rope_enable=True, mlp_ratio=4, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0, drop_path=0.3,
                 layer_scale=None, pre_norm=True, shuffle_orders=True, enable_rpe=False, enable_flash=True, upcast_attention=False,
                 upcast_softmax=False, traceable=False, mask_token=False, enc_mode=False, freeze_encoder=False, rope_base=10000,
                 shift_coords=None, jitter_coords=None, rescale_coords=None):
        self._rope_kw = dict(rope_base=rope_base, shift_coords=shift_coords, jitter_coords=jitter_coords, rescale_coords=rescale_coords)
        self._dec_rope_enable = dec_rope_enable
        super().__init__(in_channels=in_channels, order=order, stride=stride, enc_depths=enc_depths, enc_channels=enc_channels,
                         enc_num_head=enc_num_head, enc_patch_size=enc_patch_size, dec_depths=dec_depths, dec_channels=dec_channels,
                         dec_num_head=dec_num_head, dec_patch_size=dec_patch_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                         qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop, drop_path=drop_path, layer_scale=layer_scale,
                         pre_norm=pre_norm, shuffle_orders=shuffle_orders, enable_rpe=enable_rpe, enable_flash=enable_flash,
                         upcast_attention=upcast_attention, upcast_softmax=upcast_softmax, traceable=traceable, mask_token=mask_token,
                         enc_mode=enc_mode, freeze_encoder=freeze_encoder)

    def _extra_block_kw(self, decoder: bool) -> dict:
        if decoder and not self._dec_rope_enable:
            return dict(rope_base=None)                      # :862-887
        return dict(self._rope_kw)
