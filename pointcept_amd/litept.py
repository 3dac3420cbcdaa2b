"""LitePT-v1 on the engine: module-level drop-in for pointcept/models/litept/litept_v1.py (registry name "LitePT-v1", ctor kwargs
:595-625, forward :756-778, same state-dict keys / shapes).  SURVEY 8(f) rank 2.

LitePT = convolution blocks in the fine stages, attention blocks with `PointROPE` (libs/pointrope: 3-axis rotary embedding from the
INTEGER grid coordinates) in the coarse ones, grid pooling between them, a decoder that by default only un-pools:

  Embedding           (:561-591)  k = 5 submanifold conv + BatchNorm + GELU                    = m1's Embedding
  Block               (:303-401)  [conv: SubMConv3d + Linear + LayerNorm, residual | norm0]  then, if enable_attn,
                                  norm1 -> PointROPEAttention -> DropPath, residual; norm2 -> MLP -> DropPath, residual
  PointROPEAttention  (:128-274)  qkv -> [order] -> q, k rotated by PointROPE(grid_coord[order]) -> flash-attn -> [inverse] -> proj
  GridPooling         (:404-516)  m2's pooling (unique cells -> CSR -> segment reduce) + `mask`; the child is re-serialized only where
                                  the next stage has attention (`re_serialization`)
  GridUnpooling       (:519-558)  m2's

Engine mapping.  The qkv GEMM writes the padded, serialized rows through its gather table; the rotation runs on the packed
[n, 3, H, D] rows and emits the bf16 operand of the window-attention kernels (point_transformer_v3m3.RopeAttention).  The angle of
libs/pointrope/kernels.cu:44-54 is pos * (F0 / base^(i/Q)): exactly `xyz * inv_freq` with xyz = grid_coord.float() (integers below
2^24 are exact) and inv_freq[i] = F0 / base^(i/Q), so the same rotation code serves LitePT and PT-v3m3.
Precision note: the reference feeds flash-attn fp16 operands (`qkv.half()`, :235-243); the engine's window attention takes bf16
operands with fp32 accumulation (as PT-v3m1's call site does, ptv3m1:209), so q / k / v lose three mantissa bits against the
reference here -- inside the tolerance every attention parity test of this repo states (2e-2 on features).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as PF
from . import nn as PNN
from . import spconv_api as spconv
from .point_transformer_v3 import MLP, DropPath, Embedding, PointModule, PointSequential  # noqa: F401  (Embedding = :561-591)
from .point_transformer_v3m2 import GridPooling as _GridPoolingM2
from .point_transformer_v3m2 import GridUnpooling as _GridUnpoolingM2
from .point_transformer_v3m3 import RopeAttention
from .pointrope_api import PointROPE
from .structure import Point


class PointROPEAttention(RopeAttention):
    """:128-274 (always the flash branch: no RPE, no upcasts)"""

    def __init__(self, channels, num_heads, patch_size, rope_freq, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0,
                 order_index=0):
        super().__init__(channels, num_heads, patch_size, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop,
                         order_index=order_index, enable_rpe=False, enable_flash=True, upcast_attention=False, upcast_softmax=False)
        if (channels // num_heads) % 6 != 0:
            raise PF.PtcoreError(f"PointROPE needs head_dim % 6 == 0 (libs/pointrope/kernels.cu:87), got {channels // num_heads}")
        self.rope = PointROPE(freq=rope_freq)          # no parameters / buffers (:48-52)
        self._inv_freq = {}

    def _rope_on(self) -> bool:
        return True

    def _operand_dtype(self, qkv_dtype):
        """litept_v1.py:239-265: the rotated q / k come back in q.dtype and go to flash-attn uncast -- f16 under the reference's fp16
        autocast recipe (f16-operand instances of the window-attention kernels), bf16 under bf16 autocast"""
        return torch.float16 if qkv_dtype == torch.float16 else torch.bfloat16

    def _rope_inputs(self, point, order):
        key = f"_ptc_rope_pos_{self.order_index}"
        if key not in point.keys():
            point[key] = point.grid_coord[order].float()                         # :232-233 (positions of the padded, serialized rows)
        dev = point.feat.device
        f = self._inv_freq.get(dev)
        if f is None:
            Q = (self.channels // self.num_heads) // 6
            f = self.rope.F0 / (self.rope.base ** (torch.arange(Q, dtype=torch.float32) / Q))     # kernels.cu:44
            f = self._inv_freq[dev] = f.to(dev)
        return point[key], f


class Block(PointModule):
    """:303-401"""

    def __init__(self, channels, num_heads, patch_size=48, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0,
                 drop_path=0.0, norm_layer=nn.LayerNorm, act_layer=nn.GELU, pre_norm=True, order_index=0, cpe_indice_key=None,
                 enable_conv=True, enable_attn=True, rope_freq=100.0):
        super().__init__()
        self.channels, self.pre_norm = channels, pre_norm
        self.enable_conv, self.enable_attn = enable_conv, enable_attn
        if enable_conv:
            self.conv = PointSequential(spconv.SubMConv3d(channels, channels, kernel_size=3, bias=True, indice_key=cpe_indice_key),
                                        PNN.Linear(channels, channels), norm_layer(channels))
        else:
            self.norm0 = PointSequential(norm_layer(channels))
        if enable_attn:
            self.norm1 = PointSequential(norm_layer(channels))
            self.attn = PointROPEAttention(channels=channels, patch_size=patch_size, rope_freq=rope_freq, num_heads=num_heads,
                                           qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop,
                                           order_index=order_index)
            self.norm2 = PointSequential(norm_layer(channels))
            self.mlp = PointSequential(MLP(in_channels=channels, hidden_channels=int(channels * mlp_ratio), out_channels=channels,
                                           act_layer=act_layer, drop=proj_drop))
            self.drop_path = PointSequential(DropPath(drop_path) if drop_path > 0.0 else nn.Identity())

    def forward(self, point: Point):
        if self.enable_conv:
            shortcut = point.feat
            point = self.conv(point)
            point.feat = shortcut + point.feat
        else:
            point = self.norm0(point)
        if self.enable_attn:
            shortcut = point.feat
            if self.pre_norm:
                point = self.norm1(point)
            point = self.drop_path(self.attn(point))
            point.feat = shortcut + point.feat
            if not self.pre_norm:
                point = self.norm1(point)
            shortcut = point.feat
            if self.pre_norm:
                point = self.norm2(point)
            point = self.drop_path(self.mlp(point))
            point.feat = shortcut + point.feat
            if not self.pre_norm:
                point = self.norm2(point)
        point.sparse_conv_feat = point.sparse_conv_feat.replace_feature(point.feat)
        return point


class GridPooling(_GridPoolingM2):
    """:404-516"""
    trace_idx_ptr = False

    def __init__(self, in_channels, out_channels, stride=2, norm_layer=None, act_layer=None, reduce="max", shuffle_orders=True,
                 traceable=True, re_serialization=False, serialization_order="z"):
        super().__init__(in_channels, out_channels, stride=stride, norm_layer=norm_layer, act_layer=act_layer, reduce=reduce,
                         shuffle_orders=shuffle_orders, traceable=traceable)
        self.re_serialization, self.serialization_order = re_serialization, serialization_order

    def _extra_keys(self, point, point_dict, order0, idx_ptr):
        if "mask" in point.keys():                                                                 # :488-494
            point_dict["mask"] = PF.segment_csr(point.mask.float()[:, None], idx_ptr, "mean", perm=order0)[:, 0] > 0.5

    def _serialize_child(self, point, child):
        if self.re_serialization:                                                                  # :507-510
            child.serialization(order=self.serialization_order, shuffle_orders=self.shuffle_orders)


class GridUnpooling(_GridUnpoolingM2):
    """:519-558"""

    def __init__(self, in_channels, skip_channels, out_channels, norm_layer=None, act_layer=None, traceable=False):
        super().__init__(in_channels, skip_channels, out_channels, norm_layer=norm_layer, act_layer=act_layer, traceable=traceable)

    def forward(self, point):
        inverse = point.pooling_inverse
        parent = super().forward(point)
        if self.traceable:
            parent["unpooling_inverse"] = inverse                                                  # :556
        return parent


class LitePT(PointModule):
    """registry name "LitePT-v1" (:593)"""

    def __init__(self, in_channels=4, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2, 2, 2), enc_depths=(2, 2, 2, 6, 2),
                 enc_channels=(36, 72, 144, 252, 504), enc_num_head=(2, 4, 8, 14, 28), enc_patch_size=(1024, 1024, 1024, 1024, 1024),
                 enc_conv=(True, True, True, False, False), enc_attn=(False, False, False, True, True),
                 enc_rope_freq=(100.0, 100.0, 100.0, 100.0, 100.0), dec_depths=(0, 0, 0, 0), dec_channels=(72, 72, 144, 252),
                 dec_num_head=(4, 4, 8, 14), dec_patch_size=(1024, 1024, 1024, 1024), dec_conv=(False, False, False, False),
                 dec_attn=(False, False, False, False), dec_rope_freq=(100.0, 100.0, 100.0, 100.0), mlp_ratio=4, qkv_bias=True,
                 qk_scale=None, attn_drop=0.0, proj_drop=0.0, drop_path=0.3, pre_norm=True, shuffle_orders=True, enc_mode=False):
        super().__init__()
        self.num_stages = len(enc_depths)
        self.order = [order] if isinstance(order, str) else order
        self.enc_mode, self.shuffle_orders = enc_mode, shuffle_orders
        self.enc_conv, self.enc_attn, self.dec_conv, self.dec_attn = enc_conv, enc_attn, dec_conv, dec_attn
        assert self.num_stages == len(stride) + 1 == len(enc_channels) == len(enc_num_head) == len(enc_patch_size)
        assert self.enc_mode or self.num_stages == len(dec_depths) + 1 == len(dec_channels) + 1
        assert self.enc_mode or self.num_stages == len(dec_num_head) + 1 == len(dec_patch_size) + 1
        bn_layer = lambda c: PNN.BatchNorm1d(c, eps=1e-3, momentum=0.01)  # noqa: E731  (:646)
        ln_layer, act_layer = PNN.LayerNorm, PNN.GELU
        blk = dict(mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop, norm_layer=ln_layer,
                   act_layer=act_layer, pre_norm=pre_norm)
        self.embedding = Embedding(in_channels=in_channels, embed_channels=enc_channels[0], norm_layer=bn_layer, act_layer=act_layer)
        enc_dp = [x.item() for x in torch.linspace(0, drop_path, sum(enc_depths))]
        self.enc = PointSequential()
        for s in range(self.num_stages):
            dp = enc_dp[sum(enc_depths[:s]):sum(enc_depths[:s + 1])]
            enc = PointSequential()
            if s > 0:
                enc.add(GridPooling(in_channels=enc_channels[s - 1], out_channels=enc_channels[s], stride=stride[s - 1], norm_layer=bn_layer,
                                    act_layer=act_layer, re_serialization=enc_attn[s], serialization_order=self.order), name="down")
            for i in range(enc_depths[s]):
                enc.add(Block(channels=enc_channels[s], num_heads=enc_num_head[s], patch_size=enc_patch_size[s], drop_path=dp[i],
                              order_index=i % len(self.order), cpe_indice_key=f"stage{s}", enable_conv=enc_conv[s], enable_attn=enc_attn[s],
                              rope_freq=enc_rope_freq[s], **blk), name=f"block{i}")
            if len(enc) != 0:
                self.enc.add(module=enc, name=f"enc{s}")
        if not self.enc_mode:
            dec_dp = [x.item() for x in torch.linspace(0, drop_path, sum(dec_depths))]
            self.dec = PointSequential()
            dec_channels = list(dec_channels) + [enc_channels[-1]]
            for s in reversed(range(self.num_stages - 1)):
                dp = dec_dp[sum(dec_depths[:s]):sum(dec_depths[:s + 1])]
                dp.reverse()
                dec = PointSequential()
                dec.add(GridUnpooling(in_channels=dec_channels[s + 1], skip_channels=enc_channels[s], out_channels=dec_channels[s],
                                      norm_layer=bn_layer, act_layer=act_layer), name="up")
                for i in range(dec_depths[s]):
                    dec.add(Block(channels=dec_channels[s], num_heads=dec_num_head[s], patch_size=dec_patch_size[s], drop_path=dp[i],
                                  order_index=i % len(self.order), cpe_indice_key=f"stage{s}", enable_conv=dec_conv[s],
                                  enable_attn=dec_attn[s], rope_freq=dec_rope_freq[s], **blk), name=f"block{i}")
                self.dec.add(module=dec, name=f"dec{s}")

    def forward(self, data_dict):
        point = Point(data_dict)
        if self.enc_attn[0]:
            point.serialization(order=self.order, shuffle_orders=self.shuffle_orders)                 # :769-770
        point.sparsify()
        point = self.embedding(point)
        point = self.enc(point)
        if not self.enc_mode:
            point = self.dec(point)
        return point
