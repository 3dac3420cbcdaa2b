"""PT-v3m1 on the engine: drop-in for
pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py (registry name "PT-v3m1",
same constructor kwargs (:520-552), same state-dict keys/shapes, same Point
protocol in and out), with the hot path on libptcore.so:

  Point.serialization / sparsify      -> fused key kernel + batched radix sort         (structure.py)
  SerializedAttention                 -> device pad maps, gather kernel, MFMA window attention
                                         (replaces the python loop :142-164 and flash_attn :208-214)
  SerializedPooling / Unpooling       -> scan-based cluster maps, fused gather+segment reduce,
                                         gather-form backward (replaces torch.unique/sort + torch_scatter)
  CPE / stem SubMConv3d               -> hash rulebook + MFMA implicit GEMM             (spconv_api.py)
  nn.Linear / nn.LayerNorm            -> tall-skinny MFMA GEMMs with split-K weight gradients, the
                                         serialization gather folded into the qkv / proj GEMMs;
                                         one-pass LayerNorm                             (nn.py)
BatchNorm / GELU / residual adds stay on PyTorch-ROCm (ATen).

Behaviours of the reference that change numerics are reproduced on purpose:
stale sparse_conv_feat in the first decoder block's CPE (D.1), per-point DropPath (D.2), CPU-RNG
order shuffling (D.3), bf16 attention regardless of the AMP dtype (D.4).
Not implemented (raise): head_dim outside 16..64 or attention dropout with enable_flash=True.  PDNorm (pdnorm_bn / pdnorm_ln, the
PPT multi-dataset configs) selects among the engine's own per-condition norm layers; those blocks take the unfused path.
`enable_flash=False` follows the reference's patch-size rule (min(smallest scene, patch_size), ptv3m1:173-176) and runs the
same attention kernel when it can (no RPE / dropout, head_dim 16..64; bf16 operands), else the dense [P,H,K,K] branch of
:190-206 (RPE bias, upcasts, dropout) in torch ops on the GPU.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from . import config
from . import functional as PF
from . import nn as PNN
from . import ops
from . import spconv_api as spconv
from ._lib import PtcoreError
from .structure import AttrDict, Point


class PointModule(nn.Module):
    """modules taking / returning a Point (pointcept/models/modules.py:27-34)"""


class PointSequential(PointModule):
    """type-dispatching container, pointcept/models/modules.py:36-111"""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        # a (BatchNorm1d, GELU | ReLU) pair that is adjacent in the LIVE module list runs as one call (PNN.fused_act)
        for module, act in PNN.plain_feature_runs(self._modules.values()):
            if isinstance(module, PointModule):
                input = module(input)
            elif spconv.is_spconv_module(module):
                if isinstance(input, Point):
                    input.sparse_conv_feat = module(input.sparse_conv_feat)
                    input.feat = input.sparse_conv_feat.features
                else:
                    input = module(input)
            else:
                run = module if act is None else partial(module, act=act)
                if isinstance(input, Point):
                    input.feat = run(input.feat)
                    if "sparse_conv_feat" in input.keys():
                        input.sparse_conv_feat = input.sparse_conv_feat.replace_feature(input.feat)
                elif isinstance(input, spconv.SparseConvTensor):
                    if input.indices.shape[0] != 0:
                        input = input.replace_feature(run(input.features))
                else:
                    input = run(input)
        return input


def norm_then_act(owner, point):
    """`point = owner.act(owner.norm(point))` of the pooling modules (ptv3m1:437-440), as ONE BatchNorm pass when both are
    single-module PointSequentials holding a fusable pair right now (PNN.fused_act), else exactly the two calls."""
    norm, act = getattr(owner, "norm", None), getattr(owner, "act", None)
    if (type(norm) is PointSequential and type(act) is PointSequential and len(norm) == 1 and len(act) == 1
            and not PNN._hooked(norm) and not PNN._hooked(act)):
        kind = PNN.fused_act(norm[0], act[0])
        if kind is not None:
            point.feat = norm[0](point.feat, act=kind)
            if "sparse_conv_feat" in point.keys():
                point.sparse_conv_feat = point.sparse_conv_feat.replace_feature(point.feat)
            return point
    if norm is not None:
        point = norm(point)
    if act is not None:
        point = act(point)
    return point


class PDNorm(PointModule):
    """Prompt-driven normalisation of the multi-dataset (PPT) configs,
    pointcept/models/point_prompt_training/prompt_driven_normalization.py:8-49: one norm layer per dataset condition
    (`decouple`), selected by `point.condition`, optionally modulated by `point.context` (`adaptive`:
    feat * (1 + scale) + shift with (shift, scale) = Linear(SiLU(context))).  Same attribute names, so the state dict
    (`norm.{i}.*`, `modulation.1.*`) is the reference's."""

    def __init__(self, num_features, norm_layer, context_channels=256, conditions=("ScanNet", "S3DIS", "Structured3D"),
                 decouple=True, adaptive=False):
        super().__init__()
        self.conditions, self.decouple, self.adaptive = conditions, decouple, adaptive
        if self.decouple:
            self.norm = nn.ModuleList([norm_layer(num_features) for _ in conditions])
        else:
            self.norm = norm_layer     # as the reference: the factory itself (only meaningful with decouple=True)
        if self.adaptive:
            self.modulation = nn.Sequential(nn.SiLU(), nn.Linear(context_channels, 2 * num_features, bias=True))

    def forward(self, point):
        assert {"feat", "condition"}.issubset(point.keys())
        condition = point.condition if isinstance(point.condition, str) else point.condition[0]
        if self.decouple:
            assert condition in self.conditions
            norm = self.norm[self.conditions.index(condition)]
        else:
            norm = self.norm
        point.feat = norm(point.feat)
        if self.adaptive:
            assert "context" in point.keys()
            shift, scale = self.modulation(point.context).chunk(2, dim=1)
            point.feat = point.feat * (1.0 + scale) + shift
        return point


class DropPath(nn.Module):
    """timm.layers.DropPath: drops rows of dim 0, i.e. individual points for [N,C] features."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class RPE(nn.Module):
    """Relative position bias of the dense attention branch (ptv3m1:29-48): one learned [2 bnd + 1, H] table per axis
    (stored stacked, state-dict key `rpe_table` [3 (2 bnd + 1), H]), looked up by the clamped grid offset between the
    two points of a pair and summed over x, y, z.  bnd = int((4 K)^(1/3) * 2)."""

    def __init__(self, patch_size, num_heads):
        super().__init__()
        self.patch_size, self.num_heads = patch_size, num_heads
        self.pos_bnd = int((4 * patch_size) ** (1 / 3) * 2)
        self.rpe_num = 2 * self.pos_bnd + 1
        self.rpe_table = nn.Parameter(torch.zeros(3 * self.rpe_num, num_heads))
        nn.init.trunc_normal_(self.rpe_table, std=0.02)

    def forward(self, coord):                      # [P, K, K, 3] integer offsets -> [P, H, K, K]
        idx = coord.clamp(-self.pos_bnd, self.pos_bnd) + self.pos_bnd
        bias = self.rpe_table[idx[..., 0]]
        bias = bias + self.rpe_table[idx[..., 1] + self.rpe_num]
        bias = bias + self.rpe_table[idx[..., 2] + 2 * self.rpe_num]
        return bias.permute(0, 3, 1, 2)


class SerializedAttention(PointModule):
    """ptv3m1:51-222.  enable_flash=True: the MFMA window-attention kernels (attention.hip) behind the
    flash_attn_varlen_qkvpacked_func contract.  enable_flash=False: the reference's second branch -- the patch size
    shrinks to the smallest scene of the batch (:173-176) so every patch is full; without RPE / attention dropout /
    head_dim outside 16..64 that is still the same kernel (bf16 operands, fp32 accumulation), otherwise the dense
    [P, H, K, K] formulation of :190-206 in torch ops on the GPU (RPE bias, upcasts, dropout)."""

    def __init__(self, channels, num_heads, patch_size, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0,
                 order_index=0, enable_rpe=False, enable_flash=True, upcast_attention=True, upcast_softmax=True):
        super().__init__()
        assert channels % num_heads == 0
        self.channels, self.num_heads = channels, num_heads
        self.scale = qk_scale or (channels // num_heads) ** -0.5
        self.order_index = order_index
        self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
        self.enable_rpe, self.enable_flash = enable_rpe, enable_flash
        if enable_flash:
            assert enable_rpe is False, "Set enable_rpe to False when enable Flash Attention"               # ptv3m1:78-86
            assert upcast_attention is False, "Set upcast_attention to False when enable Flash Attention"
            assert upcast_softmax is False, "Set upcast_softmax to False when enable Flash Attention"
            if not 16 <= channels // num_heads <= 64:
                raise PtcoreError(f"engine flash attention needs 16 <= head_dim <= 64, got {channels // num_heads}")
            if attn_drop != 0.0 and channels // num_heads != 16:      # csrc/attention_drop.h: head_dim 16 (every PT-v3m1 / m2 stage)
                raise PtcoreError("attention dropout in the flash branch is implemented for head_dim 16 (every reference config uses 0.0)")
            self.patch_size = patch_size
            self.attn_drop = attn_drop
        else:
            self.patch_size_max = patch_size       # ptv3m1:92-97
            self.patch_size = 0
            self.attn_drop = nn.Dropout(attn_drop)
        self.qkv = PNN.Linear(channels, channels * 3, bias=qkv_bias)
        self.proj = PNN.Linear(channels, channels)
        self.proj_drop = nn.Dropout(proj_drop)
        self.rpe = RPE(patch_size, num_heads) if enable_rpe else None

    def _kernel_ok(self) -> bool:
        drop = self.attn_drop.p if isinstance(self.attn_drop, nn.Dropout) else self.attn_drop
        hd = self.channels // self.num_heads
        fits = self.patch_size <= (1024 if hd <= 32 else 672 if hd <= 48 else 512)       # ops.attn_hd_supported, host-side
        return self.rpe is None and 16 <= hd <= 64 and fits and (drop == 0.0 or not self.training)

    @torch.no_grad()
    def get_rel_pos(self, point, order):
        """ptv3m1:104-112, cached per order under the reference's key."""
        key = f"rel_pos_{self.order_index}"
        if key not in point.keys():
            g = point.grid_coord[order].reshape(-1, self.patch_size, 3)
            point[key] = g.unsqueeze(2) - g.unsqueeze(1)
        return point[key]

    def _rpe_kernel_ok(self, point) -> bool:
        """RPE branch on the window-attention kernels (attention_rpe.h) instead of the dense [P,H,K,K] formulation: whenever the
        operands are 16-bit anyway -- bf16 or fp16 autocast (the reference's matmuls then run in the autocast dtype whatever the upcast
        flags say; `configs/s3dis/semseg-pt-v3m1-1-rpe.py` runs under the default fp16 AMP): the kernel keeps fp32 logits, softmax and
        accumulation, fp16 tensors go in and out through its load / store paths around bf16 operands (as in the flash branch) --, no
        attention dropout is active, head_dim 16.  An fp32 run (no autocast) keeps the torch formulation below BY NAME: `(q * scale) @ k^T`,
        `torch.softmax`, `attn @ v` on fp32 tensors -- its results must not be rounded to 16 bits (INTEGRATION, stated deviations)."""
        drop = self.attn_drop.p if isinstance(self.attn_drop, nn.Dropout) else self.attn_drop
        return (config.RPE_KERNEL and self.rpe is not None and point.feat.is_cuda and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16) and (drop == 0.0 or not self.training)
                and ops.attn_rpe_supported(self.channels // self.num_heads, self.patch_size, self.rpe.pos_bnd))

    def _forward_rpe_kernel(self, point):
        H, K, C = self.num_heads, self.patch_size, self.channels
        pad, unpad, cu_seqlens = self.get_padding_and_inverse(point)
        order = point.serialized_order[self.order_index][pad]
        inverse = unpad[point.serialized_inverse[self.order_index]]
        key = f"_ptc_rpe_coord_{self.order_index}"
        if key not in point.keys():                                   # the role of get_rel_pos' cache (ptv3m1:104-112): O(N), not O(N K)
            point[key] = point.grid_coord[order].to(torch.int32)
        qkv = self.qkv(point.feat)[order]                              # ptv3m1:188
        q16 = qkv if qkv.dtype in (torch.bfloat16, torch.float16) else qkv.to(torch.bfloat16)
        out = PF.attn_rpe_qkvpacked(q16.reshape(-1, 3, H, C // H), cu_seqlens, K, self.scale, point[key],
                                    self.rpe.rpe_table, self.rpe.pos_bnd)
        feat = out.reshape(-1, C).to(qkv.dtype)[inverse]               # ptv3m1:206,216
        point.feat = self.proj_drop(self.proj(feat))
        return point

    def _forward_dense(self, point):
        if self._rpe_kernel_ok(point):
            return self._forward_rpe_kernel(point)
        H, K, C = self.num_heads, self.patch_size, self.channels
        pad, unpad, _ = self.get_padding_and_inverse(point)
        order = point.serialized_order[self.order_index][pad]
        inverse = unpad[point.serialized_inverse[self.order_index]]
        qkv = self.qkv(point.feat)[order]
        q, k, v = qkv.reshape(-1, K, 3, H, C // H).permute(2, 0, 3, 1, 4).unbind(dim=0)   # [P, H, K, D]
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

    @torch.no_grad()
    def get_padding_and_inverse(self, point):
        """ptv3m1:114-170, one kernel launch; cached on the Point under the reference's keys."""
        if "pad" not in point.keys() or "unpad" not in point.keys() or "cu_seqlens_key" not in point.keys():
            _, offset_host = point._host_facts()
            pad, unpad, cu, dup = ops.patch_pad_maps(point.offset, offset_host, self.patch_size)
            point["pad"], point["unpad"], point["cu_seqlens_key"], point["_ptc_dup"] = pad, unpad, cu, dup
            point["_ptc_pad_patch"] = self.patch_size
        elif point.get("_ptc_pad_patch", self.patch_size) != self.patch_size:
            # The reference caches the maps under fixed keys whatever the patch size (ptv3m1:119-124) and then hands
            # flash-attn cu_seqlens of longer windows than max_seqlen = K: rows past K are never written there.  The
            # engine's kernels size their LDS from max_seqlen, so refuse instead of computing garbage.
            raise ops.PtcoreError(
                f"pad / cu_seqlens on this Point were built for patch size {point['_ptc_pad_patch']} but this attention uses "
                f"{self.patch_size}: enc_patch_size and dec_patch_size of one stage must agree (reference quirk, ptv3m1:119-124)")
        return point["pad"], point["unpad"], point["cu_seqlens_key"]

    @torch.no_grad()
    def _index_maps(self, point):
        """per serialization order: gather index (padded slot -> point), its inverse (point -> slot),
        and the two maps that make both backward passes pure gathers; plus the same four maps as the
        int32 kv = 1 / kv = 2 gather tables that fold the permutation into the qkv / proj GEMMs."""
        key = f"_ptc_attn_maps_{self.order_index}"
        if key not in point.keys():
            pad, unpad, _ = self.get_padding_and_inverse(point)
            order = point.serialized_order[self.order_index]
            inverse = point.serialized_inverse[self.order_index]
            if config.FUSE_GATHER:   # all four int32 tables in one launch; the int64 maps are not needed
                point[key] = (None, None, None, None, ops.attn_tables(order, inverse, pad, unpad, point["_ptc_dup"]))
            else:
                gidx = order[pad]                      # ptv3m1:184
                inv = unpad[inverse]                   # ptv3m1:185
                dup_of_point = point["_ptc_dup"][inverse]   # second slot holding each point, or -1
                slots = torch.arange(gidx.numel(), device=gidx.device)
                gidx_primary = torch.where(inv[gidx] == slots, gidx, torch.full_like(gidx, -1))
                point[key] = (gidx, inv, dup_of_point, gidx_primary, None)
        return point[key]

    def forward(self, point):
        if not self.enable_flash:
            _, offset_host = point._host_facts()
            smallest = min(b - a for a, b in zip([0] + list(offset_host[:-1]), offset_host))
            self.patch_size = min(int(smallest), self.patch_size_max)      # ptv3m1:173-176
            if not self._kernel_ok():
                return self._forward_dense(point)
        H, K, C = self.num_heads, self.patch_size, self.channels
        _, _, cu_seqlens = self.get_padding_and_inverse(point)
        gidx, inv, dup_of_point, gidx_primary, tabs = self._index_maps(point)
        if tabs is not None:
            # qkv[order] == Linear(feat[order]): the GEMM reads its rows through the gather table and
            # writes the padded, serialized qkv directly (ptv3m1:188); proj reads the attention output
            # through the inverse table (ptv3m1:216,219).  Two full gather passes less per block.
            qkv_s = self.qkv(point.feat, tabs[0], tabs[1])
            drop = float(self.attn_drop) if (self.enable_flash and self.training) else 0.0        # ptv3m1:212
            if qkv_s.dtype == torch.float16 and C // H == 16:
                # fp16 autocast: qkv.to(bfloat16) and feat.to(qkv.dtype) (ptv3m1:209,215) happen inside the kernels' loads and stores
                out = PF.attn_varlen_qkvpacked(qkv_s.reshape(-1, 3, H, C // H), cu_seqlens, K, self.scale, drop)
                feat = self.proj(out.reshape(-1, C), tabs[2], tabs[3])
            else:
                out = PF.attn_varlen_qkvpacked(qkv_s.to(torch.bfloat16).reshape(-1, 3, H, C // H), cu_seqlens, K, self.scale, drop)
                feat = self.proj(out.reshape(-1, C).to(qkv_s.dtype), tabs[2], tabs[3])   # ptv3m1:215
        else:
            qkv = self.qkv(point.feat)
            # padded, serialized qkv in bf16 (ptv3m1:188,209); backward = gather through (inv, dup)
            qkv_s = PF.gather_rows(qkv.to(torch.bfloat16), gidx, inv, dup_of_point)
            out = PF.attn_varlen_qkvpacked(qkv_s.reshape(-1, 3, H, C // H), cu_seqlens, K, self.scale,
                                           float(self.attn_drop) if (self.enable_flash and self.training) else 0.0)
            feat = PF.gather_rows(out.reshape(-1, C), inv, gidx_primary)      # ptv3m1:216
            feat = feat.to(qkv.dtype)                                          # ptv3m1:215
            feat = self.proj(feat)
        feat = self.proj_drop(feat)
        point.feat = feat
        return point


class MLP(nn.Module):
    def __init__(self, in_channels, hidden_channels=None, out_channels=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_channels = out_channels or in_channels
        hidden_channels = hidden_channels or in_channels
        self.fc1 = PNN.Linear(in_channels, hidden_channels)
        self.act = act_layer()
        self.fc2 = PNN.Linear(hidden_channels, out_channels)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        if (config.FUSE_MLP and self.drop.p == 0.0 and type(self.act) in (nn.GELU, PNN.GELU) and getattr(self.act, "approximate", "none") == "none"
                and isinstance(self.fc1, PNN.Linear) and PF.mlp_gelu_supported(x, self.fc1.weight, self.fc2.weight)):
            return PF.mlp_gelu(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)   # GELU in the GEMM epilogues
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class Block(PointModule):
    def __init__(self, channels, num_heads, patch_size=48, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, act_layer=nn.GELU, pre_norm=True,
                 order_index=0, cpe_indice_key=None, enable_rpe=False, enable_flash=True, upcast_attention=True,
                 upcast_softmax=True):
        super().__init__()
        self.channels, self.pre_norm = channels, pre_norm
        self.cpe = PointSequential(
            spconv.SubMConv3d(channels, channels, kernel_size=3, bias=True, indice_key=cpe_indice_key),
            PNN.Linear(channels, channels),
            norm_layer(channels),
        )
        self.norm1 = PointSequential(norm_layer(channels))
        self.attn = self._build_attn(
            channels=channels, patch_size=patch_size, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
            attn_drop=attn_drop, proj_drop=proj_drop, order_index=order_index, enable_rpe=enable_rpe,
            enable_flash=enable_flash, upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)
        self.norm2 = PointSequential(norm_layer(channels))
        self.mlp = PointSequential(MLP(in_channels=channels, hidden_channels=int(channels * mlp_ratio),
                                       out_channels=channels, act_layer=act_layer, drop=proj_drop))
        self.drop_path = PointSequential(DropPath(drop_path) if drop_path > 0.0 else nn.Identity())
        if pre_norm:  # norm1 / norm2 feed only a GEMM: under autocast they emit its operand dtype directly
            for m in (self.norm1[0], self.norm2[0]):
                if isinstance(m, PNN.LayerNorm):
                    m.gemm_consumer = True

    def _build_attn(self, **kw):
        """the attention module of this block family (PT-v3m3 builds its RoPE variant here)"""
        return SerializedAttention(**kw)

    _ones = {}   # device -> persistent ones buffer (grown on demand): source operand of the one-launch mask below

    def _row_keep_scale(self, n, device):
        """per-point DropPath factor (timm DropPath on [N,C] drops rows) or None.
        ONE launch: fused dropout of a persistent ones vector yields 0 or 1/keep per row (bernoulli_ + div_ were two
        launches per mask, 80 per step at the bench config, profiles/r02_g_trace_copies.txt)."""
        dp = self.drop_path[0]
        p = getattr(dp, "drop_prob", 0.0)
        if p == 0.0 or not self.training:
            return None
        keep = 1.0 - p
        if not (keep > 0.0 and getattr(dp, "scale_by_keep", True)):
            return torch.empty(n, dtype=torch.float32, device=device).bernoulli_(keep)
        ones = Block._ones.get(device)
        if ones is None or ones.numel() < n:
            ones = Block._ones[device] = torch.ones(max(n, 1 << 20), dtype=torch.float32, device=device)
        return torch.nn.functional.dropout(ones[:n], p=p, training=True)

    def _fusable(self, point) -> bool:
        return (config.FUSE_BLOCK and self.pre_norm and point.feat.is_cuda and point.feat.dim() == 2
                and isinstance(self.cpe[2], PNN.LayerNorm) and isinstance(self.norm1[0], PNN.LayerNorm)
                and isinstance(self.norm2[0], PNN.LayerNorm) and ops.layer_norm_joint_available(self.channels)
                and all(ln.weight is not None and ln.bias is not None for ln in (self.cpe[2], self.norm1[0], self.norm2[0]))
                and point.feat.shape[0] > 0 and point.feat.dtype in (torch.float32, torch.bfloat16, torch.float16))

    def _forward_fused(self, point: Point):
        """Same arithmetic as `forward`, with each residual joint (branch output -> [LayerNorm] -> add ->
        [LayerNorm | cast]) done in one pass (PF.add_norm) instead of 3-4 elementwise kernels."""
        auto = torch.is_autocast_enabled("cuda")
        gemm_dt = torch.get_autocast_dtype("cuda") if auto else torch.float32
        if gemm_dt not in (torch.bfloat16, torch.float16, torch.float32):
            return None
        n, dev = point.feat.shape[0], point.feat.device
        sc = point.sparse_conv_feat
        x0 = point.feat                                           # residual stream (fp32 after the first joint)
        lin = self.cpe[1](self.cpe[0](sc).features)                  # CPE conv (stale input in dec block 0: D.1) + Linear
        if lin.dtype != gemm_dt:
            lin = lin.to(gemm_dt)
        x1, y1 = PF.add_norm(lin, x0, None, self.cpe[2], self.norm1[0], gemm_dt)        # x + LN(cpe); norm1
        point.feat = y1
        point = self.attn(point)
        a = point.feat if point.feat.dtype == gemm_dt else point.feat.to(gemm_dt)
        x2, y2 = PF.add_norm(a, x1, self._row_keep_scale(n, dev), None, self.norm2[0], gemm_dt)  # + droppath(attn); norm2
        h = self.mlp[0](y2)
        if h.dtype != gemm_dt:
            h = h.to(gemm_dt)
        x3, xb = PF.add_norm(h, x2, self._row_keep_scale(n, dev), None, None, gemm_dt if auto else None)  # + droppath(mlp)
        point.feat = x3
        point.sparse_conv_feat = sc.replace_feature(xb if xb is not None else x3)    # next conv's operand, already cast
        if xb is not None:
            PF.register_cast_twin(x3, xb)     # ... and the operand of any Linear that reads the stream (pooling, unpooling, head)
        return point

    def _exec_ok(self, point) -> bool:
        """the whole block as one C call per direction (csrc/block_exec.hip): the bench configuration class -- bf16 or fp16 autocast,
        pre-norm, LayerNorm joints, the window-attention kernel with the gather tables folded into qkv / proj, fused MLP, <= 256 channels"""
        a = self.attn
        return (config.EXEC_BLOCK and type(self) in _EXEC_BLOCK_TYPES and type(a) is SerializedAttention and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16) and self.channels % 32 == 0
                and (self.channels <= 256 or (self.channels <= 512 and ops.linear_supported_ex(self.channels, 4 * self.channels, torch.get_autocast_dtype("cuda"))))
                and a.enable_flash and not a.enable_rpe and a.num_heads * 16 == self.channels and config.FUSE_GATHER and config.FUSE_MLP
                and (a.attn_drop == 0.0 or not self.training)
                and type(self.cpe[0]) is spconv.SubMConv3d and self.cpe[0].kernel_size[0] == 3 and self.cpe[0].bias is not None
                and type(self.cpe[1]) is PNN.Linear and self.cpe[1].bias is not None and type(self.mlp[0]) is MLP
                and type(self.mlp[0].fc1) is PNN.Linear and type(self.mlp[0].fc2) is PNN.Linear and self.mlp[0].fc1.bias is not None
                and self.mlp[0].fc2.bias is not None and isinstance(self.mlp[0].act, nn.GELU) and getattr(self.mlp[0].act, "approximate", "none") == "none"
                and self.mlp[0].fc1.out_features == 4 * self.channels
                and getattr(self.mlp[0].drop, "p", 0.0) == 0.0 and getattr(a.proj_drop, "p", 0.0) == 0.0 and a.proj.bias is not None
                and all(m.elementwise_affine for m in (self.cpe[2], self.norm1[0], self.norm2[0]))
                and point.feat.dtype in (torch.float32, torch.get_autocast_dtype("cuda")))

    def _forward_exec(self, point: Point):
        sc = point.sparse_conv_feat
        conv = self.cpe[0]
        nbr, rep, blocks = conv.tables(sc)
        if rep is not None:                      # duplicate voxels (Mix3D): the adjoint needs the merge passes of the composed path
            return None
        a = self.attn
        _, _, cu = a.get_padding_and_inverse(point)
        tabs = a._index_maps(point)[4]
        if tabs is None:
            return None
        x0 = point.feat
        n, c = x0.shape
        dt = torch.get_autocast_dtype("cuda")
        xc = sc.features if sc.features.dtype == dt else sc.features.to(dt)
        blk = None if blocks is None else blocks.get(c, c, dt)
        meta = dict(dt=dt, n_pad=int(tabs[0].shape[1]), n_seq=int(cu.numel()) - 1, heads=a.num_heads, patch=int(a.patch_size), scale=float(a.scale),
                    eps_cpe=self.cpe[2].eps, eps_n1=self.norm1[0].eps, eps_n2=self.norm2[0].eps, nbr=nbr, blk=blk, tabs=tabs, cu=cu)
        m = self.mlp[0]
        params = (conv.weight, conv.bias, self.cpe[1].weight, self.cpe[1].bias, self.cpe[2].weight, self.cpe[2].bias, self.norm1[0].weight,
                  self.norm1[0].bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, self.norm2[0].weight, self.norm2[0].bias,
                  m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
        x3, xb = PF.ptv3_block(x0, xc, self._row_keep_scale(n, x0.device), self._row_keep_scale(n, x0.device), meta, params)
        point.feat = x3
        point.sparse_conv_feat = sc.replace_feature(xb)
        PF.register_cast_twin(x3, xb)
        return point

    def forward(self, point: Point):
        if self._fusable(point):
            out = self._forward_exec(point) if self._exec_ok(point) else None
            if out is None:
                out = self._forward_fused(point)
            if out is not None:
                return out
        shortcut = point.feat
        point = self.cpe(point)  # consumes point.sparse_conv_feat.features (stale after unpooling: Appendix D.1)
        point.feat = shortcut + point.feat
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


# block classes whose forward is the sequence csrc/block_exec.hip enqueues (PT-v3m2's Block without LayerScale registers itself)
_EXEC_BLOCK_TYPES = {Block}


class SerializedPooling(PointModule):
    def __init__(self, in_channels, out_channels, stride=2, norm_layer=None, act_layer=None, reduce="max",
                 shuffle_orders=True, traceable=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        assert stride == 2 ** (math.ceil(stride) - 1).bit_length()
        self.stride = stride
        assert reduce in ["sum", "mean", "min", "max"]
        self.reduce, self.shuffle_orders, self.traceable = reduce, shuffle_orders, traceable
        self.proj = PNN.Linear(in_channels, out_channels)
        if norm_layer is not None:
            self.norm = PointSequential(norm_layer(out_channels))
        if act_layer is not None:
            self.act = PointSequential(act_layer())

    def forward(self, point: Point):
        pooling_depth = (math.ceil(self.stride) - 1).bit_length()
        if pooling_depth > point.serialized_depth:
            pooling_depth = 0
        assert {"serialized_code", "serialized_order", "serialized_inverse", "serialized_depth"}.issubset(point.keys()), \
            "Run point.serialization() point cloud before SerializedPooling"
        shift = pooling_depth * 3
        coord_max, offset_host = point._host_facts()
        with torch.no_grad():
            code, order0 = point.serialized_code, point.serialized_order[0]
            # ptv3m1:383-396 without torch.unique / torch.sort: row 0 is already sorted
            levels = point.get("_ptc_pool_levels") or []
            known = levels[0] if levels and levels[0] and levels[0][-1] > 0 else None   # child offsets, prefetched
            cluster, idx_ptr, head = ops.pool_maps(code[0], order0, shift, None if known is None else known[-1])
            child_code = ops.pool_child_codes(code, head, shift)                    # ptv3m1:398
            depth = point.serialized_depth - pooling_depth
            if self.shuffle_orders:       # ptv3m1:408-412; the rows are sorted independently: permuting the codes FIRST gives the permuted
                perm = torch.randperm(child_code.shape[0])                          # order / inverse rows without gathering them (CPU RNG, ptv3m1:409)
                child_code = child_code[perm]
            order, inverse = ops.sort_keys(child_code, 0, depth * 3 + len(offset_host).bit_length())  # :399-406
            grid_coord = point.grid_coord[head] >> pooling_depth
            batch = point.batch[head]
        point_dict = AttrDict(
            feat=PF.segment_csr(self.proj(point.feat), idx_ptr, self.reduce, perm=order0, covers_all=True),   # ptv3m1:416-418
            coord=PF.segment_csr(point.coord, idx_ptr, "mean", perm=order0),                   # ptv3m1:419-421
            grid_coord=grid_coord,
            serialized_code=child_code,
            serialized_order=order,
            serialized_inverse=inverse,
            serialized_depth=depth,
            batch=batch,
        )
        if "condition" in point.keys():
            point_dict["condition"] = point.condition
        if "context" in point.keys():
            point_dict["context"] = point.context
        if self.traceable:
            point_dict["pooling_inverse"] = cluster
            point_dict["pooling_parent"] = point
        child = Point(point_dict)
        # engine-side caches: CSR of the clusters (gather-form backward of unpooling) and host facts
        child["_ptc_pool_csr"] = (order0, idx_ptr)
        child["_ptc_coord_max"] = [int(m) >> pooling_depth for m in coord_max]
        child["_ptc_offset_host"] = list(known) if known is not None else child.offset.tolist()
        child["_ptc_pool_levels"] = levels[1:]
        child["_ptc_n_dup"] = 0   # one row per cluster: pooled coordinates are unique
        child = norm_then_act(self, child)
        child.sparsify()
        return child


class SerializedUnpooling(PointModule):
    def __init__(self, in_channels, skip_channels, out_channels, norm_layer=None, act_layer=None, traceable=False):
        super().__init__()
        self.proj = PointSequential(PNN.Linear(in_channels, out_channels))
        self.proj_skip = PointSequential(PNN.Linear(skip_channels, out_channels))
        if norm_layer is not None:
            self.proj.add(norm_layer(out_channels))
            self.proj_skip.add(norm_layer(out_channels))
        if act_layer is not None:
            self.proj.add(act_layer())
            self.proj_skip.add(act_layer())
        self.traceable = traceable

    def forward(self, point):
        assert "pooling_parent" in point.keys() and "pooling_inverse" in point.keys()
        parent = point.pop("pooling_parent")
        inverse = point.pop("pooling_inverse")
        perm, idx_ptr = point["_ptc_pool_csr"]
        point = self.proj(point)
        parent = self.proj_skip(parent)
        # ptv3m1:478 -- note: parent.sparse_conv_feat is NOT refreshed here (Appendix D.1)
        parent.feat = PF.gather_by_cluster_add(parent.feat, point.feat, inverse, perm, idx_ptr)
        if self.traceable:
            parent["unpooling_parent"] = point
        return parent


class Embedding(PointModule):
    def __init__(self, in_channels, embed_channels, norm_layer=None, act_layer=None):
        super().__init__()
        self.in_channels, self.embed_channels = in_channels, embed_channels
        self.stem = PointSequential(conv=spconv.SubMConv3d(in_channels, embed_channels, kernel_size=5, padding=1,
                                                           bias=False, indice_key="stem"))
        if norm_layer is not None:
            self.stem.add(norm_layer(embed_channels), name="norm")
        if act_layer is not None:
            self.stem.add(act_layer(), name="act")

    def forward(self, point: Point):
        return self.stem(point)


class PointTransformerV3(PointModule):
    """registry name "PT-v3m1" (ptv3m1:518); kwargs exactly as ptv3m1:520-552."""

    def __init__(self, in_channels=6, order=("z", "z-trans"), stride=(2, 2, 2, 2), enc_depths=(2, 2, 2, 6, 2),
                 enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32),
                 enc_patch_size=(48, 48, 48, 48, 48), dec_depths=(2, 2, 2, 2), dec_channels=(64, 64, 128, 256),
                 dec_num_head=(4, 4, 8, 16), dec_patch_size=(48, 48, 48, 48), mlp_ratio=4, qkv_bias=True, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, drop_path=0.3, pre_norm=True, shuffle_orders=True, enable_rpe=False,
                 enable_flash=True, upcast_attention=False, upcast_softmax=False, enc_mode=False, pdnorm_bn=False,
                 pdnorm_ln=False, pdnorm_decouple=True, pdnorm_adaptive=False, pdnorm_affine=True,
                 pdnorm_conditions=("ScanNet", "S3DIS", "Structured3D")):
        super().__init__()
        self.num_stages = len(enc_depths)
        self.order = [order] if isinstance(order, str) else order
        self.enc_mode = enc_mode
        self.shuffle_orders = shuffle_orders
        assert self.num_stages == len(stride) + 1 == len(enc_channels) == len(enc_num_head) == len(enc_patch_size)
        assert self.enc_mode or self.num_stages == len(dec_depths) + 1 == len(dec_channels) + 1
        assert self.enc_mode or self.num_stages == len(dec_num_head) + 1 == len(dec_patch_size) + 1

        bn_layer = lambda c: PNN.BatchNorm1d(c, eps=1e-3, momentum=0.01)  # noqa: E731  (ptv3m1:581)
        ln_layer = PNN.LayerNorm
        pd = dict(conditions=pdnorm_conditions, decouple=pdnorm_decouple, adaptive=pdnorm_adaptive)
        if pdnorm_bn:   # ptv3m1:570-580 (PPT multi-dataset training; the per-condition layers are the engine's own)
            bn_layer = partial(PDNorm, norm_layer=partial(PNN.BatchNorm1d, eps=1e-3, momentum=0.01, affine=pdnorm_affine), **pd)
        if pdnorm_ln:   # ptv3m1:582-590
            ln_layer = partial(PDNorm, norm_layer=partial(PNN.LayerNorm, elementwise_affine=pdnorm_affine), **pd)
        act_layer = PNN.GELU
        blk = dict(mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop,
                   norm_layer=ln_layer, act_layer=act_layer, pre_norm=pre_norm, enable_rpe=enable_rpe,
                   enable_flash=enable_flash, upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)

        self.embedding = Embedding(in_channels=in_channels, embed_channels=enc_channels[0], norm_layer=bn_layer,
                                   act_layer=act_layer)
        enc_dp = [x.item() for x in torch.linspace(0, drop_path, sum(enc_depths))]
        self.enc = PointSequential()
        for s in range(self.num_stages):
            dp = enc_dp[sum(enc_depths[:s]):sum(enc_depths[:s + 1])]
            enc = PointSequential()
            if s > 0:
                enc.add(SerializedPooling(in_channels=enc_channels[s - 1], out_channels=enc_channels[s],
                                          stride=stride[s - 1], norm_layer=bn_layer, act_layer=act_layer), name="down")
            for i in range(enc_depths[s]):
                enc.add(Block(channels=enc_channels[s], num_heads=enc_num_head[s], patch_size=enc_patch_size[s],
                              drop_path=dp[i], order_index=i % len(self.order), cpe_indice_key=f"stage{s}", **blk),
                        name=f"block{i}")
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
                dec.add(SerializedUnpooling(in_channels=dec_channels[s + 1], skip_channels=enc_channels[s],
                                            out_channels=dec_channels[s], norm_layer=bn_layer, act_layer=act_layer),
                        name="up")
                for i in range(dec_depths[s]):
                    dec.add(Block(channels=dec_channels[s], num_heads=dec_num_head[s], patch_size=dec_patch_size[s],
                                  drop_path=dp[i], order_index=i % len(self.order), cpe_indice_key=f"stage{s}", **blk),
                            name=f"block{i}")
                self.dec.add(module=dec, name=f"dec{s}")

    def forward(self, data_dict):
        with PNN.batched_bn_counters():          # the BatchNorm sites' `num_batches_tracked += 1` in one launch
            return self._forward(data_dict)

    def _forward(self, data_dict):
        point = Point(data_dict)
        point.serialization(order=self.order, shuffle_orders=self.shuffle_orders)
        self._prefetch_pool_levels(point)
        caller = None
        if config.SORT_POINTS and point.feat.is_cuda:
            caller, point = point, point.physically_sorted()
        point.sparsify()
        point = self.embedding(point)
        point = self.enc(point)
        if not self.enc_mode:
            point = self.dec(point)
        if caller is not None:
            point = self._restore_order(point, caller)
        return point

    def _prefetch_pool_levels(self, point):
        """Point counts of every pooled stage, fetched with ONE host copy before any feature work is queued (the
        reference syncs twice per SerializedPooling: torch.unique and the python loop over bincounts).  With the sizes
        known the host never waits for the GPU again during the forward, so it runs far ahead and the short kernels of
        the deep stages find their launches already queued."""
        strides = self.__dict__.get("_ptc_pool_strides")
        if strides is None:                      # (walking self.modules() every step cost 0.6 ms of host time, r03_i_host_profile.txt)
            strides = self.__dict__["_ptc_pool_strides"] = [m.stride for m in self.modules() if isinstance(m, SerializedPooling)]
        if not config.PREFETCH_LEVELS or not strides or not point.feat.is_cuda or point.grid_coord.shape[0] == 0 or len(strides) > 8:
            return
        depth, shifts, cum = point.serialized_depth, [], 0
        for st in strides:
            pd = (math.ceil(st) - 1).bit_length()
            if pd > depth:
                pd = 0
            cum += 3 * pd
            depth -= pd
            shifts.append(cum)
        _, offset_host = point._host_facts()
        counts = ops.pool_level_counts(point.serialized_code[0], point.serialized_order[0], 3 * point.serialized_depth,
                                       len(offset_host), shifts).tolist()
        levels = []
        for per_scene in counts:
            acc, offs = 0, []
            for c in per_scene:
                acc += int(c)
                offs.append(acc)
            levels.append(offs)
        point["_ptc_pool_levels"] = levels

    @staticmethod
    def _restore_order(point, caller):
        """undo Point.physically_sorted on the stage-0 point (the returned point itself, or -- in
        enc_mode -- the last `pooling_parent` of the returned chain, consumed at default.py:69-74)"""
        if "pooling_parent" not in point.keys():
            return point.restore_order(caller)
        child = point
        while "pooling_parent" in child["pooling_parent"].keys():
            child = child["pooling_parent"]
        stage0 = child["pooling_parent"]
        _, inv0 = stage0["_ptc_unsort"]
        child["pooling_parent"] = stage0.restore_order(caller)
        child["pooling_inverse"] = child["pooling_inverse"][inv0]
        return point
