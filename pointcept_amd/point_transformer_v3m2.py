"""PT-v3m2 (Sonata / Concerto backbone) on the engine: module-level drop-in for
pointcept/models/point_transformer_v3/point_transformer_v3m2_sonata.py (registry name "PT-v3m2", ctor kwargs
:545-574, forward :724-732, same state-dict keys / shapes).  SURVEY 8(f) rank 2.

What differs from PT-v3m1 and how it maps to the engine:
  Embedding  (:497-541)  Linear stem + LayerNorm + GELU (optional mask token)      -> the engine's GEMM / norm kernels
  Block      (:266-362)  m1's block plus LayerScale on both branches (:29-39)       -> m1's Block; with layer_scale the
                                                                                       three residual joints run unfused
  GridPooling (:365-463) clusters = unique(grid_coord // stride | batch << 48), sorted (:392-399), CSR over the points
                         sorted by cluster (:401-405), segment_csr of the projected features (:407-413), then the
                         child is RE-SERIALIZED from its new grid coordinates (:461)
                         -> one packed lexicographic key per point, the engine's radix sort, the scan-based cluster maps
                            (ptc_pool_maps_* with shift 0: numbering = ascending key = torch.unique(sorted=True)), fused
                            gather + segmented reduce, gather-form backward; no torch.unique / torch.sort / torch_scatter
  GridUnpooling (:466-494) parent.feat = proj_skip(parent) + proj(point)[pooling_inverse]; unlike m1 it REFRESHES
                         parent.sparse_conv_feat (:489) -> PF.gather_by_cluster (backward = segmented sum)
Tie order inside a cluster = ascending point index (stable radix sort; torch.sort(cluster) is unspecified there,
SURVEY Appendix A.3): only `indices` / the order of equal rows inside segment_csr can differ, never a result.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as PF
from . import nn as PNN
from . import ops
from . import point_transformer_v3 as _m1_module
from .point_transformer_v3 import Block as _BlockM1
from .point_transformer_v3 import DropPath, PointModule, PointSequential, norm_then_act
from .structure import AttrDict, Point


class LayerScale(nn.Module):
    """point_transformer_v3m2_sonata.py:29-39"""

    def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False) -> None:
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x.mul_(self.gamma) if self.inplace else x * self.gamma


class Block(_BlockM1):
    """m1's block + LayerScale after the attention and MLP branches (:266-362)."""

    def __init__(self, channels, num_heads, patch_size=48, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0, drop_path=0.0, layer_scale=None, norm_layer=nn.LayerNorm, act_layer=nn.GELU, pre_norm=True,
                 order_index=0, cpe_indice_key=None, enable_rpe=False, enable_flash=True, upcast_attention=True,
                 upcast_softmax=True):
        super().__init__(channels, num_heads, patch_size=patch_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                         attn_drop=attn_drop, proj_drop=proj_drop, drop_path=drop_path, norm_layer=norm_layer, act_layer=act_layer,
                         pre_norm=pre_norm, order_index=order_index, cpe_indice_key=cpe_indice_key, enable_rpe=enable_rpe,
                         enable_flash=enable_flash, upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)
        self.ls1 = PointSequential(LayerScale(channels, init_values=layer_scale) if layer_scale is not None else nn.Identity())
        self.ls2 = PointSequential(LayerScale(channels, init_values=layer_scale) if layer_scale is not None else nn.Identity())
        self.has_layer_scale = layer_scale is not None
        # registration order of the reference (:299-341), so that state_dict() lists the keys in the same order
        mods = dict(self._modules)
        self._modules.clear()
        for name in ("cpe", "norm1", "ls1", "attn", "norm2", "ls2", "mlp", "drop_path"):
            self._modules[name] = mods.pop(name)
        self._modules.update(mods)

    def _fusable(self, point) -> bool:
        return not self.has_layer_scale and super()._fusable(point)   # the fused joints carry a per-row scale only

    def forward(self, point: Point):
        if self._fusable(point):                  # (no LayerScale: the block IS m1's -- one C call per direction where the executor's shapes allow)
            out = self._forward_exec(point) if self._exec_ok(point) else None
            if out is None:
                out = self._forward_fused(point)
            if out is not None:
                return out
        shortcut = point.feat
        point = self.cpe(point)
        point.feat = shortcut + point.feat
        shortcut = point.feat
        if self.pre_norm:
            point = self.norm1(point)
        point = self.drop_path(self.ls1(self.attn(point)))
        point.feat = shortcut + point.feat
        if not self.pre_norm:
            point = self.norm1(point)
        shortcut = point.feat
        if self.pre_norm:
            point = self.norm2(point)
        point = self.drop_path(self.ls2(self.mlp(point)))
        point.feat = shortcut + point.feat
        if not self.pre_norm:
            point = self.norm2(point)
        point.sparse_conv_feat = point.sparse_conv_feat.replace_feature(point.feat)
        return point


_m1_module._EXEC_BLOCK_TYPES.add(Block)


def grid_cluster_maps(grid_coord: torch.Tensor, batch: torch.Tensor, stride: int, coord_max, n_batch: int):
    """(:378-405) coarse cell of every point and the CSR of the cells, on device:
    cell = grid_coord // stride; clusters numbered by ascending (batch, x, y, z) = torch.unique(..., sorted=True, dim=0).
    Returns g [N,3] (cell coordinates), cluster [N] (pooling_inverse), order0 [N] (points sorted by cluster, stable),
    idx_ptr [M+1], head [M] (first point of every cluster)."""
    g = torch.div(grid_coord, stride, rounding_mode="trunc")
    cb = max(1, int(max(int(m) // stride for m in coord_max) + 1).bit_length())
    bb = max(1, int(n_batch - 1).bit_length())
    if 3 * cb + bb > 62:
        raise ops.PtcoreError(f"GridPooling: {3 * cb + bb} key bits needed (coordinates up to {max(coord_max)})")
    gl = g.to(torch.int64)
    key = (((batch.to(torch.int64) << cb) | gl[:, 0]) << cb | gl[:, 1]) << cb | gl[:, 2]
    order0, _ = ops.sort_keys(key[None].contiguous(), 0, 3 * cb + bb, want_inverse=False)
    cluster, idx_ptr, head = ops.pool_maps(key, order0[0], 0, None)
    return g, cluster, order0[0], idx_ptr, head


class GridPooling(PointModule):
    def __init__(self, in_channels, out_channels, stride=2, norm_layer=None, act_layer=None, reduce="max",
                 shuffle_orders=True, traceable=True):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        assert reduce in ["sum", "mean", "min", "max"]
        self.reduce, self.shuffle_orders, self.traceable = reduce, shuffle_orders, traceable
        self.proj = PNN.Linear(in_channels, out_channels)
        if norm_layer is not None:
            self.norm = PointSequential(norm_layer(out_channels))
        if act_layer is not None:
            self.act = PointSequential(act_layer())

    def forward(self, point: Point):
        if "grid_coord" in point.keys():
            grid_coord = point.grid_coord
        elif {"coord", "grid_size"}.issubset(point.keys()):
            grid_coord = torch.div(point.coord - point.coord.min(0)[0], point.grid_size, rounding_mode="trunc").int()
            point["grid_coord"] = grid_coord
        else:
            raise AssertionError("[gird_coord] or [coord, grid_size] should be include in the Point")
        coord_max, offset_host = point._host_facts()
        with torch.no_grad():
            g, cluster, order0, idx_ptr, head = grid_cluster_maps(grid_coord, point.batch, self.stride, coord_max, len(offset_host))
            child_grid = g[head]
            child_batch = point.batch[head]
        point_dict = AttrDict(
            feat=PF.segment_csr(self.proj(point.feat), idx_ptr, self.reduce, perm=order0, covers_all=True),     # :407-409
            coord=PF.segment_csr(point.coord, idx_ptr, "mean", perm=order0),                    # :410-412
            grid_coord=child_grid,
            batch=child_batch,
        )
        for key in ("origin_coord", "color"):                                                   # :416-419, :428-431
            if key in point.keys():
                point_dict[key] = PF.segment_csr(point[key], idx_ptr, "mean", perm=order0)
        for key in ("condition", "context", "name", "split"):                                   # :420-427
            if key in point.keys():
                point_dict[key] = point[key]
        if "grid_size" in point.keys():
            point_dict["grid_size"] = point.grid_size * self.stride
        self._extra_keys(point, point_dict, order0, idx_ptr)
        if self.traceable:
            point_dict["pooling_inverse"] = cluster
            point_dict["pooling_parent"] = point
            if self.trace_idx_ptr:
                point_dict["idx_ptr"] = idx_ptr
        child = Point(point_dict)
        child["_ptc_pool_csr"] = (order0, idx_ptr)      # gather-form backward of the unpooling gather
        child["_ptc_n_dup"] = 0                         # one row per cell
        child = norm_then_act(self, child)
        self._serialize_child(point, child)
        child.sparsify()
        return child

    trace_idx_ptr = True          # m2 / m3 record the CSR on the child (:436); LitePT does not

    def _extra_keys(self, point, point_dict, order0, idx_ptr):
        """keys a derived family pools in addition (LitePT: `mask`)"""

    def _serialize_child(self, point, child):
        child.serialization(order=point.order, shuffle_orders=self.shuffle_orders)              # :461 (re-serialize)


class GridUnpooling(PointModule):
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
        inverse = point.pooling_inverse
        feat = point.feat
        perm, idx_ptr = point["_ptc_pool_csr"]
        parent = self.proj_skip(parent)
        parent.feat = parent.feat + PF.gather_by_cluster(self.proj(point).feat, inverse, perm, idx_ptr)
        parent.sparse_conv_feat = parent.sparse_conv_feat.replace_feature(parent.feat)         # :489 (m1 does not)
        if self.traceable:
            point.feat = feat
            parent["unpooling_parent"] = point
        return parent


class Embedding(PointModule):
    def __init__(self, in_channels, embed_channels, norm_layer=None, act_layer=None, mask_token=False):
        super().__init__()
        self.in_channels, self.embed_channels = in_channels, embed_channels
        self.stem = PointSequential(linear=PNN.Linear(in_channels, embed_channels))
        if norm_layer is not None:
            self.stem.add(norm_layer(embed_channels), name="norm")
        if act_layer is not None:
            self.stem.add(act_layer(), name="act")
        self.mask_token = nn.Parameter(torch.zeros(1, embed_channels)) if mask_token else None

    def forward(self, point: Point):
        point = self.stem(point)
        if "mask" in point.keys():
            point.feat = torch.where(point.mask.unsqueeze(-1), self.mask_token.to(point.feat.dtype), point.feat)
        return point


class PointTransformerV3(PointModule):
    """registry name "PT-v3m2" (:544)"""
    block_cls = None        # = Block (set below the class); derived families substitute theirs

    def __init__(self, in_channels=6, order=("z", "z-trans"), stride=(2, 2, 2, 2), enc_depths=(3, 3, 3, 12, 3),
                 enc_channels=(48, 96, 192, 384, 512), enc_num_head=(3, 6, 12, 24, 32), enc_patch_size=(1024, 1024, 1024, 1024, 1024),
                 dec_depths=(3, 3, 3, 3), dec_channels=(96, 96, 192, 384), dec_num_head=(6, 6, 12, 32),
                 dec_patch_size=(1024, 1024, 1024, 1024), mlp_ratio=4, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0,
                 drop_path=0.3, layer_scale=None, pre_norm=True, shuffle_orders=True, enable_rpe=False, enable_flash=True,
                 upcast_attention=False, upcast_softmax=False, traceable=False, mask_token=False, enc_mode=False,
                 freeze_encoder=False):
        super().__init__()
        self.num_stages = len(enc_depths)
        self.order = [order] if isinstance(order, str) else list(order)
        self.shuffle_orders, self.enc_mode, self.freeze_encoder = shuffle_orders, enc_mode, freeze_encoder
        assert self.num_stages == len(stride) + 1 == len(enc_channels) == len(enc_num_head) == len(enc_patch_size)
        assert self.enc_mode or self.num_stages == len(dec_depths) + 1 == len(dec_channels) + 1
        assert self.enc_mode or self.num_stages == len(dec_num_head) + 1 == len(dec_patch_size) + 1
        ln_layer, act_layer = PNN.LayerNorm, PNN.GELU
        self.embedding = Embedding(in_channels, enc_channels[0], norm_layer=ln_layer, act_layer=act_layer, mask_token=mask_token)
        Block = self.block_cls
        block_kw = dict(mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop,
                        layer_scale=layer_scale, norm_layer=ln_layer, act_layer=act_layer, pre_norm=pre_norm, enable_rpe=enable_rpe,
                        enable_flash=enable_flash, upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)
        enc_drop_path = [x.item() for x in torch.linspace(0, drop_path, sum(enc_depths))]
        self.enc = PointSequential()
        for s in range(self.num_stages):
            dp = enc_drop_path[sum(enc_depths[:s]):sum(enc_depths[:s + 1])]
            enc = PointSequential()
            if s > 0:
                enc.add(GridPooling(enc_channels[s - 1], enc_channels[s], stride=stride[s - 1], norm_layer=ln_layer, act_layer=act_layer),
                        name="down")
            for i in range(enc_depths[s]):
                enc.add(Block(channels=enc_channels[s], num_heads=enc_num_head[s], patch_size=enc_patch_size[s], drop_path=dp[i],
                              order_index=i % len(self.order), cpe_indice_key=f"stage{s}", **block_kw, **self._extra_block_kw(False)),
                        name=f"block{i}")
            if len(enc) != 0:
                self.enc.add(module=enc, name=f"enc{s}")
        if not self.enc_mode:
            dec_drop_path = [x.item() for x in torch.linspace(0, drop_path, sum(dec_depths))]
            self.dec = PointSequential()
            dec_channels = list(dec_channels) + [enc_channels[-1]]
            for s in reversed(range(self.num_stages - 1)):
                dp = dec_drop_path[sum(dec_depths[:s]):sum(dec_depths[:s + 1])]
                dp.reverse()
                dec = PointSequential()
                dec.add(GridUnpooling(dec_channels[s + 1], enc_channels[s], dec_channels[s], norm_layer=ln_layer, act_layer=act_layer,
                                      traceable=traceable), name="up")
                for i in range(dec_depths[s]):
                    dec.add(Block(channels=dec_channels[s], num_heads=dec_num_head[s], patch_size=dec_patch_size[s], drop_path=dp[i],
                                  order_index=i % len(self.order), cpe_indice_key=f"stage{s}", **block_kw, **self._extra_block_kw(True)),
                            name=f"block{i}")
                self.dec.add(module=dec, name=f"dec{s}")
        if self.freeze_encoder:
            for p in list(self.embedding.parameters()) + list(self.enc.parameters()):
                p.requires_grad = False
        self.apply(self._init_weights)

    def _extra_block_kw(self, decoder: bool) -> dict:
        """block kwargs a derived family adds (PT-v3m3: the RoPE settings)"""
        return {}

    @staticmethod
    def _init_weights(module):
        from . import spconv_api as spconv

        if isinstance(module, (nn.Linear, spconv.SubMConv3d)):
            nn.init.trunc_normal_(module.weight, std=0.02, a=-2.0, b=2.0)
            if module.bias is not None:
                nn.init.zeros_(module.bias)

    def forward(self, data_dict):
        point = Point(data_dict)
        point = self.embedding(point)
        point.serialization(order=self.order, shuffle_orders=self.shuffle_orders)
        point.sparsify()
        point = self.enc(point)
        if not self.enc_mode:
            point = self.dec(point)
        return point


PointTransformerV3.block_cls = Block
