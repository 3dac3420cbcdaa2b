"""SpUNet-v1m1 on the engine: drop-in for
pointcept/models/sparse_unet/spconv_unet_v1m1_base.py (registry name "SpUNet-v1m1", ctor :90-98,
forward(input_dict{grid_coord, feat, offset}) -> [N, num_classes] :244-280, same state-dict keys:
conv_input.0.weight, down.{s}.0.weight, enc.{s}.block{i}.conv{1,2}.weight, up.{s}.0.weight,
dec.{s}.block{i}.{proj.0,conv1,conv2}.weight, final.{weight,bias}).

All 60 sparse convolutions of a forward run on the gather-table MFMA kernel (spconv.hip); the nine
rulebooks (stem k5, subm0..4 k3, spconv1..4 k2s2 + their transposes) are built on device by
rulebook.hip and cached per indice_key in SparseConvTensor.indice_dict.
"""
from __future__ import annotations

from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from . import config as _config
from . import nn as PNN
from . import spconv_api as spconv
from .structure import offset2batch


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):  # timm.layers.trunc_normal_
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class BasicBlock(spconv.SparseModule):
    """residual block of two SubM k=3 convs (spconv_unet_v1m1_base.py:23-85)"""
    expansion = 1

    def __init__(self, in_channels, embed_channels, stride=1, norm_fn=None, indice_key=None, bias=False):
        super().__init__()
        assert norm_fn is not None
        if in_channels == embed_channels:
            self.proj = spconv.SparseSequential(nn.Identity())
        else:
            self.proj = spconv.SparseSequential(
                spconv.SubMConv3d(in_channels, embed_channels, kernel_size=1, bias=False), norm_fn(embed_channels))
        self.conv1 = spconv.SubMConv3d(in_channels, embed_channels, kernel_size=3, stride=stride, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(embed_channels)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(embed_channels, embed_channels, kernel_size=3, stride=stride, padding=1,
                                       bias=bias, indice_key=indice_key)
        self.bn2 = norm_fn(embed_channels)
        self.stride = stride

    def forward(self, x):
        residual = x
        out = self.conv1(x)
        # both fusions are decided from the modules that are in the tree NOW (PNN.fused_act): after convert_sync_batchnorm or
        # any other module rewrite the block runs the reference's three-pass form below
        kind = PNN.fused_act(self.bn1, self.relu)               # bn1 is always followed by self.relu (spconv_unet_v1m1_base.py:77)
        out = out.replace_feature(self.bn1(out.features, act=kind) if kind else self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        if _config.FUSE_BN_TAIL and PNN.fused_act(self.bn2, self.relu) == "relu":   # relu(bn2(.) + residual) in the BatchNorm's apply pass (:79-83)
            return out.replace_feature(self.bn2(out.features, residual=self.proj(residual).features, act="relu"))
        out = out.replace_feature(self.bn2(out.features))
        out = out.replace_feature(self.relu(out.features + self.proj(residual).features))
        return out


class SpUNetBase(nn.Module):
    def __init__(self, in_channels, num_classes, base_channels=32, channels=(32, 64, 128, 256, 256, 128, 96, 96),
                 layers=(2, 3, 4, 6, 2, 2, 2, 2), enc_mode=False, skip=True):
        super().__init__()
        assert len(layers) % 2 == 0 and len(layers) == len(channels)
        self.skip = skip      # False: the decoder does not read the encoder's features (SpUNetNoSkipBase below)
        self.in_channels, self.num_classes, self.base_channels = in_channels, num_classes, base_channels
        self.channels, self.layers = channels, layers
        self.num_stages = len(layers) // 2
        self.enc_mode = enc_mode
        norm_fn = partial(PNN.BatchNorm1d, eps=1e-3, momentum=0.01)

        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(in_channels, base_channels, kernel_size=5, padding=1, bias=False, indice_key="stem"),
            norm_fn(base_channels), PNN.ReLU())
        enc_channels, dec_channels = base_channels, channels[-1]
        self.down, self.up, self.enc = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.dec = nn.ModuleList() if not self.enc_mode else None
        for s in range(self.num_stages):
            self.down.append(spconv.SparseSequential(
                spconv.SparseConv3d(enc_channels, channels[s], kernel_size=2, stride=2, bias=False,
                                    indice_key=f"spconv{s + 1}"),
                norm_fn(channels[s]), PNN.ReLU()))
            self.enc.append(spconv.SparseSequential(OrderedDict(
                (f"block{i}", BasicBlock(channels[s], channels[s], norm_fn=norm_fn, indice_key=f"subm{s + 1}"))
                for i in range(layers[s]))))
            if not self.enc_mode:
                self.up.append(spconv.SparseSequential(
                    spconv.SparseInverseConv3d(channels[len(channels) - s - 2], dec_channels, kernel_size=2, bias=False,
                                               indice_key=f"spconv{s + 1}"),
                    norm_fn(dec_channels), PNN.ReLU()))
                self.dec.append(spconv.SparseSequential(OrderedDict(
                    (f"block{i}", BasicBlock(dec_channels + enc_channels if i == 0 and skip else dec_channels, dec_channels,
                                             norm_fn=norm_fn, indice_key=f"subm{s}"))
                    for i in range(layers[len(channels) - s - 1]))))
            enc_channels = channels[s]
            dec_channels = channels[len(channels) - s - 2]
        final_in = channels[-1] if not self.enc_mode else channels[self.num_stages - 1]
        self.final = (spconv.SubMConv3d(final_in, num_classes, kernel_size=1, padding=1, bias=True)
                      if num_classes > 0 else spconv.Identity())
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Linear, spconv.SubMConv3d)):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm1d):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, input_dict):
        with PNN.batched_bn_counters():          # the 59 `num_batches_tracked += 1` of a training step in one launch
            return self._forward(input_dict)

    def _forward(self, input_dict):
        grid_coord, feat, offset = input_dict["grid_coord"], input_dict["feat"], input_dict["offset"]
        batch = offset2batch(offset, int(feat.shape[0]))
        from . import config, ops
        from . import functional as PF

        # Rows arrive in dataloader order (no spatial order at all): every gathered neighbour row is then a random
        # 64..256-byte access.  Sort the rows of the batch along a Hilbert curve once (batch-prefixed key, so scenes
        # stay contiguous and `offset` is unchanged), run the whole network on the sorted rows -- neighbours of
        # consecutive output rows are consecutive-ish input rows: L1/L2 serve the 27-point gathers -- and hand the
        # logits back in the caller's order.  Convolutions do not care about row numbering; BatchNorm sums do not either.
        unsort = None
        if config.SORT_POINTS and feat.is_cuda and grid_coord.shape[0] > 0 and not self.enc_mode:
            gc_l = grid_coord if grid_coord.dtype in (torch.int32, torch.int64) else grid_coord.long()
            depth = 16          # no host fact needed: a deeper curve than the data's is still a curve (keys stay < 2^(3 d_data))
            code = ops.serialize_encode(gc_l, batch, depth, ("hilbert",))
            order, inverse = ops.sort_keys(code, 0, 3 * depth + max(1, int(offset.numel() - 1).bit_length()))
            order, unsort = order[0], inverse[0]
            grid_coord, batch = gc_l[order], batch[order]
            feat = PF.gather_rows(feat, order, unsort)

        indices = torch.cat([batch.unsqueeze(-1).int(), grid_coord.int()], dim=1).contiguous()
        n = indices.shape[0]
        table = ops.HashTable(indices)                       # reused by the stem / subm0 rulebooks below
        rep = ops.rulebook_subm(indices, 1, table)[0]        # lowest row holding each row's voxel
        n_dup = (rep != torch.arange(n, device=rep.device, dtype=rep.dtype)).sum().reshape(1).to(torch.int64)
        host = torch.cat([ops.coord_max(grid_coord), n_dup]).tolist()
        ops.check_coord_range(host[:3], offset.numel())
        sparse_shape = [int(m) + (96 if self.skip else 1) for m in host[:3]]  # spconv_unet_v1m1_base.py:250 / :437 (one host sync)
        x = spconv.SparseConvTensor(features=feat, indices=indices, spatial_shape=sparse_shape, batch_size=int(offset.numel()))
        x.indice_dict["__hash__"] = table
        spconv.mark_duplicates(x, host[3] > 0)   # Mix3D batches: the conv backward needs to know (functional._SparseConv)
        spconv.prefetch_down_rulebooks(x, [f"spconv{s + 1}" for s in range(self.num_stages)])   # all host waits up front
        x = self.conv_input(x)
        skips = [x]
        for s in range(self.num_stages):
            x = self.down[s](x)
            x = self.enc[s](x)
            skips.append(x)
        x = skips.pop(-1)
        if not self.enc_mode:
            for s in reversed(range(self.num_stages)):
                x = self.up[s](x)
                if self.skip:
                    x = x.replace_feature(torch.cat((x.features, skips.pop(-1).features), dim=1))
                x = self.dec[s](x)
        x = self.final(x)
        if self.enc_mode:  # per-scene mean (torch_geometric.utils.scatter(reduce="mean"), :276-279)
            idx = x.indices[:, 0].long()
            f = x.features
            out = f.new_zeros((x.batch_size, f.shape[1])).index_add_(0, idx, f)
            cnt = torch.bincount(idx, minlength=x.batch_size).clamp(min=1).to(f.dtype)
            x = x.replace_feature(out / cnt[:, None])
        if unsort is not None:
            return PF.gather_rows(x.features, unsort, order)     # caller's row order; backward = gather through `order`
        return x.features


class SpUNetNoSkipBase(SpUNetBase):
    """the same U-Net without the encoder -> decoder concatenations (spconv_unet_v1m1_base.py:283-463, registry name
    `SpUNetNoSkipBase`): decoder blocks take the up-sampled features alone, so their first block has `dec_channels` inputs;
    state-dict keys and shapes are the reference class's."""

    def __init__(self, in_channels, out_channels, base_channels=32, channels=(32, 64, 128, 256, 256, 128, 96, 96),
                 layers=(2, 3, 4, 6, 2, 2, 2, 2)):
        super().__init__(in_channels, out_channels, base_channels=base_channels, channels=channels, layers=layers, skip=False)
        self.out_channels = out_channels
