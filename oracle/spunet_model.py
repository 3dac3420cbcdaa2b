"""TEST INFRASTRUCTURE (oracle).  Standalone CPU restatement of SpUNet-v1m1 that travels to the GPU box
(where /root/reference does not exist).  Pure torch fp32 on oracle/ops.py through the spconv stand-ins
of oracle/shims.py.

Follows pointcept/models/sparse_unet/spconv_unet_v1m1_base.py ("spunet"): residual block :23-85,
U-Net wiring :90-228, forward :244-280.  Module / parameter names equal the reference's, so one
state_dict loads into the reference model, this oracle and the engine model alike.
Pinned against the reference's OWN file run on oracle/shims.py: tests/golden/spunet_tiny.npz
(tests/golden/make_golden.py) and tests/test_golden_cpu.py.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import shims as sp


def _bn(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)          # spunet:108


class BasicBlock(sp.SparseModule):
    """y = relu(bn2(conv2(relu(bn1(conv1(x))))) + proj(x)); proj = identity or (1x1x1 conv, bn)   (spunet:39-85)"""

    def __init__(self, c_in, c, indice_key):
        super().__init__()
        self.proj = sp.SparseSequential(nn.Identity()) if c_in == c else \
            sp.SparseSequential(sp.SubMConv3d(c_in, c, kernel_size=1, bias=False), _bn(c))
        self.conv1 = sp.SubMConv3d(c_in, c, kernel_size=3, padding=1, bias=False, indice_key=indice_key)
        self.bn1 = _bn(c)
        self.conv2 = sp.SubMConv3d(c, c, kernel_size=3, padding=1, bias=False, indice_key=indice_key)
        self.bn2 = _bn(c)

    def forward(self, x):
        h = self.conv1(x)
        h = h.replace_feature(torch.relu(self.bn1(h.features)))
        h = self.conv2(h)
        h = h.replace_feature(self.bn2(h.features))
        return h.replace_feature(torch.relu(h.features + self.proj(x).features))


def _stage(n_blocks, c_first, c, key):
    return sp.SparseSequential(OrderedDict(
        (f"block{i}", BasicBlock(c_first if i == 0 else c, c, key)) for i in range(n_blocks)))


class SpUNetBase(nn.Module):
    def __init__(self, in_channels, num_classes, base_channels=32, channels=(32, 64, 128, 256, 256, 128, 96, 96),
                 layers=(2, 3, 4, 6, 2, 2, 2, 2), enc_mode=False, skip=True):
        super().__init__()
        self.skip = skip
        S = len(layers) // 2
        assert len(layers) == 2 * S == len(channels)
        self.num_stages, self.enc_mode = S, enc_mode
        self.conv_input = sp.SparseSequential(
            sp.SubMConv3d(in_channels, base_channels, kernel_size=5, padding=1, bias=False, indice_key="stem"),
            _bn(base_channels), nn.ReLU())
        self.down, self.up, self.enc = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.dec = None if enc_mode else nn.ModuleList()
        c_enc, c_dec = base_channels, channels[-1]
        for s in range(S):
            key = f"spconv{s + 1}"
            self.down.append(sp.SparseSequential(
                sp.SparseConv3d(c_enc, channels[s], kernel_size=2, stride=2, bias=False, indice_key=key),
                _bn(channels[s]), nn.ReLU()))
            self.enc.append(_stage(layers[s], channels[s], channels[s], f"subm{s + 1}"))
            if not enc_mode:
                # up[s] / dec[s] act at the resolution of encoder stage s-1 (the stem for s = 0), spunet:169-214
                self.up.append(sp.SparseSequential(
                    sp.SparseInverseConv3d(channels[2 * S - s - 2], c_dec, kernel_size=2, bias=False, indice_key=key),
                    _bn(c_dec), nn.ReLU()))
                self.dec.append(_stage(layers[2 * S - s - 1], c_dec + c_enc if skip else c_dec, c_dec, f"subm{s}"))
            c_enc, c_dec = channels[s], channels[2 * S - s - 2]
        c_final = channels[S - 1] if enc_mode else channels[-1]
        self.final = sp.SubMConv3d(c_final, num_classes, kernel_size=1, padding=1, bias=True) \
            if num_classes > 0 else sp.Identity()

    def forward(self, input_dict):
        grid_coord, feat, offset = input_dict["grid_coord"], input_dict["feat"], input_dict["offset"]
        counts = torch.diff(offset, prepend=offset.new_zeros(1))
        batch = torch.repeat_interleave(torch.arange(len(offset), device=offset.device), counts)   # misc.py:12-17
        x = sp.SparseConvTensor(
            features=feat, indices=torch.cat([batch[:, None].int(), grid_coord.int()], dim=1).contiguous(),
            spatial_shape=(grid_coord.max(dim=0).values + (96 if self.skip else 1)).tolist(), batch_size=int(batch[-1]) + 1)   # spunet:249-257 / :437
        x = self.conv_input(x)
        skips = [x]
        for s in range(self.num_stages):
            x = self.enc[s](self.down[s](x))
            skips.append(x)
        x = skips.pop()
        if not self.enc_mode:
            for s in range(self.num_stages - 1, -1, -1):
                x = self.up[s](x)
                if self.skip:
                    x = x.replace_feature(torch.cat([x.features, skips.pop().features], dim=1))            # spunet:272
                x = self.dec[s](x)
        x = self.final(x)
        if self.enc_mode:
            x = x.replace_feature(sp.scatter(x.features, x.indices[:, 0].long(), reduce="mean", dim=0))   # spunet:276-279
        return x.features


class SpUNetNoSkipBase(SpUNetBase):
    """spconv_unet_v1m1_base.py:283-463: no encoder -> decoder concatenation (:456-457 are commented out in the reference)"""

    def __init__(self, in_channels, out_channels, base_channels=32, channels=(32, 64, 128, 256, 256, 128, 96, 96),
                 layers=(2, 3, 4, 6, 2, 2, 2, 2)):
        super().__init__(in_channels, out_channels, base_channels, channels, layers, enc_mode=False, skip=False)


class Segmentor(nn.Module):
    """pointcept/models/default.py:13-37 (DefaultSegmentor: backbone emits the logits) with
    criteria = CrossEntropyLoss(ignore_index=-1)."""

    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone

    def forward(self, input_dict):
        seg_logits = self.backbone(input_dict)
        out = dict(seg_logits=seg_logits)
        if "segment" in input_dict:
            out["loss"] = nn.functional.cross_entropy(seg_logits.float(), input_dict["segment"], ignore_index=-1)
        return out
