"""Task wrapper used by bench.py / tests when the engine runs OUTSIDE a Pointcept checkout:
the DefaultSegmentorV2 contract of pointcept/models/default.py:40-95 (seg_head Linear on
point.feat, loss in train mode, loss + seg_logits with labels in eval mode, seg_logits otherwise).
Inside Pointcept the reference's own DefaultSegmentorV2 wraps the engine backbone unchanged.

criteria: CrossEntropyLoss(ignore_index=-1) (pointcept/models/losses/misc.py) -- the Lovasz term of
the ScanNet config (scannet/semseg-pt-v3m1-0-base.py:49-52) is a SURVEY 8(f) next-row, not built yet.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as PF
from . import nn as PNN
from .structure import Point


class DefaultSegmentorV2(nn.Module):
    def __init__(self, num_classes, backbone_out_channels, backbone, ignore_index=-1):
        super().__init__()
        self.seg_head = PNN.Linear(backbone_out_channels, num_classes) if num_classes > 0 else nn.Identity()
        self.backbone = backbone
        self.ignore_index = ignore_index

    def criteria(self, seg_logits, segment):
        return PF.cross_entropy(seg_logits, segment, self.ignore_index)   # GPU only, like every op of the engine

    def forward(self, input_dict, return_point=False):
        point = Point(input_dict)
        point = self.backbone(point)
        if isinstance(point, Point):
            while "pooling_parent" in point.keys():  # enc_mode backbones: default.py:69-74
                parent = point.pop("pooling_parent")
                inverse = point.pop("pooling_inverse")
                parent.feat = torch.cat([parent.feat, point.feat[inverse]], dim=-1)
                point = parent
            feat = point.feat
        else:
            feat = point
        seg_logits = self.seg_head(feat)
        return_dict = dict()
        if return_point:
            return_dict["point"] = point
        if self.training:
            return_dict["loss"] = self.criteria(seg_logits, input_dict["segment"])
        elif "segment" in input_dict.keys():
            return_dict["loss"] = self.criteria(seg_logits, input_dict["segment"])
            return_dict["seg_logits"] = seg_logits
        else:
            return_dict["seg_logits"] = seg_logits
        return return_dict
