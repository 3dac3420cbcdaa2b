"""Task wrapper used by bench.py / tests when the engine runs OUTSIDE a Pointcept checkout:
the DefaultSegmentorV2 contract of pointcept/models/default.py:40-95 (seg_head Linear on
point.feat, loss in train mode, loss + seg_logits with labels in eval mode, seg_logits otherwise).
Inside Pointcept the reference's own DefaultSegmentorV2 wraps the engine backbone unchanged.

criteria (pointcept/models/losses/builder.py:22-31 sums the configured terms): "ce" = CrossEntropyLoss(
ignore_index=-1) (losses/misc.py), "lovasz" = LovaszLoss(mode="multiclass", ignore_index=-1) (losses/lovasz.py);
the ScanNet config uses both with weight 1 (scannet/semseg-pt-v3m1-0-base.py:49-52).  Default ("ce",) -- the
criterion bench.py states.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as PF
from . import nn as PNN
from .structure import Point


class DefaultSegmentorV2(nn.Module):
    def __init__(self, num_classes, backbone_out_channels, backbone, ignore_index=-1, criteria=("ce",), loss_weights=None):
        super().__init__()
        for name in criteria:
            if name not in ("ce", "lovasz"):
                raise ValueError(f"unknown criterion {name!r}")
        self.criteria_names = tuple(criteria)
        self.loss_weights = tuple(loss_weights) if loss_weights is not None else (1.0,) * len(self.criteria_names)
        self.seg_head = PNN.Linear(backbone_out_channels, num_classes) if num_classes > 0 else nn.Identity()
        self.backbone = backbone
        self.ignore_index = ignore_index

    def criteria(self, seg_logits, segment):   # GPU only, like every op of the engine
        loss = 0
        for name, w in zip(self.criteria_names, self.loss_weights):
            fn = PF.cross_entropy if name == "ce" else PF.lovasz_softmax
            loss = loss + fn(seg_logits, segment, self.ignore_index) * w
        return loss

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


def semseg_eval_counts(seg_logits, input_dict, num_classes: int, ignore_index: int = -1):
    """The per-batch body of SemSegEvaluator.eval (pointcept/engines/hooks/evaluator.py:139-152) in one kernel launch:
    pred = seg_logits.max(1)[1]; with `inverse` / `origin_segment` in the batch (GridSample test mode) predictions are
    carried back to the original points; then intersection_and_union_gpu (pointcept/utils/misc.py:57-69).
    Returns (intersection, union, target), int64 [num_classes] on the device -- ready for the evaluator's all_reduce."""
    from . import ops

    if "inverse" in input_dict.keys():
        assert "origin_segment" in input_dict.keys()
        return ops.seg_eval_hist(seg_logits, input_dict["origin_segment"], num_classes, ignore_index, inverse=input_dict["inverse"])
    return ops.seg_eval_hist(seg_logits, input_dict["segment"], num_classes, ignore_index)
