"""HIP-backed mirror of the `pointops` functions the reference's evaluators, testers and SSL heads call
(libs/pointops/functions/{query,sampling,grouping,interpolation,utils}.py; call sites e.g.
pointcept/engines/hooks/evaluator.py:569, engines/test.py:1201, models/sonata/sonata_v1m1_base.py:320,
datasets/modelnet.py:100).  Same names, argument order and return conventions; `pointcept_amd.compat.install()` makes
`import pointops` resolve here.

    knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None) -> (idx [m, nsample] int32, dist [m, nsample] fp32)
    farthest_point_sampling(xyz, offset, new_offset)                 -> idx [new_offset[-1]] int32
    grouping(idx, feat, xyz, new_xyz=None, with_xyz=False)            -> [m, nsample, c (+3)]   (differentiable gather)
    interpolation(xyz, new_xyz, feat, offset, new_offset, k=3)        -> [n, c]  inverse-distance weights over k-NN
    knn_query_and_group(feat, xyz, offset, new_xyz, new_offset, idx=None, nsample=None, with_xyz=False)
    offset2batch / batch2offset

Tie order (equal distances: lower index first) is fixed here and implementation-defined in the reference.  ball_query /
random_ball_query / subtraction / aggregation / attention_*_step (PTv1 / PTv2 only) are not implemented and raise.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import PtcoreError
from .structure import batch2offset, offset2batch  # noqa: F401  (re-exported, libs/pointops/functions/utils.py)


def knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None):
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    return ops.knn_query(int(nsample), xyz, offset, new_xyz, new_offset)


def farthest_point_sampling(xyz, offset, new_offset):
    return ops.farthest_point_sampling(xyz, offset, new_offset)


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """libs/pointops/functions/grouping.py:44-68: rows gathered by idx (-1 -> zeros); with_xyz prepends the neighbour
    offsets xyz[idx] - new_xyz (zeroed for -1 slots)."""
    if new_xyz is None:
        new_xyz = xyz
    m, nsample, c = idx.shape[0], idx.shape[1], feat.shape[1]
    flat = idx.reshape(-1).long()
    present = (flat >= 0)
    safe = flat.clamp(min=0)
    grouped_feat = (feat[safe] * present[:, None].to(feat.dtype)).view(m, nsample, c)
    if not with_xyz:
        return grouped_feat
    grouped_xyz = (xyz[safe] * present[:, None].to(xyz.dtype)).view(m, nsample, 3) - new_xyz.unsqueeze(1)
    grouped_xyz = grouped_xyz * present.view(m, nsample, 1).to(xyz.dtype)
    return torch.cat((grouped_xyz, grouped_feat), -1)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """libs/pointops/functions/interpolation.py:8-27: inverse-distance weighting over the k nearest source points."""
    idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)
    recip = 1.0 / (dist + 1e-8)
    weight = recip / recip.sum(dim=1, keepdim=True)
    out = torch.zeros((new_xyz.shape[0], feat.shape[1]), dtype=torch.float32, device=xyz.device)
    for i in range(k):
        out = out + feat[idx[:, i].long(), :] * weight[:, i].unsqueeze(-1)
    return out


def knn_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, nsample=None, with_xyz=False):
    if idx is None:
        assert nsample is not None
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def _missing(name):
    def f(*args, **kwargs):
        raise PtcoreError(f"pointops.{name} is not implemented by the engine")
    f.__name__ = name
    return f


ball_query = _missing("ball_query")
random_ball_query = _missing("random_ball_query")
subtraction = _missing("subtraction")
aggregation = _missing("aggregation")
attention_relation_step = _missing("attention_relation_step")
attention_fusion_step = _missing("attention_fusion_step")
