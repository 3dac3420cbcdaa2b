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
    ball_query_and_group(...), query_and_group(nsample, xyz, new_xyz, feat, idx, offset, new_offset, dilation=0, ...)  (utils.py)
    grouping2(input, idx), interpolation2(xyz, new_xyz, input, offset, new_offset, k=3)
    offset2batch / batch2offset

    ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None)  -> (idx, dist)   HIP (pointops.hip)
    random_ball_query(..., order=None)                                                          -> (idx, dist)   HIP
    subtraction / aggregation / attention_relation_step / attention_fusion_step (PTv1 / PTv2)  differentiable torch / segment ops

Tie order (equal distances: lower index first) is fixed here and implementation-defined in the reference.
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


def grouping2(input, idx):
    """libs/pointops/functions/grouping.py:5-63 (`Grouping.apply`): input [n, c], idx [m, nsample] -> [m, nsample, c]; the
    custom CUDA backward (atomicAdd scatter) is torch's index backward here."""
    m, nsample = idx.shape
    return input[idx.reshape(-1).long()].view(m, nsample, input.shape[1])


def interpolation2(xyz, new_xyz, input, offset, new_offset, k=3):
    """libs/pointops/functions/interpolation.py:30-61: the custom-Function form of `interpolation`; the weights are constants of
    the geometry, so autograd through the weighted gather gives the same gradient."""
    return interpolation(xyz, new_xyz, input, offset, new_offset, k)


def ball_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, max_radio=None, min_radio=0, nsample=None,
                         with_xyz=False):
    """libs/pointops/functions/utils.py:21-39 (argument names as there, `radio` included)."""
    if idx is None:
        assert nsample is not None and offset is not None
        assert max_radio is not None and min_radio is not None
        idx, _ = ball_query(nsample, max_radio, min_radio, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def query_and_group(nsample, xyz, new_xyz, feat, idx, offset, new_offset, dilation=0, with_feat=True, with_xyz=True):
    """libs/pointops/functions/utils.py:42-99: kNN grouping with DILATION -- 1 + (nsample - 1)(dilation + 1) neighbours are
    queried and every (dilation + 1)-th kept; a scene with fewer points than that keeps the stride that still spans it
    (`soft_dilation`).  Returns idx alone when with_feat is False, else (grouped [m, nsample, c (+3)], idx)."""
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        total = 1 + (nsample - 1) * (dilation + 1)
        wide, _ = knn_query(total, xyz, offset, new_xyz, new_offset)              # [m, total]
        ends, new_ends = [int(v) for v in offset.tolist()], [int(v) for v in new_offset.tolist()]
        parts, start, new_start = [], 0, 0
        for end, new_end in zip(ends, new_ends):
            count = end - start
            stride = ((count - 1) / (nsample - 1) - 1) if count < total else dilation      # the reference's soft_dilation
            cols = [int((stride + 1) * j) for j in range(nsample)]
            parts.append(wide[new_start:new_end, cols])
            start, new_start = end, new_end
        idx = torch.cat(parts, dim=0)
    if not with_feat:
        return idx
    m, c = new_xyz.shape[0], feat.shape[1]
    flat = idx.reshape(-1).long()
    grouped_feat = feat[flat].view(m, nsample, c)
    if with_xyz:
        grouped_xyz = xyz[flat].view(m, nsample, 3) - new_xyz.unsqueeze(1)
        return torch.cat((grouped_xyz, grouped_feat), -1), idx
    return grouped_feat, idx


def ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None):
    """libs/pointops/functions/query.py:78-113.  Differences from the CUDA kernel, on purpose: equal distances are ordered by
    ascending index (heap sort leaves it unspecified), and in the sub-sampled branch (more than nsample candidates) the
    returned distance is the candidate's distance -- ball_query_cuda_kernel.cu:120 stores the candidate INDEX there."""
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    return ops.ball_query(int(nsample), max_radius, min_radius, xyz, offset, new_xyz, new_offset)


def random_ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None, order=None):
    """libs/pointops/functions/query.py:29-75: the first nsample in-range points along a random permutation of every scene's
    points (torch.randperm per scene, :47-53; `order=` injects it for tests)."""
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    if order is None:
        parts, s0 = [], 0
        for s1 in offset.tolist():
            parts.append(torch.randperm(s1 - s0, dtype=torch.int32, device=xyz.device) + s0)
            s0 = s1
        order = torch.cat(parts)
    return ops.ball_query(int(nsample), max_radius, min_radius, xyz, offset, new_xyz, new_offset, order=order)


def subtraction(input1, input2, idx):
    """libs/pointops/functions/subtraction.py / src/subtraction/subtraction_cuda_kernel.cu:5-30:
    out[n, s, :] = input1[n, :] - input2[idx[n, s], :]   (differentiable; the CUDA backward scatters with atomics, torch's
    index backward here)."""
    n, ns = idx.shape
    return input1.unsqueeze(1) - input2[idx.reshape(-1).long()].view(n, ns, -1)


def aggregation(input, position, weight, idx):
    """libs/pointops/functions/aggregation.py / src/aggregation/aggregation_cuda_kernel.cu:5-39 (PTv1 vector attention):
    out[n, c] = sum_s (input[idx[n, s], c] + position[n, s, c]) * weight[n, s, c % w_c]."""
    n, ns, c = position.shape
    w_c = weight.shape[-1]
    g = input[idx.reshape(-1).long()].view(n, ns, c) + position
    w = weight.repeat(1, 1, c // w_c) if c != w_c else weight          # channel c uses weight column c % w_c
    return (g * w).sum(1)


def _csr_by_target(index_target, n):
    """edges sorted by target row + CSR pointer: the segmented (atomics-free, fixed-order) form of the scatter-adds"""
    order = torch.sort(index_target.long(), stable=True).indices
    counts = torch.bincount(index_target.long(), minlength=n)
    indptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)])
    return order, indptr


def attention_relation_step(query, key, weight, index_target, index_refer):
    """libs/pointops/functions/attention.py:11-62 / src/attention/attention_cuda_kernel.cu:9-25:
    relation[m, g] = sum_c query[index_target[m], g, c] * key[index_refer[m], g, c] * weight[c]   (differentiable)"""
    return (query[index_target.long()] * key[index_refer.long()] * weight).sum(-1)


def attention_fusion_step(weight, value, index_target, index_refer):
    """libs/pointops/functions/attention.py:64-120 / attention_cuda_kernel.cu:46-62:
    out[index_target[m], g, c] += weight[m, g] * value[index_refer[m], g, c].  The reference accumulates with atomicAdd
    (run-to-run different sums); here the edges are sorted by target and reduced per target row in a fixed order
    (PF.segment_csr "sum": csrc/rows.hip), output [n, g, c]."""
    from . import functional as PF

    n, g, c = value.shape
    order, indptr = _csr_by_target(index_target, n)
    contrib = (weight.unsqueeze(-1) * value[index_refer.long()]).reshape(-1, g * c)
    if contrib.is_cuda:
        out = PF.segment_csr(contrib, indptr, "sum", perm=order)
    else:
        out = torch.zeros((n, g * c), dtype=contrib.dtype).index_add_(0, index_target.long(), contrib)
    return out.view(n, g, c)
