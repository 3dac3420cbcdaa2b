"""HIP-backed mirror of the `pointops` functions the reference's evaluators, testers and SSL heads call
(libs/pointops/functions/{query,sampling,grouping,interpolation,utils}.py; call sites e.g.
pointcept/engines/hooks/evaluator.py:569, engines/test.py:1201, models/sonata/sonata_v1m1_base.py:320,
datasets/modelnet.py:100).  Same names, argument order and return conventions; `pointcept_amd.compat.install()` makes
`import pointops` resolve here.

    knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None) -> (idx [m, nsample] int32, dist [m, nsample] fp32)
    farthest_point_sampling(xyz, offset, new_offset)                 -> idx [new_offset[-1]] int32
    grouping(idx, feat, xyz, new_xyz=None, with_xyz=False)            -> [m, nsample, c (+3)]   HIP (pointops_edges.hip), fwd + bwd
    interpolation(xyz, new_xyz, feat, offset, new_offset, k=3)        -> [n, c]  inverse-distance weights over k-NN; HIP fwd + bwd
    knn_query_and_group(feat, xyz, offset, new_xyz, new_offset, idx=None, nsample=None, with_xyz=False)
    ball_query_and_group(...), query_and_group(nsample, xyz, new_xyz, feat, idx, offset, new_offset, dilation=0, ...)  (utils.py)
    grouping2(input, idx), interpolation2(xyz, new_xyz, input, offset, new_offset, k=3)
    offset2batch / batch2offset

    ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None)  -> (idx, dist)   HIP (pointops.hip)
    random_ball_query(..., order=None)                                                          -> (idx, dist)   HIP
    subtraction / aggregation (PTv1)                                                            HIP (pointops_edges.hip), fwd + bwd,
                                                                                                 gradients as segmented sums (no atomics)
    attention_relation_step / attention_fusion_step (PTv2)                                      differentiable torch / segment ops

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


class _EdgeGather(torch.autograd.Function):
    """out[t, s] = [xyz[j] - new_xyz[t] |] feat[j] for j = idx[t, s] (zeros for j < 0): ptc_edge_rows_fwd; every gradient a segmented sum
    over the edges sorted by source row (ptc_edge_scatter_bwd): fixed order, where grouping_backward_cuda scatters with atomicAdd."""

    @staticmethod
    def forward(ctx, feat, xyz, new_xyz, idx):
        idx = idx.to(torch.int32).contiguous()
        m, ns = idx.shape
        c = feat.shape[1]
        f32 = feat.float()
        if xyz is None:
            out = ops.edge_rows(0, f32, None, idx)
        else:
            out = torch.empty((m, ns, 3 + c), dtype=torch.float32, device=feat.device)
            ops.edge_rows(2, xyz.float(), new_xyz.float(), idx, out=out, out_col0=0)
            ops.edge_rows(0, f32, None, idx, out=out, out_col0=3)
        ctx.save_for_backward(idx)
        ctx.meta = (feat.shape[0], c, xyz is not None, None if xyz is None else xyz.shape[0], feat.dtype)
        # the reference concatenates fp32 coordinate offsets with feat (torch.cat promotes): half-precision features under autocast must
        # not drag the relative coordinates down to 16 bits -- only the feat-only result goes back to feat's dtype
        return out.to(feat.dtype if xyz is None else torch.promote_types(xyz.dtype, feat.dtype))

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n, c, with_xyz, n_xyz, dtype = ctx.meta
        m, ns = idx.shape
        g = g.float().contiguous()
        col0 = 3 if with_xyz else 0
        d_feat = d_xyz = d_new = None
        if ctx.needs_input_grad[0]:
            d_feat = ops.edge_scatter_bwd(0, ops.EdgeCSR(idx, n), g, None, ns, c, g_col0=col0).to(dtype)
        if with_xyz and ctx.needs_input_grad[1]:
            d_xyz = ops.edge_scatter_bwd(0, ops.EdgeCSR(idx, n_xyz), g, None, ns, 3, g_col0=0)
        if with_xyz and ctx.needs_input_grad[2]:
            d_new = -ops.edge_reduce(2, None, g, None, idx, m, ns, 3, pos_stride=g.shape[-1], pos_col0=0)
        return d_feat, d_xyz, d_new, None


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """libs/pointops/functions/grouping.py:44-68: rows gathered by idx (-1 -> zeros); with_xyz prepends the neighbour
    offsets xyz[idx] - new_xyz (zeroed for -1 slots).  One kernel per operand, written straight into the [m, nsample, 3 + c] result."""
    if new_xyz is None:
        new_xyz = xyz
    if with_xyz:
        return _EdgeGather.apply(feat, xyz, new_xyz, idx)
    return _EdgeGather.apply(feat, None, None, idx)


class _EdgeInterpolate(torch.autograd.Function):
    """out[t] = sum_i weight[t, i] feat[idx[t, i]] (interpolation_forward_cuda); gradient of feat = segmented sum over the edges sorted by
    source row (interpolation_backward_cuda without its atomics).  The weights are constants of the geometry (interpolation.py:30-61)."""

    @staticmethod
    def forward(ctx, feat, idx, weight):
        idx = idx.to(torch.int32).contiguous()
        weight = weight.float().contiguous()
        ctx.save_for_backward(idx, weight)
        ctx.meta = (feat.shape[0], feat.shape[1])
        return ops.edge_reduce(0, feat.float().contiguous(), None, weight, idx, idx.shape[0], idx.shape[1], feat.shape[1])

    @staticmethod
    def backward(ctx, g):
        idx, weight = ctx.saved_tensors
        n, c = ctx.meta
        return ops.edge_scatter_bwd(2, ops.EdgeCSR(idx, n), g.float().contiguous(), weight, idx.shape[1], c), None, None


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """libs/pointops/functions/interpolation.py:8-27: inverse-distance weighting over the k nearest source points (fp32 result)."""
    idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)
    recip = 1.0 / (dist + 1e-8)
    weight = recip / recip.sum(dim=1, keepdim=True)
    return _EdgeInterpolate.apply(feat, idx, weight)


def knn_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, nsample=None, with_xyz=False):
    if idx is None:
        assert nsample is not None
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def grouping2(input, idx):
    """libs/pointops/functions/grouping.py:5-63 (`Grouping.apply`): input [n, c], idx [m, nsample] -> [m, nsample, c]; the
    custom CUDA backward (atomicAdd scatter) is the segmented sum of `_EdgeGather` here."""
    return _EdgeGather.apply(input, None, None, idx)


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
    if with_xyz:
        return _EdgeGather.apply(feat, xyz, new_xyz, idx), idx
    return _EdgeGather.apply(feat, None, None, idx), idx


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


class _EdgeSubtract(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input1, input2, idx):
        idx = idx.to(torch.int32).contiguous()
        ctx.save_for_backward(idx)
        ctx.meta = (input2.shape[0], input1.shape[1])
        return ops.edge_rows(1, input2.float().contiguous(), input1.float().contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n2, c = ctx.meta
        m, ns = idx.shape
        g = g.float().contiguous()
        d1 = ops.edge_reduce(2, None, g, None, None, m, ns, c, pos_stride=c, pos_col0=0) if ctx.needs_input_grad[0] else None
        d2 = ops.edge_scatter_bwd(1, ops.EdgeCSR(idx, n2), g, None, ns, c) if ctx.needs_input_grad[1] else None
        return d1, d2, None


def subtraction(input1, input2, idx):
    """libs/pointops/functions/subtraction.py / src/subtraction/subtraction_cuda_kernel.cu:5-36:
    out[n, s, :] = input1[n, :] - input2[idx[n, s], :]; backward: row sums for input1, the negated segmented sum over the edges of every
    source row for input2 (the CUDA backward scatters both with atomicAdd)."""
    return _EdgeSubtract.apply(input1, input2, idx)


class _EdgeAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, position, weight, idx):
        idx = idx.to(torch.int32).contiguous()
        input, position, weight = input.float().contiguous(), position.float().contiguous(), weight.float().contiguous()
        n, ns, c = position.shape
        ctx.save_for_backward(input, position, weight, idx)
        return ops.edge_reduce(1, input, position, weight, idx, n, ns, c, w_c=weight.shape[-1])

    @staticmethod
    def backward(ctx, g):
        input, position, weight, idx = ctx.saved_tensors
        n, ns, c = position.shape
        g = g.float().contiguous()
        d_in = ops.edge_scatter_bwd(3, ops.EdgeCSR(idx, input.shape[0]), g, weight, ns, c, w_c=weight.shape[-1]) if ctx.needs_input_grad[0] else None
        d_pos, d_w = ops.aggregation_edge_bwd(input, position, weight, idx, g)
        return d_in, d_pos, d_w, None


def aggregation(input, position, weight, idx):
    """libs/pointops/functions/aggregation.py / src/aggregation/aggregation_cuda_kernel.cu:5-45 (PTv1 vector attention):
    out[n, c] = sum_s (input[idx[n, s], c] + position[n, s, c]) * weight[n, s, c % w_c]; three gradients, none of them atomic."""
    return _EdgeAggregate.apply(input, position, weight, idx)


class _AttnRelation(torch.autograd.Function):
    """relation[m, g] = sum_c query[it[m], g, c] key[ir[m], g, c] weight[c]  on csrc/pointops_edges.hip (ptc_pair_dot_weighted;
    the three gradients are segmented sums over the pairs sorted by the row they scatter to: ptc_pair_segment_sum)"""

    @staticmethod
    def forward(ctx, query, key, weight, index_target, index_refer):
        ctx.save_for_backward(query, key, weight, index_target, index_refer)
        return ops.pair_dot_weighted(query, key, weight, index_target, index_refer)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        query, key, weight, it, ir = ctx.saved_tensors
        g = g.float().contiguous()
        dq = dk = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            dq, qa = ops.pair_segment_sum(g, key, weight, query, ops.EdgeCSR(it.view(-1, 1), query.shape[0]), ir, want_prod=ctx.needs_input_grad[2])
            if ctx.needs_input_grad[2]:        # d weight[c] = sum_{n, g} query[n, g, c] A[n, g, c]: one fixed-order column sum
                dw = ops.column_sum(qa.view(-1, qa.shape[-1])).to(weight.dtype).view_as(weight)
            dq = dq.to(query.dtype) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1]:
            dk, _ = ops.pair_segment_sum(g, query, weight, None, ops.EdgeCSR(ir.view(-1, 1), key.shape[0]), it)
            dk = dk.to(key.dtype)
        return dq, dk, dw, None, None


def attention_relation_step(query, key, weight, index_target, index_refer):
    """libs/pointops/functions/attention.py:11-62 / src/attention/attention_cuda_kernel.cu:9-45:
    relation[m, g] = sum_c query[index_target[m], g, c] * key[index_refer[m], g, c] * weight[c]   (differentiable in query, key and
    weight; where the reference scatters its three gradients with atomicAdd these are fixed-order segmented sums)"""
    return _AttnRelation.apply(query, key, weight, index_target, index_refer)


class _AttnFusion(torch.autograd.Function):
    """out[n, g, c] = sum_{m: it[m] = n} weight[m, g] value[ir[m], g, c]: a segmented sum over the pairs sorted by target row"""

    @staticmethod
    def forward(ctx, weight, value, index_target, index_refer):
        ctx.save_for_backward(weight, value, index_target, index_refer)
        out, _ = ops.pair_segment_sum(weight, value, None, None, ops.EdgeCSR(index_target.view(-1, 1), value.shape[0]), index_refer)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        weight, value, it, ir = ctx.saved_tensors
        g = g.float().contiguous()
        dweight = dvalue = None
        if ctx.needs_input_grad[0]:
            dweight = ops.pair_dot_weighted(g, value, None, it, ir).to(weight.dtype)
        if ctx.needs_input_grad[1]:
            dvalue, _ = ops.pair_segment_sum(weight, g, None, None, ops.EdgeCSR(ir.view(-1, 1), value.shape[0]), it)
            dvalue = dvalue.to(value.dtype)
        return dweight, dvalue, None, None


def attention_fusion_step(weight, value, index_target, index_refer):
    """libs/pointops/functions/attention.py:64-120 / attention_cuda_kernel.cu:46-82:
    out[index_target[m], g, c] += weight[m, g] * value[index_refer[m], g, c], output [n, g, c] (n = value rows, as the reference
    allocates it).  The reference accumulates with atomicAdd (run-to-run different sums); here the pairs are sorted by target and each
    output row adds its pairs in ascending order (ptc_pair_segment_sum), and so do both gradients.  GPU only, like every engine op."""
    return _AttnFusion.apply(weight, value, index_target, index_refer)
