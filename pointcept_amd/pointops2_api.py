"""HIP-backed mirror of `pointops2.functions.pointops` (libs/pointops2/functions/pointops.py), the operator module of
Stratified Transformer (pointcept/models/stratified_transformer/stratified_transformer_v1m{1,2}_*.py:21-31 import it as
`import pointops2.pointops as pointops`).  Same names, argument order and return conventions; `pointcept_amd.compat.install()`
makes `import pointops2.pointops` (and `pointops2.functions.pointops`) resolve here.

    furthestsampling(xyz, offset, new_offset)                              -> idx [m] int32                          :16-34
    knnquery(nsample, xyz, new_xyz, offset, new_offset)                    -> (idx [m, nsample] int32, dist fp32)    :37-56
    grouping(input, idx)                                                   -> [m, nsample, c]                        :59-90
    attention_step1(q, k, index0, index1) / attention_step1_v2(q, k, index1, index0_offsets, n_max)      -> [M, h]   :93-258
    attention_step2(attn, v, index0, index1) / attention_step2_v2(...)                                   -> [N, h, d] :261-404
    dot_prod_with_idx(q, index, table, rel_idx)                                                          -> [M, h]   :407-473
    dot_prod_with_idx_v2(q, index_q, k, index_k, table_q, table_k, rel_idx) / _v3(q, index_q_offsets, n_max, k, ...)  :476-755
    attention_step2_with_rel_pos_value(attn, v, index0, index1, table, rel_idx) / _v2(attn, v, index0_offsets, n_max, ...)  :758-961
    queryandgroup, Divide2Patch, subtraction, aggregation, interpolation, interpolation_v2, interpolation2          :964-1193

The v1 / v2 / v3 forms differ in how the pair list arrives (one query index per pair, or CSR offsets + n_max), not in what they
compute; all of them run on the two pair operators of csrc/pointops2.hip (`ptc_pair_dot_*`, `ptc_pair_aggregate_*`).  With
offsets the per-query sums are fixed-order segment loops; gradients that scatter by key or table entry use float atomics
exactly where the reference's kernels use atomicAdd.  fp32, GPU only (no CPU fallback).
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops
from . import pointops_api as _p1


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def _index_from_offsets(offsets: torch.Tensor, m: int) -> torch.Tensor:
    """query index of every pair from the CSR offsets [Nq+1] (the inverse of what the models do with `index_0_offsets`)"""
    counts = (offsets[1:] - offsets[:-1]).long()
    return torch.repeat_interleave(torch.arange(counts.numel(), device=offsets.device, dtype=torch.int32), counts, output_size=m)


# ------------------------------------------------------------------------------------------------ shared families
def furthestsampling(xyz, offset, new_offset):
    return _p1.farthest_point_sampling(xyz, offset, new_offset)


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    if new_xyz is None:
        new_xyz, new_offset = xyz, offset
    return _p1.knn_query(nsample, xyz, offset, new_xyz, new_offset)      # dist already sqrt'ed, as pointops.py:54


def grouping(input, idx):
    """[n, c], [m, nsample] -> [m, nsample, c] (differentiable gather; the CUDA backward scatters with atomicAdd)"""
    m, nsample = idx.shape
    return input[idx.reshape(-1).long()].view(m, nsample, input.shape[1])


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True, return_indx=False):
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    m, c = new_xyz.shape[0], feat.shape[1]
    flat = idx.reshape(-1).long()
    grouped_feat = feat[flat].view(m, nsample, c)
    out = grouped_feat
    if use_xyz:
        grouped_xyz = xyz[flat].view(m, nsample, 3) - new_xyz.unsqueeze(1)
        out = torch.cat((grouped_xyz, grouped_feat), -1)
    return (out, idx) if return_indx else out


def Divide2Patch(nsample, xyz, offset, return_offset=False, anchor_scale=None):
    scale = anchor_scale or nsample
    off = [int(o) for o in offset.tolist()]
    counts, prev, total = [], 0, 0
    for o in off:
        total += (o - prev) // scale
        counts.append(total)
        prev = o
    new_offset = torch.tensor(counts, dtype=torch.int32, device=xyz.device)
    idx = furthestsampling(xyz, offset, new_offset)
    p_idx, _ = knnquery(nsample, xyz, xyz[idx.long()], offset, new_offset)
    return (p_idx, new_offset) if return_offset else p_idx


subtraction = _p1.subtraction
aggregation = _p1.aggregation


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    return _p1.interpolation(xyz, new_xyz, feat, offset, new_offset, k)


def interpolation_v2(xyz, new_xyz, feat, offset, new_offset, k=3):
    idx, _ = knnquery(k, xyz, new_xyz, offset, new_offset)
    dist = torch.sqrt(((new_xyz.unsqueeze(1) - xyz[idx.long()]) ** 2).sum(-1) + 1e-8)
    recip = 1.0 / (dist + 1e-8)
    weight = recip / recip.sum(dim=1, keepdim=True)
    out = torch.zeros((new_xyz.shape[0], feat.shape[1]), dtype=torch.float32, device=xyz.device)
    for i in range(k):
        out = out + feat[idx[:, i].long(), :] * weight[:, i].unsqueeze(-1)
    return out


def interpolation2(xyz, new_xyz, input, offset, new_offset, k=3):
    """pointops.py:1157-1193 (custom Function there; the weights are constants of the geometry, so plain autograd through
    the weighted gather gives the same gradient)"""
    return interpolation(xyz, new_xyz, input, offset, new_offset, k)


# ------------------------------------------------------------------------------------------------ pair operators
class _PairDot(Function):
    """out[m,h] = [qk] q[i0].k[i1] + [tq] q[i0].Tq(m) + [tk] k[i1].Tk(m)"""

    @staticmethod
    def forward(ctx, q, k, table_q, table_k, i0, offsets, i1, rel_idx, with_qk):
        out = ops.pair_dot_fwd(q, k, i0, i1, table_q, table_k, rel_idx, with_qk)
        ctx.save_for_backward(q, k, table_q, table_k, i0, offsets, i1, rel_idx)
        ctx.with_qk = with_qk
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        q, k, table_q, table_k, i0, offsets, i1, rel_idx = ctx.saved_tensors
        need = ctx.needs_input_grad
        dq, dk, dtq, dtk = ops.pair_dot_bwd(g.contiguous(), q, k, i0, offsets, i1, table_q, table_k, rel_idx, ctx.with_qk,
                                            want_q=need[0], want_k=need[1] and k is not None,
                                            want_tq=need[2] and table_q is not None, want_tk=need[3] and table_k is not None)
        return dq, dk, dtq, dtk, None, None, None, None, None


class _PairAggregate(Function):
    """out[n,h,c] = sum_{pairs of n} attn[m,h] (v[i1[m],h,c] + [tv] Tv(m,h,c))"""

    @staticmethod
    def forward(ctx, attn, v, table_v, i0, offsets, i1, rel_idx, n_q):
        out = ops.pair_aggregate_fwd(attn, v, i0, offsets, i1, table_v, rel_idx, n_q)
        ctx.save_for_backward(attn, v, table_v, i0, i1, rel_idx)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        attn, v, table_v, i0, i1, rel_idx = ctx.saved_tensors
        need = ctx.needs_input_grad
        da, dv, dtv = ops.pair_aggregate_bwd(g.contiguous(), attn, v, i0, i1, table_v, rel_idx, want_attn=need[0], want_v=need[1],
                                             want_tv=need[2] and table_v is not None)
        return da, dv, dtv, None, None, None, None, None


def _prep(*ts):
    return tuple(None if t is None else t.contiguous() for t in ts)


def attention_step1(q, k, index0, index1):
    q, k = _prep(q, k)
    return _PairDot.apply(q, k, None, None, _i32(index0).contiguous(), None, _i32(index1).contiguous(), None, True)


def attention_step1_v2(q, k, index1, index0_offsets, n_max):
    q, k = _prep(q, k)
    off, i1 = _i32(index0_offsets).contiguous(), _i32(index1).contiguous()
    return _PairDot.apply(q, k, None, None, _index_from_offsets(off, i1.numel()), off, i1, None, True)


def dot_prod_with_idx(q, index, table, rel_idx):
    q, table = _prep(q, table)
    return _PairDot.apply(q, None, table, None, _i32(index).contiguous(), None, None, _i32(rel_idx).contiguous(), False)


def dot_prod_with_idx_v2(q, index_q, k, index_k, table_q, table_k, rel_idx):
    q, k, table_q, table_k = _prep(q, k, table_q, table_k)
    return _PairDot.apply(q, k, table_q, table_k, _i32(index_q).contiguous(), None, _i32(index_k).contiguous(),
                          _i32(rel_idx).contiguous(), False)


def dot_prod_with_idx_v3(q, index_q_offsets, n_max, k, index_k, table_q, table_k, rel_idx):
    q, k, table_q, table_k = _prep(q, k, table_q, table_k)
    off, i1 = _i32(index_q_offsets).contiguous(), _i32(index_k).contiguous()
    return _PairDot.apply(q, k, table_q, table_k, _index_from_offsets(off, i1.numel()), off, i1, _i32(rel_idx).contiguous(), False)


def attention_step2(attn, v, index0, index1):
    attn, v = _prep(attn, v)
    i0 = _i32(index0).contiguous()
    n_q = int(i0.max().item()) + 1 if i0.numel() else 0          # pointops.py:278: N_q = index0.max().item() + 1 (a host sync there too)
    return _PairAggregate.apply(attn, v, None, i0, None, _i32(index1).contiguous(), None, n_q)


attention_step2_v2 = attention_step2


def attention_step2_with_rel_pos_value(attn, v, index0, index1, table, rel_idx):
    attn, v, table = _prep(attn, v, table)
    i0 = _i32(index0).contiguous()
    n_q = int(i0.max().item()) + 1 if i0.numel() else 0          # pointops.py:776: N_q = index0.max().item() + 1, not N_v
    return _PairAggregate.apply(attn, v, table, i0, None, _i32(index1).contiguous(), _i32(rel_idx).contiguous(), n_q)


def attention_step2_with_rel_pos_value_v2(attn, v, index0_offsets, n_max, index1, table, rel_idx):
    attn, v, table = _prep(attn, v, table)
    off, i1 = _i32(index0_offsets).contiguous(), _i32(index1).contiguous()
    return _PairAggregate.apply(attn, v, table, _index_from_offsets(off, i1.numel()), off, i1, _i32(rel_idx).contiguous(), v.shape[0])
