"""TEST INFRASTRUCTURE.  CPU stand-ins for `pointcept_amd.ops` so that the engine's PYTHON layer -- modules, autograd
wrappers, index bookkeeping, host-fact prefetching, criteria -- can be exercised in the `-m "not gpu"` tier.

    with mock_backend.cpu_ops():
        model = PointTransformerV3(...)      # construct INSIDE the context (activation absorption reads the switches)
        out = model(batch_on_cpu)

Every stand-in restates the documented contract of the op it replaces (pointcept_amd/ops.py docstrings,
include/ptcore.h) on oracle/ functions or plain torch.  NOTHING here tests a kernel -- that is the `-m gpu` tier, which
runs the real library.  The product never imports this module; outside the context manager every op still refuses CPU
tensors.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch

from oracle import losses as olosses
from oracle import maps as omaps
from oracle import ops as oops
from oracle import sfc as osfc


def _np(t):
    return t.detach().cpu().numpy()


def coord_max(grid_coord):
    return grid_coord.max(0).values.to(torch.int64) if grid_coord.shape[0] else torch.zeros(3, dtype=torch.int64)


def serialize_encode(grid_coord, batch, depth, orders):
    return torch.from_numpy(osfc.encode_c(_np(grid_coord).astype(np.int64), None if batch is None else _np(batch).astype(np.int64),
                                          int(depth), tuple(orders)))


def sort_keys(keys, begin_bit, end_bit, want_inverse=True):
    squeeze = keys.dim() == 1
    k2 = keys.reshape(1, -1) if squeeze else keys
    width = end_bit - begin_bit
    masked = (_np(k2).astype(np.uint64) >> np.uint64(begin_bit)) & np.uint64((1 << width) - 1 if width < 64 else 0xFFFFFFFFFFFFFFFF)
    order = np.argsort(masked, axis=1, kind="stable")
    inverse = np.empty_like(order)
    for r in range(order.shape[0]):
        inverse[r, order[r]] = np.arange(order.shape[1])
    o, i = torch.from_numpy(order), torch.from_numpy(inverse)
    if squeeze:
        return o[0], (i[0] if want_inverse else None)
    return o, (i if want_inverse else None)


def patch_pad_maps(offset, offset_host, patch):
    pad, unpad, cu = omaps.pad_maps(np.asarray(offset_host, dtype=np.int64), int(patch))
    return tuple(torch.from_numpy(x) for x in (pad, unpad, cu, omaps.dup_map(pad, unpad)))


def attn_tables(order, inverse, pad, unpad, dup):
    point = order[pad]
    slots = torch.arange(pad.numel())
    t_qkv_fwd = point.to(torch.int32)[None]
    t_proj_bwd = torch.where(unpad[inverse[point]] == slots, point, torch.full_like(point, -1)).to(torch.int32)[None]
    slot = unpad[inverse]
    t_qkv_bwd = torch.stack([slot, dup[inverse]]).to(torch.int32)
    return t_qkv_fwd, t_qkv_bwd, slot.to(torch.int32)[None], t_proj_bwd


def pool_level_counts(code0, order0, batch_shift, n_batch, shifts):
    c = _np(code0).astype(np.uint64)
    b = (c >> np.uint64(batch_shift)).astype(np.int64)
    return torch.tensor([[len(np.unique(c[b == s] >> np.uint64(sh))) for s in range(n_batch)] for sh in shifts], dtype=torch.int64)


def pool_maps(code0, order0, shift, n_cluster=None):
    c = _np(code0) >> shift
    uniq, cluster, counts = np.unique(c, return_inverse=True, return_counts=True)
    assert n_cluster is None or n_cluster == len(uniq), (n_cluster, len(uniq))
    idx_ptr = np.concatenate([[0], np.cumsum(counts)])
    head = _np(order0)[idx_ptr[:-1]]
    return torch.from_numpy(cluster.astype(np.int64)), torch.from_numpy(idx_ptr.astype(np.int64)), torch.from_numpy(head.astype(np.int64))


def pool_child_codes(code, head, shift):
    return code[:, head] >> shift


def gather_rows(src, idx, idx2=None):
    def take(i):
        i = i.long()
        return src[i.clamp(min=0)] * (i >= 0)[:, None].to(src.dtype)
    out = take(idx)
    return out if idx2 is None else out + take(idx2)


def segment_csr_fwd(src, perm, indptr, reduce):
    rows = src if perm is None else src[perm.long()]
    indptr = indptr.long()
    n_seg = indptr.numel() - 1
    out = oops.segment_csr(rows.detach(), indptr, reduce)
    arg = None
    if reduce in ("max", "min"):
        arg = torch.zeros((n_seg, src.shape[1]), dtype=torch.int32)
        for s in range(n_seg):                                       # small test inputs only
            a, b = int(indptr[s]), int(indptr[s + 1])
            if b > a:
                seg = rows[a:b]
                first = (seg == out[s][None]).float().argmax(0) + a   # first arg-max / arg-min row of the segment
                arg[s] = (first if perm is None else perm.long()[first]).to(torch.int32)
    return out, arg


def segment_csr_bwd(grad_out, perm, indptr, arg, n_src, reduce, covers_all=False):
    indptr = indptr.long()
    n_seg, c = indptr.numel() - 1, grad_out.shape[1]
    g = torch.zeros((int(n_src), c), dtype=grad_out.dtype)
    counts = indptr[1:] - indptr[:-1]
    if reduce in ("max", "min"):
        live = (counts > 0)[:, None].expand(-1, c)
        cols = torch.arange(c)[None].expand(n_seg, -1)
        g.index_put_((arg.long()[live], cols[live]), grad_out[live], accumulate=True)
        return g
    seg = torch.repeat_interleave(torch.arange(n_seg), counts)
    rows = torch.arange(int(indptr[-1])) if perm is None else perm.long()[: int(indptr[-1])]
    val = grad_out[seg]
    if reduce == "mean":
        val = val / counts[seg][:, None].to(val.dtype)
    g.index_add_(0, rows, val)
    return g


class HashTable:
    def __init__(self, indices):
        assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.shape[1] == 4
        self.indices = indices.contiguous()


def rulebook_subm(indices, ksize, table=None):
    ind = table.indices if table is not None else indices
    return torch.from_numpy(oops.subm_rulebook(_np(ind), int(ksize)))


def rulebook_down(indices, coord_bits, batch_bits):
    out_indices, _, nbr_down, nbr_up = oops.down_rulebook(_np(indices))
    return torch.from_numpy(out_indices), torch.from_numpy(nbr_down), torch.from_numpy(nbr_up)


def spconv_fwd(feat, weight, bias, nbr, blk=None):
    f, w = feat.float(), weight.float()
    if nbr is None:      # the dense GEMM behind nn.Linear: the same torch call as the oracle model's (bit-identical fp32 sums -- the shrunk
        out = torch.nn.functional.linear(f, w[:, 0, :], None if bias is None else bias.float())      # dry-run scenes are ill-conditioned)
    else:
        out = oops.gather_conv(f, w, None if bias is None else bias.float(), _np(nbr))
    return out.to(feat.dtype)


def spconv_wgrad(feat, dout, nbr, want_bias=False, blk=None):
    f, g = feat.float(), dout.float()
    if nbr is None:
        dw = (g.t() @ f)[:, None, :]
    else:
        fpad = torch.cat([f, f.new_zeros(1, f.shape[1])])
        dw = torch.stack([g.t() @ fpad[nbr[k].long()] for k in range(nbr.shape[0])], dim=1)
    return (dw, g.sum(0)) if want_bias else dw


def attn_varlen_fwd(qkv, cu_seqlens, max_seqlen, softmax_scale, dropout_p=0.0, seed=0):
    # f16 qkv, head_dim 16: the reference's casts around its bf16 kernel, flash_attn(qkv.to(bfloat16)).to(qkv.dtype); other head dims:
    # f16 operands as they are (LitePT's call site)
    arith = torch.bfloat16 if qkv.shape[-1] == 16 else qkv.dtype
    out, lse = oops.attention_varlen(qkv.to(arith).float(), cu_seqlens.tolist(), float(softmax_scale), return_lse=True,
                                     dropout_p=dropout_p, seed=seed)
    return out.to(arith).to(qkv.dtype), lse


def attn_varlen_bwd(qkv, out, dout, lse, cu_seqlens, max_seqlen, softmax_scale, dropout_p=0.0, seed=0):
    arith = torch.bfloat16 if qkv.shape[-1] == 16 else qkv.dtype
    q = qkv.detach().to(arith).float().requires_grad_(True)
    with torch.enable_grad():
        o = oops.attention_varlen(q, cu_seqlens.tolist(), float(softmax_scale), dropout_p=dropout_p, seed=seed)
    o.backward(dout.to(arith).float())
    return q.grad.to(arith).to(qkv.dtype)


def knn_query(nsample, xyz, offset, new_xyz, new_offset):
    from oracle import pointops as opo

    i, d = opo.knn_query(int(nsample), _np(xyz), _np(offset), _np(new_xyz), _np(new_offset))
    return torch.from_numpy(i), torch.from_numpy(d)


def ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz, new_offset, order=None):
    from oracle import pointops as opo

    i, d = opo.ball_query(int(nsample), float(max_radius), float(min_radius), _np(xyz), _np(offset), _np(new_xyz), _np(new_offset),
                          order=None if order is None else _np(order))
    return torch.from_numpy(i), torch.from_numpy(d)


def farthest_point_sampling(xyz, offset, new_offset):
    from oracle import pointops as opo

    return torch.from_numpy(opo.farthest_point_sampling(_np(xyz), _np(offset), _np(new_offset)))


def _pair_dot(q, k, i0, i1, tq, tk, rel, with_qk):
    from oracle import pointops2 as o2

    out = 0
    if with_qk:
        out = out + o2.attention_step1(q, k, i0, i1)
    if tq is not None:
        out = out + o2.dot_prod_with_idx(q, i0, tq, rel)
    if tk is not None:
        out = out + o2.dot_prod_with_idx(k, i1, tk, rel)
    return out


def pair_dot_fwd(q, k, i0, i1, table_q, table_k, rel_idx, with_qk):
    return _pair_dot(q, k, i0, i1, table_q, table_k, rel_idx, with_qk).detach()


def pair_dot_bwd(g, q, k, i0, offsets, i1, table_q, table_k, rel_idx, with_qk, want_q=True, want_k=True, want_tq=True, want_tk=True):
    leaves = [None if t is None else t.detach().clone().requires_grad_(True) for t in (q, k, table_q, table_k)]
    with torch.enable_grad():
        out = _pair_dot(leaves[0], leaves[1], i0, i1, leaves[2], leaves[3], rel_idx, with_qk)
        live = [t for t in leaves if t is not None]
        gr = dict(zip([id(t) for t in live], torch.autograd.grad(out, live, g, allow_unused=True)))
    res = [None if t is None else (gr[id(t)] if gr[id(t)] is not None else torch.zeros_like(t)) for t in leaves]
    return tuple(r if w else None for r, w in zip(res, (want_q, want_k, want_tq, want_tk)))


def pair_aggregate_fwd(attn, v, i0, offsets, i1, table_v, rel_idx, n_q):
    from oracle import pointops2 as o2

    return o2.attention_step2(attn, v, i0, i1, int(n_q), table_v, rel_idx).detach()


def pair_aggregate_bwd(g, attn, v, i0, i1, table_v, rel_idx, want_attn=True, want_v=True, want_tv=True):
    from oracle import pointops2 as o2

    leaves = [None if t is None else t.detach().clone().requires_grad_(True) for t in (attn, v, table_v)]
    with torch.enable_grad():
        out = o2.attention_step2(leaves[0], leaves[1], i0, i1, g.shape[0], leaves[2], rel_idx)
        live = [t for t in leaves if t is not None]
        gr = dict(zip([id(t) for t in live], torch.autograd.grad(out, live, g)))
    res = [None if t is None else gr[id(t)] for t in leaves]
    return tuple(r if w else None for r, w in zip(res, (want_attn, want_v, want_tv)))


def rope3d_(tokens, positions, base, fwd):
    from oracle import pointrope as orope

    H, D = tokens.shape[-2], tokens.shape[-1]
    t = tokens.detach().float().reshape(1, -1, H, D).numpy()
    out = orope.pointrope(t, positions.reshape(1, -1, 3).numpy(), float(base), float(fwd))
    with torch.no_grad():
        tokens.copy_(torch.from_numpy(out).reshape(tokens.shape).to(tokens.dtype))
    return tokens


def rope3d_xyz(qkv, xyz, inv_freq, rot_slabs, sign, out_dtype=None):
    from oracle import pointrope as orope

    n, S, H, D = qkv.shape
    t = qkv.detach().float().numpy()
    out = t.copy()
    rot = orope.rope_xyz(t[:, :rot_slabs].reshape(n, rot_slabs * H, D), xyz.numpy(), inv_freq.numpy(), sign)
    out[:, :rot_slabs] = rot.reshape(n, rot_slabs, H, D)
    return torch.from_numpy(out).to(out_dtype or qkv.dtype)


def cross_entropy_fwd(logits, target, ignore_index):
    lg = logits.float()
    lse = torch.logsumexp(lg, dim=1)
    valid = (target != ignore_index) & (target >= 0) & (target < lg.shape[1])
    picked = lg.gather(1, target.clamp(0, lg.shape[1] - 1)[:, None])[:, 0]
    return ((lse - picked) * valid).sum(), valid.float().sum(), lse


def cross_entropy_bwd(logits, target, lse, scale, ignore_index):
    lg = logits.float()
    valid = (target != ignore_index) & (target >= 0) & (target < lg.shape[1])
    p = torch.exp(lg - lse[:, None])
    p[torch.arange(lg.shape[0])[valid], target[valid]] -= 1.0
    return (p * valid[:, None] * scale.float()).to(logits.dtype)


def lovasz_softmax(logits, target, ignore_index):
    loss, d = olosses.lovasz_softmax(_np(logits.float()), _np(target), int(ignore_index))
    return torch.tensor(loss, dtype=torch.float32), torch.from_numpy(d).float()


def column_sum(x):
    return x.float().sum(0)


def voxel_keys(coord, grid_size):
    from oracle import voxelize

    v = voxelize.voxels(_np(coord), float(grid_size))
    return (torch.from_numpy(v["grid_coord"]), torch.from_numpy(v["min_coord"].astype(np.int64)),
            torch.from_numpy(v["key"].view(np.int64).copy()))


# ---- edge-list operators of libs/pointops (csrc/pointops_edges.hip): plain torch restatements of the reference kernels' sums
def _edge_src_rows(src, idx):
    flat = idx.reshape(-1).long()
    ok = (flat >= 0) & (flat < src.shape[0])
    return src[flat.clamp(0, max(src.shape[0] - 1, 0))] * ok[:, None].to(src.dtype), ok


def edge_rows(mode, src, a, idx, out=None, out_col0=0):
    m, ns = idx.shape
    c = src.shape[1]
    rows, ok = _edge_src_rows(src.float(), idx)
    rows = rows.view(m, ns, c)
    if mode == 1:
        rows = a.float()[:, None, :] - rows
    elif mode == 2:
        rows = (rows - a.float()[:, None, :]) * ok.view(m, ns, 1).float()
    if out is None:
        return rows.contiguous()
    out[:, :, out_col0:out_col0 + c] = rows
    return out


def edge_reduce(mode, src, pos, w, idx, m, nsample, c, w_c=1, pos_stride=0, pos_col0=0):
    if mode == 2:
        g = pos.reshape(m, nsample, -1)[:, :, pos_col0:pos_col0 + c].float()
        if idx is not None:
            g = g * (idx.reshape(m, nsample, 1) >= 0).float()
        return g.sum(1)
    rows, _ = _edge_src_rows(src.float(), idx)
    rows = rows.view(m, nsample, c)
    if mode == 0:
        return (rows * w.float().view(m, nsample, 1)).sum(1)
    return ((rows + pos.float()) * w.float().repeat(1, 1, c // w_c)).sum(1)


class EdgeCSR:
    def __init__(self, idx, n_src):
        self.idx, self.n_src = idx.to(torch.int32), int(n_src)

    def build(self):
        return self


def edge_scatter_bwd(mode, csr, g, w, nsample, c, w_c=1, g_col0=0):
    idx = csr.idx.reshape(-1).long()
    g = g.float().reshape(-1, g.shape[-1])[:, g_col0:g_col0 + c]
    if mode >= 2:
        g = g.repeat_interleave(nsample, dim=0)
    if mode == 1:
        g = -g
    elif mode == 2:
        g = g * w.float().reshape(-1, 1)
    elif mode == 3:
        g = g * w.float().reshape(-1, w_c).repeat(1, c // w_c)
    ok = (idx >= 0) & (idx < csr.n_src)
    return torch.zeros(csr.n_src, c, dtype=torch.float32).index_add_(0, idx[ok], g[ok])


def layer_norm_fwd(x, gamma, beta, eps, out_dtype):
    xf = x.float()
    y = torch.nn.functional.layer_norm(xf, (xf.shape[1],), None if gamma is None else gamma.float(), None if beta is None else beta.float(), eps)
    return y.to(out_dtype), xf.mean(1), torch.rsqrt(xf.var(1, unbiased=False) + eps)


def layer_norm_bwd(dy, x, mean, rstd, gamma, want_affine=True):
    # torch's own backward of the same call (the oracle model's op): eps is recovered from the saved statistics' definition
    xf = x.detach().float().requires_grad_(True)
    g = None if gamma is None else gamma.detach().float().requires_grad_(True)
    b = None if gamma is None else torch.zeros_like(g).requires_grad_(True)
    eps = float((1.0 / rstd[0].double() ** 2 - xf[0].double().var(unbiased=False)).clamp(min=0)) if xf.shape[0] else 1e-5
    with torch.enable_grad():
        y = torch.nn.functional.layer_norm(xf, (xf.shape[1],), g, b, eps)
    grads = torch.autograd.grad(y, [xf] + ([g, b] if g is not None else []), dy.float())
    if gamma is None:
        xh = (x.float() - mean[:, None]) * rstd[:, None]
        return grads[0].to(x.dtype), ((dy.float() * xh).sum(0) if want_affine else None), (dy.float().sum(0) if want_affine else None)
    return grads[0].to(x.dtype), (grads[1] if want_affine else None), (grads[2] if want_affine else None)


def pair_dot_weighted(a, b, w, ia, ib):
    r = a.float()[ia.reshape(-1).long()] * b.float()[ib.reshape(-1).long()]
    return (r if w is None else r * w.float().view(1, 1, -1)).sum(-1)


def pair_segment_sum(s, b, w, self_rows, csr, oidx, want_prod=False):
    rows = csr.idx.reshape(-1).long()
    A = torch.zeros((csr.n_src,) + tuple(b.shape[1:]), dtype=torch.float32).index_add_(0, rows, s.float().unsqueeze(-1) * b.float()[oidx.reshape(-1).long()])
    prod = self_rows.float() * A if want_prod else None
    return (A if w is None else A * w.float().view(1, 1, -1)), prod


def aggregation_edge_bwd(src, pos, w, idx, g):
    m, ns, c = pos.shape
    w_c = w.shape[-1]
    rows, _ = _edge_src_rows(src.float(), idx)
    x = rows.view(m, ns, c) + pos.float()
    gp = g.float()[:, None, :] * w.float().repeat(1, 1, c // w_c)
    gw = (g.float()[:, None, :] * x).view(m, ns, c // w_c, w_c).sum(2)
    return gp, gw


_STANDINS = dict(
    coord_max=coord_max, serialize_encode=serialize_encode, sort_keys=sort_keys, patch_pad_maps=patch_pad_maps,
    attn_tables=attn_tables, pool_level_counts=pool_level_counts, pool_maps=pool_maps, pool_child_codes=pool_child_codes,
    gather_rows=gather_rows, segment_csr_fwd=segment_csr_fwd, segment_csr_bwd=segment_csr_bwd, HashTable=HashTable,
    rulebook_subm=rulebook_subm, rulebook_down=rulebook_down, spconv_fwd=spconv_fwd, spconv_wgrad=spconv_wgrad,
    attn_varlen_fwd=attn_varlen_fwd, attn_varlen_bwd=attn_varlen_bwd, cross_entropy_fwd=cross_entropy_fwd, rope3d_xyz=rope3d_xyz, rope3d_=rope3d_, knn_query=knn_query, ball_query=ball_query,
    farthest_point_sampling=farthest_point_sampling, pair_dot_fwd=pair_dot_fwd, pair_dot_bwd=pair_dot_bwd,
    pair_aggregate_fwd=pair_aggregate_fwd, pair_aggregate_bwd=pair_aggregate_bwd,
    attn_rope_supported=lambda d, k: False,      # CPU tier: the rotation pass + the attention stand-in (the same arithmetic)
    attn_hd_supported=lambda d, k: 16 <= d <= 64 and k <= (1024 if d <= 32 else 672 if d <= 48 else 512),
    edge_rows=edge_rows, edge_reduce=edge_reduce, EdgeCSR=EdgeCSR, edge_scatter_bwd=edge_scatter_bwd, aggregation_edge_bwd=aggregation_edge_bwd,
    pair_dot_weighted=pair_dot_weighted, pair_segment_sum=pair_segment_sum,
    cross_entropy_bwd=cross_entropy_bwd, lovasz_softmax=lovasz_softmax, column_sum=column_sum, voxel_keys=voxel_keys,
    layer_norm_supported=lambda c: False, layer_norm_joint_available=lambda c: False, layer_norm_available=lambda c: c % 2 == 0 and c <= 1024, layer_norm_fwd=layer_norm_fwd, layer_norm_bwd=layer_norm_bwd, batch_norm_supported=lambda c, dt: False, gather_rows_add_supported=lambda s_, a_: False, mlp_supported=lambda c, dt: False, linear_supported_ex=lambda a, b, dt: False)


@contextlib.contextmanager
def cpu_ops():
    """pointcept_amd.ops -> the stand-ins above; the layers' "GPU only" guards -> no-ops.  Restored on exit."""
    from pointcept_amd import nn as PNN
    from pointcept_amd import flash_attn_api, ops, spconv_api

    saved = {k: getattr(ops, k) for k in _STANDINS}
    guards = [(PNN, "_require_gpu", PNN._require_gpu), (spconv_api, "_require_gpu", spconv_api._require_gpu),
              (flash_attn_api, "_require_gpu", flash_attn_api._require_gpu)]
    try:
        for k, v in _STANDINS.items():
            setattr(ops, k, v)
        for mod, name, _ in guards:
            setattr(mod, name, lambda *a, **kw: None)
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
        for mod, name, fn in guards:
            setattr(mod, name, fn)
