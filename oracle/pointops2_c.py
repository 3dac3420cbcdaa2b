"""TEST INFRASTRUCTURE (oracle).  CPU stand-in for the compiled extension `pointops2_cuda` of libs/pointops2 (setup.py), so that the
reference's OWN python module libs/pointops2/functions/pointops.py (autograd Functions, N_q = index0.max() + 1, the merged / sorted
rel_idx bookkeeping of the v2 forms, argument orders) can be imported and executed in the authoring container:

    P2 = load_reference_module()          # `pointops2.functions.pointops` of the reference on these stand-ins

Each entry point has the C++ signature the python file calls (caller-allocated, pre-zeroed outputs written in place).  The
arithmetic comes from oracle/pointops2.py (the torch formulations the reference's own operator tests check their kernels against)
and, for the families shared with libs/pointops, from oracle/pointops_c.py; backward entry points evaluate the same formulation
under autograd.  The v1 / v2 / v3 kernel generations compute the same sums (v2 / v3 differ in how threads find their pairs:
per-query offsets + n_max), which is what lets one restatement serve them.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch

from . import pointops2 as F2
from . import pointops_c as C1

REF = os.environ.get("POINTCEPT_REFERENCE", "/root/reference")


def _index_from_offsets(offsets, m):
    counts = (offsets[1:] - offsets[:-1]).long()
    return torch.repeat_interleave(torch.arange(counts.numel()), counts)[:m]


def _grads(fn, inputs, grad_out):
    """d sum(fn(*inputs) * grad_out) / d inputs, in fp32 like the kernels"""
    leaves = [t.detach().clone().requires_grad_(True) for t in inputs]
    with torch.enable_grad():
        out = fn(*leaves)
        return torch.autograd.grad(out, leaves, grad_out)


# ---- families shared with libs/pointops (same kernels, other names)
def furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):
    C1.farthest_point_sampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx)


def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    C1.knn_query_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2)


grouping_forward_cuda = C1.grouping_forward_cuda
grouping_backward_cuda = C1.grouping_backward_cuda
subtraction_forward_cuda = C1.subtraction_forward_cuda
subtraction_backward_cuda = C1.subtraction_backward_cuda
aggregation_forward_cuda = C1.aggregation_forward_cuda
aggregation_backward_cuda = C1.aggregation_backward_cuda
interpolation_forward_cuda = C1.interpolation_forward_cuda
interpolation_backward_cuda = C1.interpolation_backward_cuda


# ---- attention step 1: out[m, h] = q[index0[m], h, :] . k[index1[m], h, :]        (src/attention{,_v2}/attention_cuda_kernel*.cu)
def attention_step1_forward_cuda(N_k, M, h, C, q, k, index0, index1, output):
    output.copy_(F2.attention_step1(q, k, index0, index1))


def attention_step1_backward_cuda(N_q, M, h, C, grad_out, index0, index1, q, k, grad_q, grad_k):
    gq, gk = _grads(lambda a, b: F2.attention_step1(a, b, index0, index1), (q, k), grad_out)
    grad_q.copy_(gq)
    grad_k.copy_(gk)


def attention_step1_forward_cuda_v2(N_k, M, h, C, n_max, q, k, index0_offsets, index1, output):
    attention_step1_forward_cuda(N_k, M, h, C, q, k, _index_from_offsets(index0_offsets, M), index1, output)


def attention_step1_backward_cuda_v2(N_q, M, h, C, n_max, grad_out, index0_offsets, index1, q, k, grad_q, grad_k):
    attention_step1_backward_cuda(N_q, M, h, C, grad_out, _index_from_offsets(index0_offsets, M), index1, q, k, grad_q, grad_k)


# ---- attention step 2: out[index0[m], h, :] += attn[m, h] * v[index1[m], h, :]
def attention_step2_forward_cuda(N, M, h, C, attn, v, index0, index1, output):
    output.copy_(F2.attention_step2(attn, v, index0, index1, output.shape[0]))


def attention_step2_backward_cuda(N, M, h, C, grad_out, index0, index1, attn, v, grad_attn, grad_v):
    ga, gv = _grads(lambda a, b: F2.attention_step2(a, b, index0, index1, grad_out.shape[0]), (attn, v), grad_out)
    grad_attn.copy_(ga)
    grad_v.copy_(gv)


# ---- relative position encoding, step 1: out[m, h] = q[index[m], h, :] . T(m, h, :),  T = sum_a table[rel_idx[m, a], h, :, a]
def dot_prod_with_idx_forward_cuda(N, M, h, hdim, q, index, table, rel_idx, output):
    output.copy_(F2.dot_prod_with_idx(q, index, table, rel_idx))


def dot_prod_with_idx_backward_cuda(N, M, h, hdim, grad_out, q, index, table, rel_idx, grad_q, grad_table):
    gq, gt = _grads(lambda a, t: F2.dot_prod_with_idx(a, index, t, rel_idx), (q, table), grad_out)
    grad_q.copy_(gq)
    grad_table.copy_(gt)


def dot_prod_with_idx_forward_cuda_v2(N, M, h, hdim, n_max, T, q, index_q, k, index_k, table_q, table_k, rel_idx, rel_idx_offsets,
                                      sort_indices, output):
    # the python wrapper's merged / sorted rel_idx bookkeeping must describe the pairs it hands over (pointops.py:499-511)
    assert int(rel_idx_offsets[-1]) == M and sort_indices.numel() == M and rel_idx_offsets.numel() == T + 1
    output.copy_(F2.dot_prod_with_idx_v3(q, index_q, k, index_k, table_q, table_k, rel_idx))


def dot_prod_with_idx_backward_cuda_v2(N, M, h, hdim, n_max, T, grad_out, q, index_q, k, index_k, table_q, table_k, rel_idx,
                                       rel_idx_offsets, sort_indices, grad_q, grad_k, grad_table_q, grad_table_k):
    g = _grads(lambda a, b, tq, tk: F2.dot_prod_with_idx_v3(a, index_q, b, index_k, tq, tk, rel_idx), (q, k, table_q, table_k), grad_out)
    for dst, src in zip((grad_q, grad_k, grad_table_q, grad_table_k), g):
        dst.copy_(src)


def dot_prod_with_idx_forward_cuda_v3(N, M, h, hdim, n_max, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, output):
    output.copy_(F2.dot_prod_with_idx_v3(q, _index_from_offsets(index_q_offsets, M), k, index_k, table_q, table_k, rel_idx))


def dot_prod_with_idx_backward_cuda_v3(N, M, h, hdim, n_max, grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx,
                                       grad_q, grad_k, grad_table_q, grad_table_k):
    iq = _index_from_offsets(index_q_offsets, M)
    g = _grads(lambda a, b, tq, tk: F2.dot_prod_with_idx_v3(a, iq, b, index_k, tq, tk, rel_idx), (q, k, table_q, table_k), grad_out)
    for dst, src in zip((grad_q, grad_k, grad_table_q, grad_table_k), g):
        dst.copy_(src)


# ---- relative position encoding, step 2: out[index0[m], h, :] += attn[m, h] * (v[index1[m], h, :] + T(m, h, :))
def attention_step2_with_rel_pos_value_forward_cuda(N_q, M, h, hdim, attn, v, index0, index1, table, rel_idx, output):
    output.copy_(F2.attention_step2(attn, v, index0, index1, output.shape[0], table, rel_idx))


def attention_step2_with_rel_pos_value_backward_cuda(N_q, M, h, hdim, grad_out, index0, index1, attn, v, table, rel_idx, grad_attn,
                                                     grad_v, grad_table):
    g = _grads(lambda a, b, t: F2.attention_step2(a, b, index0, index1, grad_out.shape[0], t, rel_idx), (attn, v, table), grad_out)
    for dst, src in zip((grad_attn, grad_v, grad_table), g):
        dst.copy_(src)


def attention_step2_with_rel_pos_value_forward_cuda_v2(N, M, h, hdim, n_max, attn, v, index0_offsets, index1, table, rel_idx, output):
    attention_step2_with_rel_pos_value_forward_cuda(N, M, h, hdim, attn, v, _index_from_offsets(index0_offsets, M), index1, table, rel_idx,
                                                    output)


def attention_step2_with_rel_pos_value_backward_cuda_v2(N, M, h, hdim, n_max, grad_out, index0_offsets, index1, attn, v, table, rel_idx,
                                                        grad_attn, grad_v, grad_table):
    attention_step2_with_rel_pos_value_backward_cuda(N, M, h, hdim, grad_out, _index_from_offsets(index0_offsets, M), index1, attn, v,
                                                     table, rel_idx, grad_attn, grad_v, grad_table)


_NAMES = [n for n in dir() if "_cuda" in n and not n.startswith("_")]


def load_reference_module(name: str = "pointops2_reference"):
    """libs/pointops2/functions/pointops.py of the reference executed with `pointops2_cuda` = the stand-ins above (cached under
    `name`).  The file allocates with torch.cuda.FloatTensor / IntTensor and calls .cuda(): the caller points those at their CPU
    equivalents for the duration of its test (see tests/test_oracle_vs_reference.py)."""
    if name in sys.modules:
        return sys.modules[name]
    sys.dont_write_bytecode = True
    path = os.path.join(REF, "libs", "pointops2", "functions", "pointops.py")
    if not os.path.exists(path):
        raise RuntimeError(f"reference not found under {REF}")
    c_mod = types.ModuleType("pointops2_cuda")
    for n in _NAMES:
        setattr(c_mod, n, globals()[n])
    saved = sys.modules.get("pointops2_cuda")
    sys.modules["pointops2_cuda"] = c_mod
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is None:
            sys.modules.pop("pointops2_cuda", None)
        else:
            sys.modules["pointops2_cuda"] = saved
    sys.modules[name] = mod
    return mod
