"""TEST INFRASTRUCTURE (oracle).  CPU stand-in for the compiled extension `pointops._C` of libs/pointops (setup.py:27), so that the
reference's OWN python package (libs/pointops/functions/*.py: the autograd Functions, argument handling, sqrt / weights / masks
around the kernels) can be imported and executed in the authoring container:

    P = load_reference_package()          # the reference's `pointops` package on these stand-ins
    P.knn_query(...), P.grouping(...), P.aggregation(...), ...

Every function below restates ONE CUDA kernel of libs/pointops/src/*/ *_cuda_kernel.cu with the C++ entry point's signature
(caller-allocated, pre-zeroed outputs written in place) -- "parity unpinned" for the kernels themselves (CUDA only, cannot run
here); what this buys is that everything ABOVE the kernels is the reference's own code.  Ties in the neighbour searches: lower
index first (the kernels leave it to the scheduler).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

from . import pointops as _np_ops

REF = os.environ.get("POINTCEPT_REFERENCE", "/root/reference")


def _i(t):
    return t.detach().cpu().numpy()


# ---- src/knn_query/knn_query_cuda_kernel.cu:60-108 ; src/ball_query/ball_query_cuda_kernel.cu:59-123 ; random_ball_query :58-108
def knn_query_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    i, d = _np_ops.knn_query(int(nsample), _i(xyz), _i(offset), _i(new_xyz), _i(new_offset))
    idx.copy_(torch.from_numpy(i))
    dist2.copy_(torch.from_numpy(d * d))          # the wrapper takes the square root (functions/query.py:28)


def ball_query_cuda(m, nsample, min_radius, max_radius, xyz, new_xyz, offset, new_offset, idx, dist2):
    i, d = _np_ops.ball_query(int(nsample), float(max_radius), float(min_radius), _i(xyz), _i(offset), _i(new_xyz), _i(new_offset))
    idx.copy_(torch.from_numpy(i))
    dist2.copy_(torch.from_numpy(d * d))


def random_ball_query_cuda(m, nsample, min_radius, max_radius, order, xyz, new_xyz, offset, new_offset, idx, dist2):
    i, d = _np_ops.ball_query(int(nsample), float(max_radius), float(min_radius), _i(xyz), _i(offset), _i(new_xyz), _i(new_offset),
                              order=_i(order))
    idx.copy_(torch.from_numpy(i))
    dist2.copy_(torch.from_numpy(d * d))


# ---- src/sampling/sampling_cuda_kernel.cu:15-122
def farthest_point_sampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):
    idx.copy_(torch.from_numpy(_np_ops.farthest_point_sampling(_i(xyz), _i(offset), _i(new_offset))))


# ---- src/grouping/grouping_cuda_kernel.cu (output[m, s, c] = input[idx[m, s], c]; backward: atomicAdd scatter)
def grouping_forward_cuda(m, nsample, c, input, idx, output):
    output.copy_(input[idx.reshape(-1).long()].view(m, nsample, c))


def grouping_backward_cuda(m, nsample, c, grad_output, idx, grad_input):
    grad_input.index_add_(0, idx.reshape(-1).long(), grad_output.reshape(m * nsample, c))


# ---- src/interpolation/interpolation_cuda_kernel.cu (output[n, c] += input[idx[n, i], c] * weight[n, i])
def interpolation_forward_cuda(n, c, k, input, idx, weight, output):
    output.add_((input[idx.reshape(-1).long()].view(n, k, c) * weight.view(n, k, 1)).sum(1))


def interpolation_backward_cuda(n, c, k, grad_output, idx, weight, grad_input):
    grad_input.index_add_(0, idx.reshape(-1).long(), (grad_output.view(n, 1, c) * weight.view(n, k, 1)).reshape(n * k, c))


# ---- src/subtraction/subtraction_cuda_kernel.cu (output[n, s, c] = input1[n, c] - input2[idx[n, s], c])
def subtraction_forward_cuda(n, nsample, c, input1, input2, idx, output):
    output.copy_(input1.view(n, 1, c) - input2[idx.reshape(-1).long()].view(n, nsample, c))


def subtraction_backward_cuda(n, nsample, c, idx, grad_output, grad_input1, grad_input2):
    grad_input1.add_(grad_output.sum(1))
    grad_input2.index_add_(0, idx.reshape(-1).long(), -grad_output.reshape(n * nsample, c))


# ---- src/aggregation/aggregation_cuda_kernel.cu (output[n, c] += (input[idx[n, s], c] + position[n, s, c]) * weight[n, s, c % w_c])
def _wexp(weight, c, w_c):
    return weight[:, :, torch.arange(c) % w_c]


def aggregation_forward_cuda(n, nsample, c, w_c, input, position, weight, idx, output):
    g = input[idx.reshape(-1).long()].view(n, nsample, c)
    output.add_(((g + position) * _wexp(weight, c, w_c)).sum(1))


def aggregation_backward_cuda(n, nsample, c, w_c, input, position, weight, idx, grad_output, grad_input, grad_position, grad_weight):
    w = _wexp(weight, c, w_c)
    go = grad_output.view(n, 1, c)
    grad_input.index_add_(0, idx.reshape(-1).long(), (go * w).reshape(n * nsample, c))
    grad_position.copy_(go * w)
    g = input[idx.reshape(-1).long()].view(n, nsample, c)
    gw = (go * (g + position)).reshape(n * nsample, c)
    grad_weight.view(n * nsample, w_c).index_add_(1, torch.arange(c) % w_c, gw)


# ---- src/attention/attention_cuda_kernel.cu
def attention_relation_step_forward_cuda(m, g, c, query, key, weight, index_target, index_refer, output):
    output.add_((query[index_target.long()] * key[index_refer.long()] * weight.view(1, 1, c)).sum(-1))


def attention_relation_step_backward_cuda(m, g, c, query, grad_query, key, grad_key, weight, grad_weight, index_target, index_refer,
                                          grad_output):
    q, k, go = query[index_target.long()], key[index_refer.long()], grad_output.view(m, g, 1)
    grad_query.index_add_(0, index_target.long(), go * k * weight.view(1, 1, c))
    grad_key.index_add_(0, index_refer.long(), go * q * weight.view(1, 1, c))
    grad_weight.add_((go * k * q).sum((0, 1)))


def attention_fusion_step_forward_cuda(m, g, c, weight, value, index_target, index_refer, output):
    output.index_add_(0, index_target.long(), weight.view(m, g, 1) * value[index_refer.long()])


def attention_fusion_step_backward_cuda(m, g, c, weight, grad_weight, value, grad_value, index_target, index_refer, grad_output):
    go = grad_output[index_target.long()]
    grad_weight.add_((go * value[index_refer.long()]).sum(-1))
    grad_value.index_add_(0, index_refer.long(), go * weight.view(m, g, 1))


_NAMES = [n for n in dir() if n.endswith("_cuda")]


def load_reference_package(name: str = "pointops_reference"):
    """The reference's libs/pointops/functions package (installed as `pointops`, setup.py:23-24) executed on the stand-ins above.
    While it is imported the names `pointops` / `pointops._C` in sys.modules point at it (functions/utils.py imports from `pointops`);
    they are restored afterwards and the package is returned (also cached under `name`)."""
    if name in sys.modules:
        return sys.modules[name]
    sys.dont_write_bytecode = True
    fdir = os.path.join(REF, "libs", "pointops", "functions")
    if not os.path.isdir(fdir):
        raise RuntimeError(f"reference not found under {REF}")
    c_mod = types.ModuleType("pointops._C")
    for n in _NAMES:
        setattr(c_mod, n, globals()[n])
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "pointops" or k.startswith("pointops.")}
    for k in saved:
        del sys.modules[k]
    try:
        spec = importlib.util.spec_from_file_location("pointops", os.path.join(fdir, "__init__.py"), submodule_search_locations=[fdir])
        pkg = importlib.util.module_from_spec(spec)
        sys.modules["pointops"] = pkg
        sys.modules["pointops._C"] = c_mod
        pkg._C = c_mod
        spec.loader.exec_module(pkg)
    finally:
        for k in [k for k in sys.modules if k == "pointops" or k.startswith("pointops.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    sys.modules[name] = pkg
    return pkg
