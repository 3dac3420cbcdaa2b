"""TEST INFRASTRUCTURE (oracle).  Pure-torch CPU stand-ins for the un-vendored third-party modules
the reference's hot-path files import, built on oracle/ops.py:

  addict.Dict                      (structure.py:20, ptv3m1:415)     -- recalled addict 2.4 semantics
  timm.layers.{DropPath,trunc_normal_}  (ptv3m1:15,315; spconv_unet_v1m1_base.py:17)
  torch_scatter.segment_csr        (ptv3m1:416-421)
  spconv.pytorch.{SparseConvTensor,SubMConv3d,SparseConv3d,SparseInverseConv3d,SparseSequential,
                  SparseModule,Identity,modules.is_spconv_module}
  torch_geometric.utils.scatter    (spconv_unet_v1m1_base.py:15,278; enc_mode only)

They serve two purposes: (1) oracle/ref_import.py seeds sys.modules with them so the reference's
own model files import and run UNMODIFIED on CPU; (2) oracle/ptv3_model.py,
the standalone restatement that travels to the GPU box, is built from the same classes.
"""
from __future__ import annotations

import math
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import ops


# ------------------------------------------------------------------------------------------------
# addict.Dict
# ------------------------------------------------------------------------------------------------
class Dict(dict):
    def __init__(__self, *args, **kwargs):
        object.__setattr__(__self, "__parent", kwargs.pop("__parent", None))
        object.__setattr__(__self, "__key", kwargs.pop("__key", None))
        for arg in args:
            if not arg:
                continue
            elif isinstance(arg, dict):
                for key, val in arg.items():
                    __self[key] = __self._hook(val)
            elif isinstance(arg, tuple) and (not isinstance(arg[0], tuple)):
                __self[arg[0]] = __self._hook(arg[1])
            else:
                for key, val in iter(arg):
                    __self[key] = __self._hook(val)
        for key, val in kwargs.items():
            __self[key] = __self._hook(val)

    def __setattr__(self, name, value):
        if hasattr(self.__class__, name):
            raise AttributeError("'Dict' object attribute '{0}' is read-only".format(name))
        self[name] = value

    def __setitem__(self, name, value):
        super().__setitem__(name, value)
        try:
            p = object.__getattribute__(self, "__parent")
            key = object.__getattribute__(self, "__key")
        except AttributeError:
            p, key = None, None
        if p is not None:
            p[key] = self
            object.__delattr__(self, "__parent")
            object.__delattr__(self, "__key")

    @classmethod
    def _hook(cls, item):
        if isinstance(item, dict):
            return cls(item)
        elif isinstance(item, (list, tuple)):
            return type(item)(cls._hook(elem) for elem in item)
        return item

    def __getattr__(self, item):
        return self.__getitem__(item)

    def __missing__(self, name):
        return self.__class__(__parent=self, __key=name)

    def __delattr__(self, name):
        del self[name]


# ------------------------------------------------------------------------------------------------
# timm.layers
# ------------------------------------------------------------------------------------------------
class DropPath(nn.Module):
    """Stochastic depth on dim 0 -- i.e. PER POINT for [N,C] features."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


# ------------------------------------------------------------------------------------------------
# spconv.pytorch subset
# ------------------------------------------------------------------------------------------------
class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, indice_dict=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = list(spatial_shape)
        self.batch_size = batch_size
        self.indice_dict = {} if indice_dict is None else indice_dict

    def replace_feature(self, feature):
        t = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.indice_dict)
        return t


class SparseModule(nn.Module):
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class Identity(nn.Identity):
    pass


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        from collections import OrderedDict

        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            self.add_module(name, module)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
        self.add_module(name, module)

    def forward(self, input):
        for k, module in self._modules.items():
            if is_spconv_module(module):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input = input.replace_feature(module(input.features))
            else:
                input = module(input)
        return input


class _ConvBase(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, algo=None, **kw):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.stride = stride if isinstance(stride, int) else stride[0]
        self.indice_key = indice_key
        k = self.kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        fan_in = k * k * k * in_channels
        bound = 1.0 / math.sqrt(fan_in)
        nn.init.uniform_(self.weight, -bound * math.sqrt(3.0), bound * math.sqrt(3.0))
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)


class SubMConv3d(_ConvBase):
    def forward(self, x: SparseConvTensor):
        if self.kernel_size == 1:
            out = x.features @ self.weight.reshape(self.out_channels, self.in_channels).t()
            if self.bias is not None:
                out = out + self.bias
            return x.replace_feature(out)
        key = ("subm", self.indice_key, self.kernel_size)
        nbr = x.indice_dict.get(key) if self.indice_key is not None else None
        if nbr is None:
            nbr = ops.subm_rulebook(x.indices.cpu().numpy(), self.kernel_size)
            if self.indice_key is not None:
                x.indice_dict[key] = nbr
        return x.replace_feature(ops.gather_conv(x.features, self.weight, self.bias, nbr))


class SparseConv3d(_ConvBase):
    def forward(self, x: SparseConvTensor):
        assert self.kernel_size == 2 and self.stride == 2, "oracle implements k=2,s=2 only"
        out_indices, out_of_in, nbr_down, nbr_up = ops.down_rulebook(x.indices.cpu().numpy())
        if self.indice_key is not None:
            x.indice_dict[("down", self.indice_key)] = (x.indices, x.spatial_shape, nbr_up)
        out = ops.gather_conv(x.features, self.weight, self.bias, nbr_down)
        shape = [(s + 1) // 2 for s in x.spatial_shape]
        return SparseConvTensor(out, torch.as_tensor(out_indices), shape, x.batch_size, x.indice_dict)


class SparseInverseConv3d(_ConvBase):
    def forward(self, x: SparseConvTensor):
        fine_indices, fine_shape, nbr_up = x.indice_dict[("down", self.indice_key)]
        out = ops.gather_conv(x.features, self.weight, self.bias, nbr_up)
        return SparseConvTensor(out, fine_indices, fine_shape, x.batch_size, x.indice_dict)


# ------------------------------------------------------------------------------------------------
# torch_scatter / torch_geometric
# ------------------------------------------------------------------------------------------------
def segment_csr(src, indptr, out=None, reduce="sum"):
    return ops.segment_csr(src, indptr, reduce)


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = src.new_zeros((n,) + tuple(src.shape[1:])).index_add(0, index, src)
    if reduce == "mean":
        cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
        out = out / cnt.reshape((-1,) + (1,) * (src.ndim - 1))
    return out


def voxel_grid(pos, size, batch=None, start=None, end=None):
    """torch_geometric.nn.pool.voxel_grid (third party, absent here; point_transformer_v2m2_base.py:15,254-256) = torch_cluster's
    grid_cluster on [pos | batch]: cell = trunc((pos - start) / size) per dimension, linearised with x fastest and the batch index
    slowest (number of cells per dimension from `end`, default the data maximum).  Only the ORDER of the ids matters to the caller
    (it runs torch.unique on them and needs scenes to stay contiguous)."""
    pos = pos.unsqueeze(-1) if pos.dim() == 1 else pos
    dim = pos.shape[1]
    if batch is None:
        batch = pos.new_zeros(pos.shape[0], dtype=torch.long)
    p = torch.cat([pos, batch.view(-1, 1).to(pos.dtype)], dim=-1)
    sz = torch.tensor(([float(size)] * dim if not isinstance(size, (list, tuple)) else list(size)) + [1.0], dtype=pos.dtype)
    st = torch.zeros(dim + 1, dtype=pos.dtype) if start is None or not isinstance(start, (list, tuple, torch.Tensor)) else torch.cat(
        [torch.as_tensor(start, dtype=pos.dtype).reshape(-1), pos.new_zeros(1)])
    if start is not None and not isinstance(start, (list, tuple, torch.Tensor)):
        st = torch.tensor([float(start)] * dim + [0.0], dtype=pos.dtype)
    elif start is None:
        st = p.min(0).values
    en = p.max(0).values if end is None else torch.cat([torch.as_tensor(end, dtype=pos.dtype).reshape(-1), batch.max().to(pos.dtype).reshape(1)])
    p = p - st.unsqueeze(0)
    num = torch.div(en - st, sz).to(torch.long) + 1
    stride = torch.cat([torch.ones(1, dtype=torch.long), num.cumprod(0)])[: dim + 1]
    return (torch.div(p, sz.unsqueeze(0)).to(torch.long) * stride.unsqueeze(0)).sum(1)


def scatter_min(src, index, dim=0, out=None, dim_size=None):
    """torch_scatter.scatter_min over dim 0 (pointcept/datasets/utils.py:249): (per-group minimum, unused argmin)."""
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    idx = index.reshape((-1,) + (1,) * (src.ndim - 1)).expand_as(src)
    big = torch.iinfo(src.dtype).max if not src.is_floating_point() else float("inf")
    res = src.new_full((n,) + tuple(src.shape[1:]), big).scatter_reduce(0, idx, src, "amin", include_self=True)
    return res, None


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                     **unused):
    """CPU stand-in for flash_attn (ptv3m1:208-214): 16-bit in (bf16 at the PTv3 call sites, fp16 at LitePT's,
    litept_v1.py:235-260), fp32 math, same 16-bit type out.  Lets the reference run its FLASH branch (enable_flash=True)
    unmodified on CPU."""
    assert qkv.dtype in (torch.bfloat16, torch.float16) and dropout_p == 0 and not causal
    scale = qkv.shape[-1] ** -0.5 if softmax_scale is None else softmax_scale
    return ops.attention_varlen(qkv.float(), cu_seqlens.tolist(), scale).to(qkv.dtype)


def install_third_party(mods=None):
    """Seed sys.modules with the stand-ins (idempotent)."""
    m = sys.modules if mods is None else mods

    def mod(name, **attrs):
        t = types.ModuleType(name)
        t.__dict__.update(attrs)
        m[name] = t
        return t

    mod("addict", Dict=Dict)
    timm = mod("timm")
    timm.layers = mod("timm.layers", DropPath=DropPath, trunc_normal_=trunc_normal_)
    mod("torch_scatter", segment_csr=segment_csr, scatter_min=scatter_min)
    mod("flash_attn", flash_attn_varlen_qkvpacked_func=flash_attn_varlen_qkvpacked_func)
    sp_modules = mod("spconv.pytorch.modules", is_spconv_module=is_spconv_module, SparseModule=SparseModule)
    sp = mod("spconv")
    sp.pytorch = mod(
        "spconv.pytorch", SparseConvTensor=SparseConvTensor, SubMConv3d=SubMConv3d, SparseConv3d=SparseConv3d,
        SparseInverseConv3d=SparseInverseConv3d, SparseSequential=SparseSequential, SparseModule=SparseModule,
        Identity=Identity, modules=sp_modules)
    tg = mod("torch_geometric")
    tg.utils = mod("torch_geometric.utils", scatter=scatter)
    tg.nn = mod("torch_geometric.nn")
    tg.nn.pool = mod("torch_geometric.nn.pool", voxel_grid=voxel_grid)
