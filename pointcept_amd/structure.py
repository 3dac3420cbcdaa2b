"""Point structure of the engine: the reference's `Point` dict protocol
(pointcept/models/utils/structure.py:20-148) with serialization() and sparsify() running on
libptcore.so kernels instead of ~200 elementwise launches + 4 torch.argsort calls.

Inside a Pointcept checkout (after pointcept_amd.compat.install()) `Point` subclasses the
reference's own class, so `isinstance(point, pointcept...Point)` checks in
DefaultSegmentorV2 (pointcept/models/default.py:68) keep working; standalone it subclasses a small
attribute dict with addict's nested-dict semantics.
"""
from __future__ import annotations

import torch

from . import functional as PF
from . import ops
from . import spconv_api as spconv


class AttrDict(dict):
    """attribute-access dict; like addict.Dict the constructor converts nested dicts into copies
    (so `pooling_parent` stored in a child Point is a shallow copy of the parent Point)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for arg in args:
            if not arg:
                continue
            items = arg.items() if isinstance(arg, dict) else arg
            for k, v in items:
                self[k] = self._hook(v)
        for k, v in kwargs.items():
            self[k] = self._hook(v)

    @classmethod
    def _hook(cls, item):
        if isinstance(item, dict):
            return cls(item)
        if isinstance(item, (list, tuple)):
            return type(item)(cls._hook(e) for e in item)
        return item

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


def _reference_point_base():
    try:  # inside a Pointcept checkout the engine's Point IS-A reference Point
        from pointcept.models.utils.structure import Point as RefPoint  # type: ignore

        return RefPoint
    except Exception:
        return None


_RefPoint = _reference_point_base()


@torch.no_grad()
def offset2bincount(offset):
    return torch.diff(offset, prepend=torch.zeros(1, device=offset.device, dtype=offset.dtype))


@torch.no_grad()
def offset2batch(offset, n=None):
    """pointcept/models/utils/misc.py offset2batch.  `n` = the number of points (offset[-1]) when the caller already knows it on the host
    (the row count of coord / feat): repeat_interleave has to fetch its output size from the device -- a host sync at the very top of the
    step plus an 88 us kernel at 819 200 points (profiles/r05_r_bench_kernel_stats.csv, compute_cuda_kernel) -- whereas "how many scene
    ends are <= i" is a search of B boundaries per point with nothing to wait for."""
    if n is not None:
        return torch.bucketize(torch.arange(int(n), device=offset.device), offset, right=True)
    bincount = offset2bincount(offset)
    return torch.arange(len(bincount), device=offset.device, dtype=torch.long).repeat_interleave(bincount)


@torch.no_grad()
def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).long()


class Point(_RefPoint if _RefPoint is not None else AttrDict):
    """Keys follow the reference: coord, grid_coord, feat, offset, batch, serialized_{depth,code,order,
    inverse}, sparse_shape, sparse_conv_feat, pooling_parent, pooling_inverse.  Engine-side caches
    use keys prefixed `_ptc_` (host copy of offset, coordinate maxima)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if "batch" not in self.keys() and "offset" in self.keys():
            rows = next((int(self[k].shape[0]) for k in ("coord", "grid_coord", "feat") if k in self.keys() and isinstance(self[k], torch.Tensor)), None)
            self["batch"] = offset2batch(self.offset, rows)
        elif "offset" not in self.keys() and "batch" in self.keys():
            self["offset"] = batch2offset(self.batch)

    # -- host-side facts, fetched with ONE device sync per Point -----------------------------------
    def _host_facts(self):
        if "_ptc_coord_max" not in self.keys():
            if "grid_coord" not in self.keys():
                assert {"grid_size", "coord"}.issubset(self.keys())
                self["grid_coord"] = torch.div(self.coord - self.coord.min(0)[0], self.grid_size,
                                               rounding_mode="trunc").int()  # structure.py:68-70
            cmax = ops.coord_max(self.grid_coord)
            # duplicate voxel coordinates (Mix3D batches) change the adjoint of the sparse convolutions
            # (functional._SparseConv): count them here so that the answer rides in the same host sync
            n = self.grid_coord.shape[0]
            if n > 0 and "batch" in self.keys():
                idx = torch.cat([self.batch.unsqueeze(-1).int(), self.grid_coord.int()], dim=1).contiguous()
                rep = ops.rulebook_subm(idx, 1, ops.HashTable(idx))[0]
                n_dup = (rep != torch.arange(n, device=rep.device, dtype=rep.dtype)).sum().reshape(1)
            else:
                n_dup = cmax.new_zeros(1)
            packed = torch.cat([cmax, n_dup.to(torch.int64), self.offset.to(torch.int64)])
            host = packed.tolist()  # the single host sync (reference: structure.py:74,138,145 + ptv3m1:142-164)
            ops.check_coord_range(host[:3], self.offset.numel())
            # the sync-free offset2batch(offset, n) trusts the row count it is given; the reference's repeat_interleave
            # would have produced offset[-1] entries and failed downstream.  Checked here, where offset reaches the host anyway.
            for k in ("coord", "grid_coord", "feat", "batch"):
                if k in self.keys() and isinstance(self[k], torch.Tensor) and self.offset.numel() and int(self[k].shape[0]) != int(host[-1]):
                    raise ValueError(f"Point: `{k}` has {int(self[k].shape[0])} rows but offset[-1] = {int(host[-1])}")
            self["_ptc_coord_max"] = host[:3]
            self["_ptc_n_dup"] = int(host[3])
            self["_ptc_offset_host"] = host[4:]
        return self["_ptc_coord_max"], self["_ptc_offset_host"]

    def serialization(self, order="z", depth=None, shuffle_orders=False):
        """structure.py:53-110: codes for every order in one kernel, one batched radix sort."""
        order = [order] if isinstance(order, str) else list(order)
        self["order"] = order
        assert "batch" in self.keys()
        coord_max, offset_host = self._host_facts()
        if depth is None:
            depth = int(max(coord_max) + 1).bit_length()  # structure.py:74
        self["serialized_depth"] = depth
        nb = len(offset_host).bit_length()
        assert depth * 3 + nb <= 63  # structure.py:77
        assert depth <= 16            # structure.py:82
        # shuffle_orders (structure.py:102-106: code / order / inverse rows permuted by a CPU randperm) without its three row gathers of
        # [k, N] int64 tensors: every row is a function of its own order name alone, so encoding and sorting the names in permuted order
        # IS the permuted result (same generator, same draw)
        enc_order = order
        if shuffle_orders:
            perm = torch.randperm(len(order))  # CPU default generator, as structure.py:103
            enc_order = [order[int(p)] for p in perm]
        code = ops.serialize_encode(self.grid_coord, self.batch, depth, enc_order)
        sorted_order, inverse = ops.sort_keys(code, 0, depth * 3 + nb)
        self["serialized_code"] = code
        self["serialized_order"] = sorted_order
        self["serialized_inverse"] = inverse

    # -- physical row order ------------------------------------------------------------------------
    _NOT_PER_POINT = ("offset", "serialized_code", "serialized_order", "serialized_inverse")

    def physically_sorted(self):
        """Engine-internal working copy whose ROWS are stored in the order of the first serialization
        curve (row r holds original point serialized_order[0][r]).  Dataloader order is spatially
        random, so every neighbour gather of the stage-0 convolutions and the serialization gathers
        of attention would otherwise miss L2 (4 MiB per XCD); pooled stages are born curve-sorted
        (clusters are numbered by ascending code, ptv3m1:385-390).  All index maps are re-expressed
        in the new row numbering, so every operator of the model is unchanged; `restore_order`
        undoes the permutation on the way out.  Requires serialization() to have run."""
        pi, inv0 = self.serialized_order[0], self.serialized_inverse[0]
        n = pi.numel()
        d = {}
        for key, val in self.items():
            if key in self._NOT_PER_POINT or key.startswith("_ptc_unsort"):
                continue
            if isinstance(val, torch.Tensor) and val.dim() >= 1 and val.shape[0] == n and key != "feat":
                d[key] = val[pi]
            else:
                d[key] = val
        d["offset"] = self.offset  # codes carry the batch index in their top bits: scenes stay contiguous
        d["feat"] = PF.gather_rows(self.feat, pi, inv0) if self.feat.is_floating_point() and self.feat.dim() == 2 \
            else self.feat[pi]
        d["serialized_code"] = self.serialized_code[:, pi]
        d["serialized_order"] = inv0[self.serialized_order]      # sorted rank -> new row
        d["serialized_inverse"] = self.serialized_inverse[:, pi]  # new row -> sorted rank
        w = Point(d)
        w["_ptc_unsort"] = (pi, inv0)
        return w

    def restore_order(self, template):
        """Inverse of physically_sorted: `template` (the caller-order Point the working copy was made
        from) with this point's features brought back to caller order; order-independent caches
        (pad / unpad / cu_seqlens_key) are carried over."""
        pi, inv0 = self["_ptc_unsort"]
        out = Point(template)
        out["feat"] = PF.gather_rows(self.feat, inv0, pi)
        for key in ("pad", "unpad", "cu_seqlens_key", "sparse_shape"):
            if key in self.keys():
                out[key] = self[key]
        return out

    def sparsify(self, pad=96):
        """structure.py:112-148"""
        assert {"feat", "batch"}.issubset(self.keys())
        coord_max, offset_host = self._host_facts()
        if "sparse_shape" in self.keys():
            sparse_shape = self.sparse_shape
        else:
            sparse_shape = [int(m) + pad for m in coord_max]  # torch.add(max(grid_coord), pad).tolist()
        indices = torch.cat([self.batch.unsqueeze(-1).int(), self.grid_coord.int()], dim=1).contiguous()
        self["sparse_shape"] = sparse_shape
        self["sparse_conv_feat"] = spconv.SparseConvTensor(
            features=self.feat, indices=indices, spatial_shape=sparse_shape, batch_size=len(offset_host))
        spconv.mark_duplicates(self["sparse_conv_feat"], self.get("_ptc_n_dup", 0) > 0)
