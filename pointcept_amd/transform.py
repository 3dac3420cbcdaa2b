"""Device-side front end of the hot path: GridSample, SphereCrop and the Mix3D collate of the
reference's dataloader, for point clouds that already live on the GPU.

GridSample: the voxelisation transform of pointcept/datasets/transform.py:839-1011 for point clouds
that already live on the GPU (SURVEY 8(f) rank 1 -- the step immediately before the hot path; on the reference it runs
in numpy inside 24 dataloader workers).  Same constructor arguments, same dict protocol (`index_valid_keys`,
`inverse`, `grid_coord`, `min_coord`, `displacement`, train / test modes), torch tensors instead of numpy arrays.

    voxel of a point  : floor(coord / grid_size) in float64, min-shifted           ptc_voxel_keys   (voxelize.hip)
    key               : FNV-64 over the three coordinates (hash_type="fnv")         ptc_voxel_keys
    argsort(key)      : stable LSD radix sort over 64 bits                          ptc_sort_keys    (scan_sort.hip)
    unique / inverse / count : flags + scan + fill over the sorted keys             ptc_pool_maps_*  (maps.hip)
    representative    : idx_sort[start + r % count], r random per voxel             torch indexing

Differences, all documented: numpy's argsort is unstable, so WHICH point of a voxel sits at a given rank is
implementation-defined there and "ascending index" here (the voxel set, `inverse`, `count` and the key order are
identical); the random offsets come from torch's generator (`rand=` injects them for tests) and are drawn from
[0, 2^31) instead of [0, count.max()) so that no host sync is needed; hash_type="ravel" and `frame_pcd_offset` are
not implemented (raise).  There is no CPU path.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from ._lib import PtcoreError

_DEFAULT_KEYS = ["coord", "color", "normal", "superpoint", "strength", "segment", "instance"]   # transform.py:27-35


def index_operator(data_dict, index, duplicate=False):
    """transform.py:23-52: apply `index` to every key listed in data_dict["index_valid_keys"]."""
    if "index_valid_keys" not in data_dict:
        data_dict["index_valid_keys"] = list(_DEFAULT_KEYS)
    if not duplicate:
        for key in data_dict["index_valid_keys"]:
            if key in data_dict:
                data_dict[key] = data_dict[key][index]
        return data_dict
    out = dict()
    for key in data_dict.keys():
        if key in data_dict["index_valid_keys"]:
            out[key] = data_dict[key][index]
        elif key == "index_valid_keys":
            out[key] = list(data_dict[key])      # each part extends its own copy
        else:
            out[key] = data_dict[key]
    return out


class GridSample:
    def __init__(self, grid_size=0.05, hash_type="fnv", mode="train", return_inverse=False, return_grid_coord=False,
                 return_min_coord=False, return_displacement=False, project_displacement=False):
        if hash_type != "fnv":
            raise PtcoreError("GridSample: only hash_type='fnv' (the default of every reference config) is implemented")
        assert mode in ["train", "test"]
        self.grid_size = grid_size
        self.mode = mode
        self.return_inverse = return_inverse
        self.return_grid_coord = return_grid_coord
        self.return_min_coord = return_min_coord
        self.return_displacement = return_displacement
        self.project_displacement = project_displacement

    @torch.no_grad()
    def voxels(self, coord: torch.Tensor):
        """-> dict(grid_coord [N,3] i64, min_coord [3] i64, key [N] i64, idx_sort [N], inverse [N] (voxel id of every
        point, voxels numbered by ascending key as np.unique does), idx_ptr [V+1] (CSR of the voxels over idx_sort))."""
        grid_coord, min_coord, key = ops.voxel_keys(coord, self.grid_size)
        idx_sort, _ = ops.sort_keys(key, 0, 64, want_inverse=False)
        inverse, idx_ptr, _ = ops.pool_maps(key, idx_sort, 0)     # one host sync: the number of voxels sizes the outputs
        return dict(grid_coord=grid_coord, min_coord=min_coord, key=key, idx_sort=idx_sort, inverse=inverse, idx_ptr=idx_ptr)

    def _extras(self, out, data_dict, v, index):
        if self.return_inverse:
            out["inverse"] = v["inverse"]
        if self.return_grid_coord:
            out["grid_coord"] = v["grid_coord"][index]
            if "grid_coord" not in out["index_valid_keys"]:
                out["index_valid_keys"] = list(out["index_valid_keys"]) + ["grid_coord"]
        if self.return_min_coord:
            out["min_coord"] = (v["min_coord"].double() * self.grid_size).reshape(1, 3)
        if self.return_displacement:
            scaled = data_dict["coord"].double() / self.grid_size - v["min_coord"].double()
            disp = scaled - v["grid_coord"].double() - 0.5           # [0, 1] -> [-0.5, 0.5] from the voxel centre
            if self.project_displacement:
                disp = (disp * data_dict["normal"].double()).sum(-1, keepdim=True)
            out["displacement"] = disp[index]
            if "displacement" not in out["index_valid_keys"]:
                out["index_valid_keys"] = list(out["index_valid_keys"]) + ["displacement"]
        return out

    @torch.no_grad()
    def __call__(self, data_dict, rand: Optional[torch.Tensor] = None):
        assert "coord" in data_dict.keys()
        if "frame_pcd_offset" in data_dict:
            raise PtcoreError("GridSample: frame_pcd_offset is not implemented")
        src = dict(data_dict)                      # originals, for the displacement (the reference computes it before indexing)
        if data_dict["coord"].shape[0] == 0:
            raise PtcoreError("GridSample: empty point cloud")
        v = self.voxels(data_dict["coord"])
        idx_sort, idx_ptr = v["idx_sort"], v["idx_ptr"]
        start, count = idx_ptr[:-1], idx_ptr[1:] - idx_ptr[:-1]
        if self.mode == "train":
            if rand is None:
                rand = torch.randint(0, 2 ** 31 - 1, (count.numel(),), device=count.device)
            idx_unique = idx_sort[start + rand.to(count.dtype) % count]           # transform.py:877-882
            if "sampled_index" in data_dict:                                       # transform.py:883-891
                idx_unique = torch.unique(torch.cat([idx_unique, data_dict["sampled_index"].to(idx_unique.dtype)]))
                mask = torch.zeros(data_dict["segment"].shape[0], dtype=torch.bool, device=idx_unique.device)
                mask[data_dict["sampled_index"]] = True
                data_dict["sampled_index"] = torch.where(mask[idx_unique])[0]
            out = index_operator(data_dict, idx_unique)
            return self._extras(out, src, v, idx_unique)
        parts = []                                                                  # transform.py:916-949
        for i in range(int(count.max().item()) if count.numel() else 0):
            idx_part = idx_sort[start + i % count]
            part = index_operator(data_dict, idx_part, duplicate=True)
            part["index"] = idx_part
            parts.append(self._extras(part, src, v, idx_part))
        return parts


class SphereCrop:
    """pointcept/datasets/transform.py:1014-1057 on device tensors: keep the `point_max` points nearest a centre.
    The reference sorts ALL squared distances (np.argsort) and keeps the first point_max -- the output is in ascending
    distance order; here the fp32 distance bits are radix-sorted (non-negative floats order like their bit patterns), ties
    in ascending index order (numpy's argsort is unstable there).  mode "random" draws the centre from torch's generator
    (`center_index=` injects it); "center" takes point N // 2 (:1033); "given" follows :1034-1047.  mode "all" is
    accepted by the reference's assert but never implemented there (:1048-1049); same here."""

    def __init__(self, point_max=80000, sample_rate=None, mode="random"):
        self.point_max, self.sample_rate = point_max, sample_rate
        assert mode in ["random", "center", "all", "given"]
        self.mode = mode

    @torch.no_grad()
    def __call__(self, data_dict, center_index: Optional[int] = None):
        assert "coord" in data_dict.keys()
        coord = data_dict["coord"]
        n = coord.shape[0]
        point_max = int(self.sample_rate * n) if self.sample_rate is not None else self.point_max
        if n <= point_max:
            return data_dict
        if self.mode == "random":
            ci = int(torch.randint(n, (1,)).item()) if center_index is None else int(center_index)
            center = coord[ci]
        elif self.mode == "center":
            center = coord[n // 2]
        elif self.mode == "given":
            corr = data_dict["correspondence"].reshape(data_dict["correspondence"].shape[0], -1)
            given = (corr != -1).all(dim=1)
            if int(given.sum()) == 0:
                ci = int(torch.randint(n, (1,)).item()) if center_index is None else int(center_index)
                center = coord[ci]
            else:
                center = coord[given].mean(dim=0)
        else:
            raise NotImplementedError
        d2 = ((coord - center) ** 2).sum(1)
        if d2.dtype == torch.float64:
            keys, bits = d2.view(torch.int64), 63
        else:
            keys, bits = d2.float().view(torch.int32).to(torch.int64), 31
        order, _ = ops.sort_keys(keys[None].contiguous(), 0, bits, want_inverse=False)
        return index_operator(data_dict, order[0][:point_max])


def collate_fn(batch):
    """pointcept/datasets/utils.py:19-73 for device tensors: concatenate along points; keys containing "offset" become
    cumulative offsets of the concatenation (:52-60); samples that are sequences of tensors get their lengths appended and a
    cumulative int offset as the last element (:35-40); anything else goes to torch's default_collate (:72).  (The image /
    correspondence branches of the multi-modal pre-training datasets, :44-45,61-71, are outside the PTv3 / SpUNet path and raise.)"""
    from collections.abc import Mapping, Sequence

    if not isinstance(batch, Sequence):
        raise TypeError(f"{type(batch)} is not supported.")
    if isinstance(batch[0], torch.Tensor):
        return torch.cat(list(batch))
    if isinstance(batch[0], str):
        return list(batch)
    if isinstance(batch[0], (int, float)):
        return torch.tensor(list(batch))
    if isinstance(batch[0], list) and not any(isinstance(e, torch.Tensor) for e in batch[0]):
        return torch.cat([torch.tensor(d) for d in batch])
    if isinstance(batch[0], Sequence):                               # utils.py:35-40: (coord, feat, label, ...) tuples / lists of tensors
        rows = [list(d) + [torch.tensor([d[0].shape[0]], device=d[0].device if isinstance(d[0], torch.Tensor) else None)] for d in batch]
        out = [collate_fn(list(samples)) for samples in zip(*rows)]
        out[-1] = torch.cumsum(out[-1], dim=0).int()
        return out
    if isinstance(batch[0], Mapping):
        if "img_num" in batch[0] or any("correspondence" in k for k in batch[0]):
            raise PtcoreError("collate_fn: the image / correspondence branches are not part of this engine")
        out = {}
        for key in batch[0]:
            if "offset" in key:
                parts = [torch.diff(d[key], prepend=d[key].new_zeros(1)) for d in batch]
                out[key] = torch.cumsum(torch.cat(parts), dim=0)
            else:
                out[key] = collate_fn([d[key] for d in batch])
        return out
    from torch.utils.data.dataloader import default_collate

    return default_collate(batch)                                    # utils.py:72


def point_collate_fn(batch, mix_prob=0, mix: Optional[bool] = None):
    """pointcept/datasets/utils.py:208-258: collate, then with probability mix_prob Mix3D -- every second scene boundary
    is dropped (:231-236), so pairs of scenes become ONE batch item whose grid coordinates overlap (duplicate voxels are
    legal input of the engine); instance ids of the second scene of a pair are shifted (:222-230); when the
    sample carries `grid_size`, grid_coord is recomputed from the merged coordinates per batch item (:245-250).
    `mix=` injects the coin flip for tests (the reference uses python's `random`)."""
    import random

    assert isinstance(batch[0], dict)
    batch = collate_fn(batch)
    do_mix = (random.random() < mix_prob) if mix is None else bool(mix)
    if not do_mix:
        return batch
    if "instance" in batch.keys():
        offset = batch["offset"].tolist()
        start, num_instance = 0, 0
        for i in range(len(offset)):
            seg = batch["instance"][start:offset[i]]
            if i % 2 == 0:
                num_instance = int(seg.max()) if seg.numel() else 0
            else:
                seg += num_instance * (seg != -1)
            start = offset[i]
    for key in [k for k in batch.keys() if "offset" in k]:
        batch[key] = torch.cat([batch[key][1:-1:2], batch[key][-1].unsqueeze(0)], dim=0)
    if "grid_coord" in batch and "grid_size" in batch:
        off = batch["offset"]
        counts = torch.diff(off, prepend=off.new_zeros(1))
        bidx = torch.repeat_interleave(torch.arange(off.numel(), device=off.device), counts)
        gs = batch["grid_size"][0] if torch.is_tensor(batch["grid_size"]) and batch["grid_size"].dim() > 0 else batch["grid_size"]
        gc = torch.floor(batch["coord"] / gs).to(torch.int64)
        mn = gc.new_full((off.numel(), 3), torch.iinfo(torch.int64).max).scatter_reduce(0, bidx[:, None].expand(-1, 3), gc, "amin")
        batch["grid_coord"] = gc - mn[bidx]
    return batch
