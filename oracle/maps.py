"""TEST INFRASTRUCTURE (oracle).  Integer index maps of the PTv3 path, numpy restatements.

  pad_maps      <- SerializedAttention.get_padding_and_inverse
                   pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:114-170
  pooling_maps  <- SerializedPooling.forward index arithmetic, same file :372-412
  offset2batch / batch2offset <- pointcept/models/utils/misc.py:12-34
Pinned to the reference's own code via tests/golden/padmaps_*.npz and pooling_*.npz.
"""
from __future__ import annotations

import numpy as np


def offset2bincount(offset):
    offset = np.asarray(offset, dtype=np.int64)
    return np.diff(offset, prepend=0)  # misc.py:12-16


def offset2batch(offset):
    bc = offset2bincount(offset)
    return np.repeat(np.arange(len(bc), dtype=np.int64), bc)  # misc.py:24-29


def batch2offset(batch):
    return np.cumsum(np.bincount(np.asarray(batch, dtype=np.int64))).astype(np.int64)  # misc.py:32-34


def pad_maps(offset, patch_size: int):
    """Transcription of ptv3m1:114-170 (python loop over scenes included)."""
    offset = np.asarray(offset, dtype=np.int64)
    K = int(patch_size)
    bincount = offset2bincount(offset)
    bincount_pad = (bincount + K - 1) // K * K                                   # :126-133
    mask_pad = bincount > K                                                      # :135
    bincount_pad = (~mask_pad) * bincount + mask_pad * bincount_pad              # :136
    _offset = np.concatenate([[0], offset])                                      # :137
    _offset_pad = np.concatenate([[0], np.cumsum(bincount_pad)])                 # :138
    pad = np.arange(_offset_pad[-1], dtype=np.int64)                             # :139
    unpad = np.arange(_offset[-1], dtype=np.int64)                               # :140
    cu = []
    for i in range(len(offset)):                                                 # :142
        unpad[_offset[i]:_offset[i + 1]] += _offset_pad[i] - _offset[i]          # :143
        if bincount[i] != bincount_pad[i]:                                       # :144
            r = bincount[i] % K
            pad[_offset_pad[i + 1] - K + r:_offset_pad[i + 1]] = pad[
                _offset_pad[i + 1] - 2 * K + r:_offset_pad[i + 1] - K]           # :145-154
        pad[_offset_pad[i]:_offset_pad[i + 1]] -= _offset_pad[i] - _offset[i]    # :155
        cu.append(np.arange(_offset_pad[i], _offset_pad[i + 1], K, dtype=np.int32))  # :156-164
    cu_seqlens = np.concatenate(cu + [np.asarray([_offset_pad[-1]], dtype=np.int32)]).astype(np.int32)  # :167-169
    return pad, unpad, cu_seqlens


def dup_map(pad: np.ndarray, unpad: np.ndarray) -> np.ndarray:
    """Engine-side extra map: dup[rank] = the second padded slot holding `rank`, or -1.
    Derived from (pad, unpad) by definition, independent of the kernel's closed form."""
    dup = np.full(unpad.shape[0], -1, dtype=np.int64)
    slots = np.arange(pad.shape[0], dtype=np.int64)
    secondary = unpad[pad] != slots
    dup[pad[secondary]] = slots[secondary]
    return dup


def pooling_maps(code: np.ndarray, stride: int, serialized_depth: int):
    """ptv3m1:372-406 without the shuffle: returns dict(cluster, counts, idx_ptr, indices, head,
    code, order, inverse, pooling_depth).  `indices` = stable argsort(cluster) (the reference's
    torch.sort is unstable; any member order inside a cluster is equivalent for max/mean)."""
    import math

    pooling_depth = (math.ceil(stride) - 1).bit_length()                          # :372
    if pooling_depth > serialized_depth:
        pooling_depth = 0                                                         # :373-374
    c = code >> (pooling_depth * 3)                                               # :383
    _, cluster, counts = np.unique(c[0], return_inverse=True, return_counts=True)  # :384-390
    indices = np.argsort(cluster, kind="stable")                                  # :392
    idx_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)           # :394
    head = indices[idx_ptr[:-1]]                                                  # :396
    code_c = c[:, head]                                                           # :398
    order = np.argsort(code_c, axis=1, kind="stable")                             # :399
    inverse = np.empty_like(order)
    ar = np.arange(code_c.shape[1], dtype=np.int64)
    for r in range(code_c.shape[0]):
        inverse[r, order[r]] = ar                                                 # :400-406
    return dict(cluster=cluster.astype(np.int64), counts=counts.astype(np.int64), idx_ptr=idx_ptr,
                indices=indices.astype(np.int64), head=head.astype(np.int64), code=code_c, order=order,
                inverse=inverse, pooling_depth=pooling_depth)
