#!/usr/bin/env python
"""Halo rows per 128-row block at every level of SpUNet's voxel pyramid, on the engine's own tables (bench batch: 8 x 100000 voxels, level 0
in Hilbert order, the coarse levels numbered by ptc_rulebook_down): how many blocks exceed conv8's 352-row / conv7's 416-row LDS image."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcept_amd import ops, synthetic  # noqa: E402

DEV = torch.device("cuda:0")
b = synthetic.to_torch(synthetic.indoor_batch(8, 100000), DEV)
gc, off = b["grid_coord"], b["offset"]
bt = torch.repeat_interleave(torch.arange(off.numel(), device=DEV), torch.diff(off, prepend=off.new_zeros(1)))
code = ops.serialize_encode(gc.long(), bt, 16, ("hilbert",))
order, _ = ops.sort_keys(code, 0, 48 + 3)
ind = torch.cat([bt[:, None].int(), gc.int()], 1)[order[0]].contiguous()
for lvl in range(5):
    nbr = ops.rulebook_subm(ind, 3, ops.HashTable(ind))
    blk = ops.BlockTables(nbr)
    h = blk.hcnt.float()
    print(f"level {lvl}: n={ind.shape[0]} blocks={h.numel()} halo mean {h[h >= 0].mean():.0f} max {int(h.max())} marked-overflow {(h < 0).sum().item()} "
          f"beyond352 {((h > 352) | (h < 0)).sum().item()} beyond416 {((h > 416) | (h < 0)).sum().item()}")
    if lvl == 4:
        break
    cb = int(ind[:, 1:].max().item() >> 1).bit_length()
    ind = ops.rulebook_down(ind, cb, 3)[0]
