#!/usr/bin/env python
"""CPU statistics that size the block-staged convolution (csrc/conv7.h, csrc/blocks.hip) on the synthetic indoor scenes:
  * halo: distinct input rows named by a block of 128 consecutive output rows (rows in curve order);
  * tap occupancy: which fraction of the (32-row tile, tap) pairs of a 3^3 table has at least one neighbour -- the MFMA work
    a kernel that skips empty (tile, tap) pairs would keep.
Uses the oracle rulebook (test infrastructure): this is a design tool, not a product path.

    python tools/halo_stats.py [--scenes 2] [--points 102400] [--order z|hilbert]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ops as oops, sfc  # noqa: E402
from pointcept_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=2)
ap.add_argument("--points", type=int, default=102400)
ap.add_argument("--order", default="z")
ap.add_argument("--stride", type=int, default=1, help="grid stride of the level (2 = after one pooling)")
ap.add_argument("--outdoor", action="store_true", help="LiDAR sweeps of BASELINE configs[4] instead of the indoor rooms")
a = ap.parse_args()
halo, occ_tile, occ_half, occ_blk, nbrs = [], [], [], [], []
for s in range(a.scenes):
    sc = synthetic.outdoor_scene(5000 + s, azimuth_steps=3300) if a.outdoor else synthetic.indoor_scene(s, a.points)
    gc = np.unique(sc["grid_coord"] // a.stride, axis=0)
    code = sfc.encode_c(gc, None, 16, [a.order])[0]
    gc = gc[np.argsort(code, kind="stable")]
    idx = np.concatenate([np.zeros((gc.shape[0], 1), np.int64), gc], 1)
    nbr = oops.subm_rulebook(idx, 3)          # [27, n]
    n = nbr.shape[1]
    nbrs.append((nbr >= 0).sum() / n)
    for b in range(0, n - 127, 128):
        blk = nbr[:, b:b + 128]
        halo.append(np.unique(blk[blk >= 0]).size)
    nt = n // 32
    t = (nbr[:, :nt * 32] >= 0).reshape(27, nt, 32)
    occ_tile.append(t.any(2).mean())
    occ_blk.append((nbr[:, :n // 128 * 128] >= 0).reshape(27, n // 128, 128).any(2).mean())
    occ_half.append((nbr[:, :n // 16 * 16] >= 0).reshape(27, n // 16, 16).any(2).mean())
halo = np.asarray(halo)
print(f"{a.scenes} scenes x <= {a.points} voxels (grid stride {a.stride}), rows in {a.order}-order: {np.mean(nbrs):.2f} neighbours per voxel of 27")
print(f"halo of a 128-row block: mean {halo.mean():.0f}, p50 {np.percentile(halo, 50):.0f}, p99 {np.percentile(halo, 99):.0f}, max {halo.max()}")
print(f"(32-row tile, tap) pairs with >= 1 neighbour: {np.mean(occ_tile):.3f}; (16-row tile, tap): {np.mean(occ_half):.3f}; (128-row block, tap): {np.mean(occ_blk):.3f}; dense fraction {np.mean(nbrs) / 27:.3f}")
