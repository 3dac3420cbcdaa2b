"""Seeded synthetic voxelised scenes (SURVEY section 8(d) / Appendix B generators) for bench.py and
the tests: there is no network for datasets, so every measurement uses these.

indoor_scene : room 7 x 5 x 2.8 m (floor + 4 walls, 12000 pts/m^2) + 14 boxes, voxelised at 0.02 m,
               sphere-cropped to `point_max` voxels (pointcept/datasets/transform.py:1015-1057),
               feat = colour(3) | normal(3), segment in [0,20) with 5 % ignore (-1).
Layout matches point_collate_fn (pointcept/datasets/utils.py:19-73): coord [N,3] f32,
grid_coord [N,3] i64, feat [N,6] f32, segment [N] i64, offset [B] i64.
"""
from __future__ import annotations

import numpy as np


def _rect(rng, o, u, v, n):
    a = rng.random((n, 1))
    b = rng.random((n, 1))
    return np.asarray(o, float) + a * np.asarray(u, float) + b * np.asarray(v, float)


def _box(rng, c, s, n):
    c = np.array(c, float)
    s = np.array(s, float)
    P, Nrm = [], []
    for ax in range(3):
        for sg in (-1, 1):
            o = c.copy()
            o[ax] += sg * s[ax] / 2
            u = np.zeros(3)
            v = np.zeros(3)
            u[(ax + 1) % 3] = s[(ax + 1) % 3]
            v[(ax + 2) % 3] = s[(ax + 2) % 3]
            P.append(_rect(rng, o - u / 2 - v / 2, u, v, n))
            nn = np.zeros(3)
            nn[ax] = sg
            Nrm.append(np.tile(nn, (n, 1)))
    return np.concatenate(P), np.concatenate(Nrm)


def indoor_scene(seed: int, point_max: int = 102400, grid: float = 0.02, density: int = 12000):
    rng = np.random.default_rng(seed)
    L, W, H = 7.0, 5.0, 2.8
    parts = [
        (_rect(rng, [0, 0, 0], [L, 0, 0], [0, W, 0], int(L * W * density)), [0, 0, 1]),
        (_rect(rng, [0, 0, 0], [L, 0, 0], [0, 0, H], int(L * H * density)), [0, 1, 0]),
        (_rect(rng, [0, W, 0], [L, 0, 0], [0, 0, H], int(L * H * density)), [0, -1, 0]),
        (_rect(rng, [0, 0, 0], [0, W, 0], [0, 0, H], int(W * H * density)), [1, 0, 0]),
        (_rect(rng, [L, 0, 0], [0, W, 0], [0, 0, H], int(W * H * density)), [-1, 0, 0]),
    ]
    P = [p for p, _ in parts]
    Nr = [np.tile(np.asarray(nv, float), (p.shape[0], 1)) for p, nv in parts]
    for _ in range(14):
        s = rng.uniform(0.3, 1.6, 3)
        bp, bn = _box(rng, [rng.uniform(0.8, L - 0.8), rng.uniform(0.8, W - 0.8), s[2] / 2], s, int(density * 1.2))
        P.append(bp)
        Nr.append(bn)
    p = np.concatenate(P)
    nrm = np.concatenate(Nr)
    gc = np.floor(p / grid).astype(np.int64)
    gc -= gc.min(0)
    # one point per voxel (GridSample train mode keeps one representative per voxel)
    key = (gc[:, 0] * 4096 + gc[:, 1]) * 4096 + gc[:, 2]
    _, first = np.unique(key, return_index=True)
    gc, nrm = gc[first], nrm[first]
    if gc.shape[0] > point_max:  # SphereCrop: the point_max voxels nearest a random voxel
        centre = gc[rng.integers(gc.shape[0])]
        d2 = ((gc - centre) ** 2).sum(1)
        keep = np.argpartition(d2, point_max)[:point_max]
        keep = keep[rng.permutation(point_max)]  # dataloader order is not spatially sorted
        gc, nrm = gc[keep], nrm[keep]
    gc = gc - gc.min(0)
    n = gc.shape[0]
    coord = ((gc + 0.5) * grid).astype(np.float32)
    colour = rng.random((n, 3)).astype(np.float32)
    feat = np.concatenate([colour, nrm.astype(np.float32)], axis=1)
    segment = rng.integers(0, 20, size=n).astype(np.int64)
    segment[rng.random(n) < 0.05] = -1
    return dict(coord=coord, grid_coord=gc.astype(np.int64), feat=feat, segment=segment)


def outdoor_scene(seed: int, point_max: int = 0, grid: float = 0.05, azimuth_steps: int = 2200):
    """LiDAR-like sweep (SURVEY 8(d) outdoor generator, BASELINE configs[4]): 64-96 beams at elevations -30..+10 deg,
    `azimuth_steps` columns; below-horizon beams hit the ground plane z = -1.8 m (range clipped to 60 m), the others
    return from U(8, 50) m; 2.5 % Gaussian range noise; voxelised at `grid` (0.05 m -> extent ~2400 voxels, depth 12);
    feat = coord | strength (in_channels = 4, nuscenes/semseg-pt-v3m1-0-base.py:16); 16 classes."""
    rng = np.random.default_rng(seed)
    beams = int(rng.integers(64, 97))
    elev = np.deg2rad(np.linspace(-30.0, 10.0, beams))[:, None]
    azim = np.linspace(0.0, 2 * np.pi, azimuth_steps, endpoint=False)[None, :] + rng.uniform(0, 2 * np.pi)
    down = np.broadcast_to(elev < 0, (beams, azimuth_steps))
    r_ground = np.minimum(1.8 / np.maximum(np.sin(-elev), 1e-6), 60.0)
    r = np.where(down, np.broadcast_to(r_ground, (beams, azimuth_steps)), rng.uniform(8.0, 50.0, (beams, azimuth_steps)))
    r = r * (1.0 + 0.025 * rng.standard_normal(r.shape))
    keep = rng.random(r.shape) < np.where(down, 1.0, 0.35)     # most above-horizon beams see the sky
    ce = np.broadcast_to(np.cos(elev), r.shape)
    p = np.stack([r * ce * np.cos(azim), r * ce * np.sin(azim), r * np.broadcast_to(np.sin(elev), r.shape)], axis=-1)[keep]
    gc = np.floor(p / grid).astype(np.int64)
    gc -= gc.min(0)
    key = (gc[:, 0] * 8192 + gc[:, 1]) * 8192 + gc[:, 2]
    _, first = np.unique(key, return_index=True)
    first = first[rng.permutation(first.shape[0])]
    if point_max and first.shape[0] > point_max:   # half the budget nearest the sensor (dense), half anywhere (keeps the extent)
        near = np.argsort((p[first] ** 2).sum(1), kind="stable")
        far = np.ones(first.shape[0], dtype=bool)
        far[near[: point_max // 2]] = False
        first = np.concatenate([first[near[: point_max // 2]], first[np.flatnonzero(far)[: point_max - point_max // 2]]])
        first = first[rng.permutation(first.shape[0])]
    gc, p = gc[first], p[first]
    gc = gc - gc.min(0)
    n = gc.shape[0]
    coord = p.astype(np.float32)
    feat = np.concatenate([coord, rng.random((n, 1)).astype(np.float32)], axis=1)
    segment = rng.integers(0, 16, size=n).astype(np.int64)
    segment[rng.random(n) < 0.05] = -1
    return dict(coord=coord, grid_coord=gc.astype(np.int64), feat=feat, segment=segment)


def collate(scenes):
    """point_collate_fn equivalent: concatenate and build the cumulative `offset`."""
    out = {k: np.concatenate([s[k] for s in scenes]) for k in scenes[0]}
    out["offset"] = np.cumsum([s["coord"].shape[0] for s in scenes]).astype(np.int64)
    return out


def indoor_batch(batch: int, point_max: int, rank: int = 0, base_seed: int = 0):
    """seeds s = 1000*rank + scene_index (+ base_seed)."""
    return collate([indoor_scene(base_seed + 1000 * rank + i, point_max) for i in range(batch)])


def to_torch(batch, device):
    import torch

    return {k: torch.from_numpy(v).to(device) for k, v in batch.items()}
