"""TEST INFRASTRUCTURE (oracle).  numpy restatement of the voxelisation inside GridSample
(pointcept/datasets/transform.py:867-882, fnv_hash_vec :997-1011), with a STABLE argsort so that the point found at
every rank is defined (the reference's np.argsort default is an unstable introsort: its tie order is not).
Pinned by tests/golden/gridsample.npz = outputs of the reference transform itself (voxel set, inverse, counts)."""
from __future__ import annotations

import numpy as np


def fnv_hash_vec(arr: np.ndarray) -> np.ndarray:
    a = arr.astype(np.uint64)
    h = np.full(a.shape[0], 14695981039346656037, dtype=np.uint64)
    for j in range(a.shape[1]):
        h = h * np.uint64(1099511628211)
        h = np.bitwise_xor(h, a[:, j])
    return h


def voxels(coord: np.ndarray, grid_size: float):
    scaled = coord / np.array(grid_size)                  # float32 / float64 0-d array -> float64 (transform.py:867)
    grid = np.floor(scaled).astype(np.int64)
    mn = grid.min(0)
    grid = grid - mn
    key = fnv_hash_vec(grid)
    idx_sort = np.argsort(key, kind="stable")
    _, inverse_sorted, count = np.unique(key[idx_sort], return_inverse=True, return_counts=True)
    inverse = np.zeros_like(inverse_sorted)
    inverse[idx_sort] = inverse_sorted
    return dict(grid_coord=grid, min_coord=mn, key=key, idx_sort=idx_sort, inverse=inverse, count=count)


def select_train(v, rand: np.ndarray) -> np.ndarray:
    start = np.cumsum(np.insert(v["count"], 0, 0)[:-1])
    return v["idx_sort"][start + rand % v["count"]]      # transform.py:877-882
