"""TEST INFRASTRUCTURE (oracle).  numpy restatements of two pointops kernels that exist only as CUDA in the reference
(libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-108, src/sampling/sampling_cuda_kernel.cu:15-122) --
"parity unpinned": they cannot be executed here; semantics follow the kernels' loops, with the tie order the reference
leaves implementation-defined fixed to "lower index first".  fp32 arithmetic, ((dx*dx + dy*dy) + dz*dz), like the kernels."""
from __future__ import annotations

import numpy as np


def _d2(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    d = (a.astype(np.float32) - b.astype(np.float32))
    sq = d * d
    return (sq[..., 0] + sq[..., 1]) + sq[..., 2]


def knn_query(nsample, xyz, offset, new_xyz, new_offset):
    m = new_xyz.shape[0]
    idx = np.full((m, nsample), -1, dtype=np.int32)
    d2 = np.full((m, nsample), np.float32(1e10), dtype=np.float32)
    s0 = q0 = 0
    for s1, q1 in zip(offset, new_offset):
        pts = xyz[s0:s1]
        for q in range(q0, q1):
            d = _d2(pts, new_xyz[q][None, :])
            order = np.lexsort((np.arange(len(d)), d))[:nsample]
            idx[q, : len(order)] = order + s0
            d2[q, : len(order)] = d[order]
        s0, q0 = s1, q1
    return idx, np.sqrt(d2)


def farthest_point_sampling(xyz, offset, new_offset):
    out = np.zeros(int(new_offset[-1]), dtype=np.int32)
    s0 = q0 = 0
    for s1, q1 in zip(offset, new_offset):
        if q1 > q0 and s1 > s0:
            tmp = np.full(s1 - s0, np.float32(1e10), dtype=np.float32)
            old = s0
            out[q0] = s0
            for j in range(q0 + 1, q1):
                tmp = np.minimum(_d2(xyz[s0:s1], xyz[old][None, :]), tmp)
                old = s0 + int(np.argmax(tmp))          # first maximum = lowest index
                out[j] = old
        s0, q0 = s1, q1
    return out


def ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz, new_offset, order=None):
    """libs/pointops/src/ball_query/ball_query_cuda_kernel.cu:59-123 (order=None) and
    src/random_ball_query/random_ball_query_cuda_kernel.cu:58-108 (order = permutation): -> (idx, dist) with dist = sqrt(dist2)
    as the python wrappers return (functions/query.py:75,113).  Ties: lower index first.  The sub-sampled branch returns the
    candidate's distance (the CUDA kernel writes the candidate index into dist2 there, :120)."""
    m = new_xyz.shape[0]
    idx = np.full((m, nsample), -1, dtype=np.int32)
    d2o = np.full((m, nsample), np.float32(1e10), dtype=np.float32)
    mn, mx = np.float32(min_radius) * np.float32(min_radius), np.float32(max_radius) * np.float32(max_radius)
    s0 = q0 = 0
    for s1, q1 in zip(offset, new_offset):
        cand_ids = np.arange(s0, s1) if order is None else np.asarray(order[s0:s1])
        pts = xyz[cand_ids]
        for q in range(q0, q1):
            d = _d2(pts, new_xyz[q][None, :])
            inr = (d <= np.float32(1e-5)) | ((d >= mn) & (d < mx))
            ci, cd = cand_ids[inr], d[inr]
            if order is None:
                ci, cd = ci[:2048], cd[:2048]                      # the kernel's candidate array bound
                o = np.lexsort((ci, cd))
                ci, cd = ci[o], cd[o]
                if len(ci) > nsample:
                    sep = np.float32(len(ci)) / np.float32(nsample)
                    pick = (sep * np.arange(nsample, dtype=np.float32)).astype(np.int32)
                    ci, cd = ci[pick], cd[pick]
            else:
                ci, cd = ci[:nsample], cd[:nsample]
            idx[q, :len(ci)] = ci
            d2o[q, :len(ci)] = cd
        s0, q0 = s1, q1
    return idx, np.sqrt(d2o)
