"""Mirror of torch_scatter.segment_csr as the reference calls it
(pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:416-421, pointcept/models/default.py:332):
    segment_csr(src, indptr, out=None, reduce="sum") -> Tensor   (reduce in sum|mean|min|max)
backed by the fused gather + CSR reduce kernel (rows.hip).  max/min backward goes to the first
arg-max row of each segment.
"""
from __future__ import annotations

from . import functional as PF
from ._lib import PtcoreError


def segment_csr(src, indptr, out=None, reduce="sum"):
    if out is not None:
        raise PtcoreError("segment_csr: `out=` is not implemented")
    if reduce not in ("sum", "mean", "min", "max"):
        raise PtcoreError(f"segment_csr: bad reduce {reduce!r}")
    if src.dim() > 2:
        flat = src.reshape(src.shape[0], -1)
        return PF.segment_csr(flat, indptr, reduce).reshape((indptr.numel() - 1,) + tuple(src.shape[1:]))
    return PF.segment_csr(src, indptr, reduce)
