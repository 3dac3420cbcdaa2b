"""`pointrope` operator API on the engine: libs/pointrope's extension module surface

    pointrope.pointrope(tokens [B,N,H,D], positions [B,N,3] int64, base, F0)      in place (pointrope.cpp:51-67)

plus the autograd wrapper and module LitePT builds on it (pointcept/models/litept/litept_v1.py:27-59), so that the
reference's model file finds the same names.  The CUDA kernel (kernels.cu:19-100) is replaced by csrc/rope.hip.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import PtcoreError


def pointrope(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """same checks as pointrope.cpp:56-61, same in-place contract"""
    if tokens.dim() != 4:
        raise PtcoreError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise PtcoreError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise PtcoreError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise PtcoreError("seq_length differs between tokens & positions")
    if positions.size(2) != 3:
        raise PtcoreError("positions.shape[2] must be equal to 3")
    if tokens.is_cuda != positions.is_cuda:
        raise PtcoreError("tokens and positions are not on the same device")
    ops.rope3d_(tokens, positions, base, fwd)


class PointROPE_func(torch.autograd.Function):
    """litept_v1.py:27-46: forward rotates in place with F0, backward rotates the incoming gradient with -F0."""

    @staticmethod
    def forward(ctx, tokens, positions, base, F0=1):
        ctx.save_for_backward(positions)
        ctx.saved_base, ctx.saved_F0 = base, F0
        pointrope(tokens, positions, base, F0)
        ctx.mark_dirty(tokens)
        return tokens

    @staticmethod
    def backward(ctx, grad_res):
        positions, base, F0 = ctx.saved_tensors[0], ctx.saved_base, ctx.saved_F0
        grad_res = grad_res.contiguous()
        pointrope(grad_res, positions, base, -F0)
        return grad_res, None, None, None


class PointROPE(torch.nn.Module):
    """litept_v1.py:48-59: tokens arrive [B, H, N, D]"""

    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base, self.F0 = freq, F0

    def forward(self, tokens, positions):
        tokens = tokens.transpose(1, 2).contiguous()
        positions = positions.contiguous()
        tokens = PointROPE_func.apply(tokens, positions, self.base, self.F0)
        return tokens.transpose(1, 2).contiguous()
