"""TEST INFRASTRUCTURE (oracle).  CPU restatement (numpy, fp32 arithmetic) of libs/pointrope: the rotary embedding of
libs/pointrope/kernels.cu:19-75 (the formula of the CUDA kernel, which is what training runs) -- identical math to
pointrope_cpu (libs/pointrope/pointrope.cpp:13-49) up to the order of one multiply / divide:
    kernel : f = pos * (F0 / base^(i/Q))          cpu : f = F0 * pos / base^(i/Q)
Pinned by tests/golden/pointrope.npz = outputs of the reference's OWN pointrope_cpu compiled from
/root/reference/libs/pointrope/pointrope.cpp (oracle/build_ref.py -> oracle/_ref/), and live against that build in
tests/test_oracle_vs_reference.py.
"""
from __future__ import annotations

import numpy as np


def pointrope(tokens: np.ndarray, positions: np.ndarray, base: float, fwd: float) -> np.ndarray:
    """tokens [B,N,H,D] float32 (D % 6 == 0), positions [B,N,3] int64 -> rotated copy (the operator itself is in place)."""
    tok = np.asarray(tokens, dtype=np.float32).copy()
    B, N, H, D = tok.shape
    assert D % 6 == 0 and positions.shape == (B, N, 3)
    Q = D // 6
    i = np.arange(Q, dtype=np.float32)
    inv_freq = (np.float32(fwd) / np.power(np.float32(base), i / np.float32(Q))).astype(np.float32)      # kernels.cu:44
    for a in range(3):
        f = positions[:, :, a].astype(np.float32)[:, :, None] * inv_freq[None, None, :]                      # [B,N,Q]  kernels.cu:54
        c, s = np.cos(f).astype(np.float32)[:, :, None, :], np.sin(f).astype(np.float32)[:, :, None, :]
        u = tok[:, :, :, a * 2 * Q:a * 2 * Q + Q].copy()
        v = tok[:, :, :, a * 2 * Q + Q:a * 2 * Q + 2 * Q].copy()
        tok[:, :, :, a * 2 * Q:a * 2 * Q + Q] = u * c - v * s
        tok[:, :, :, a * 2 * Q + Q:a * 2 * Q + 2 * Q] = v * c + u * s
    return tok


def rope_xyz(tokens: np.ndarray, xyz: np.ndarray, inv_freq: np.ndarray, sign: float = 1.0) -> np.ndarray:
    """PT-v3m3 `Point3DRoPE` (pointcept/models/point_transformer_v3/point_transformer_v3m3_utonia.py:43-102) on tokens [n, H, D]
    (D % 6 == 0), continuous coordinates xyz [n, 3] and the module's inv_freq [D/6] (:53-55):
        emb = cat(x f, x f | y f, y f | z f, z f)                      (:58-73)
        out = tok * cos(emb) + rotate_half_per_chunk(tok) * sin(emb)    (:75-77, 88-92)
    i.e. inside each of the three chunks of D/3, element i of the first half rotates with element i of the second half.
    sign = -1 gives the inverse rotation (= the gradient map).  fp32 arithmetic, rotated copy.
    Pinned by the rope_* arrays of tests/golden/ptv3m3_tiny.npz (outputs of the reference class) and live in
    tests/test_oracle_vs_reference.py."""
    tok = np.asarray(tokens, dtype=np.float32).copy()
    n, H, D = tok.shape
    assert D % 6 == 0 and xyz.shape == (n, 3)
    Q = D // 6
    inv_freq = np.asarray(inv_freq, dtype=np.float32).reshape(Q)
    for a in range(3):
        f = np.asarray(xyz, dtype=np.float32)[:, a:a + 1] * inv_freq[None, :]                                # [n, Q]
        c, s = np.cos(f).astype(np.float32)[:, None, :], (np.float32(sign) * np.sin(f).astype(np.float32))[:, None, :]
        u = tok[:, :, a * 2 * Q:a * 2 * Q + Q].copy()
        v = tok[:, :, a * 2 * Q + Q:a * 2 * Q + 2 * Q].copy()
        tok[:, :, a * 2 * Q:a * 2 * Q + Q] = u * c - v * s
        tok[:, :, a * 2 * Q + Q:a * 2 * Q + 2 * Q] = v * c + u * s
    return tok
