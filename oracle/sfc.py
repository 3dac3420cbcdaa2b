"""TEST INFRASTRUCTURE (oracle).  Serialization codes on CPU.

Two independent restatements of pointcept/models/utils/serialization/{default,z_order,hilbert}.py:
  * `encode_c`   -- the plain-C port in sfc_oracle.c (LUT z-order, bit-plane Hilbert), fast enough
                    for BASELINE-size inputs (8 x 102400 points x 4 orders in well under a second);
  * `encode_py`  -- a scalar numpy/python transcription of SURVEY Appendix A.1/A.2, for small
                    cases only.
Both are pinned to the reference's own `encode` through tests/golden/serialization_*.npz.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(HERE, "_build")
_LIB = os.path.join(_BUILD, "libsfc_oracle.so")
ORDER_CODES = {"z": 0, "z-trans": 1, "hilbert": 2, "hilbert-trans": 3}


def build(force: bool = False) -> str:
    """gcc -O2 -shared sfc_oracle.c -> oracle/_build/libsfc_oracle.so"""
    src = os.path.join(HERE, "sfc_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        os.makedirs(_BUILD, exist_ok=True)
        subprocess.run(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-o", _LIB, src], check=True)
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_serialize_encode.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ]
        _lib.oracle_serialize_encode.restype = None
    return _lib


def encode_c(grid_coord: np.ndarray, batch: np.ndarray | None, depth: int, orders) -> np.ndarray:
    """[k, N] int64 codes (default.py:8-24 for every order in `orders`)."""
    gc = np.ascontiguousarray(grid_coord, dtype=np.int64)
    n = gc.shape[0]
    b = None if batch is None else np.ascontiguousarray(batch, dtype=np.int64)
    oc = np.asarray([ORDER_CODES[o] for o in orders], dtype=np.int32)
    out = np.empty((len(orders), n), dtype=np.int64)
    _load().oracle_serialize_encode(
        gc.ctypes.data, None if b is None else b.ctypes.data, n, int(depth), oc.ctypes.data, len(orders), out.ctypes.data
    )
    return out


def _z_key_py(x: int, y: int, z: int, depth: int) -> int:
    # z_order.py:40-50
    key = 0
    for i in range(depth):
        key |= ((x >> i) & 1) << (3 * i + 2) | ((y >> i) & 1) << (3 * i + 1) | ((z >> i) & 1) << (3 * i)
    return key


def _hilbert_key_py(X: list, b: int) -> int:
    # SURVEY Appendix A.2 (hilbert.py:150-192)
    X = list(X)
    for q in range(b - 1, -1, -1):
        Q = 1 << q
        P = Q - 1
        for i in range(3):
            if X[i] & Q:
                X[0] ^= P
            else:
                t = (X[0] ^ X[i]) & P
                X[0] ^= t
                X[i] ^= t
    g = 0
    for q in range(b - 1, -1, -1):
        for i in range(3):
            g = (g << 1) | ((X[i] >> q) & 1)
    h = g
    s = 1
    while s < 3 * b:
        h ^= h >> s
        s <<= 1
    return h


def encode_py(grid_coord: np.ndarray, batch: np.ndarray | None, depth: int, orders) -> np.ndarray:
    gc = np.asarray(grid_coord, dtype=np.int64)
    n = gc.shape[0]
    m = (1 << depth) - 1
    out = np.empty((len(orders), n), dtype=np.int64)
    for r, o in enumerate(orders):
        for i in range(n):
            x, y, z = (int(v) & m for v in gc[i])
            if o == "z":
                c = _z_key_py(x, y, z, depth)
            elif o == "z-trans":
                c = _z_key_py(y, x, z, depth)
            elif o == "hilbert":
                c = _hilbert_key_py([x, y, z], depth)
            else:
                c = _hilbert_key_py([y, x, z], depth)
            if batch is not None:
                c |= int(batch[i]) << (3 * depth)
            out[r, i] = c
    return out


def serialization(grid_coord: np.ndarray, batch: np.ndarray, orders, depth: int | None = None):
    """Point.serialization without shuffling (structure.py:72-100): depth, code, order, inverse.
    Tie order = stable (ascending original index), the engine's canonical order (Appendix A.3)."""
    gc = np.asarray(grid_coord, dtype=np.int64)
    if depth is None:
        depth = int(gc.max() + 1).bit_length()  # structure.py:74
    code = encode_c(gc, batch, depth, orders)
    order = np.argsort(code, axis=1, kind="stable")
    inverse = np.empty_like(order)
    ar = np.arange(code.shape[1], dtype=np.int64)
    for r in range(code.shape[0]):
        inverse[r, order[r]] = ar  # structure.py:94-100
    return depth, code, order, inverse
