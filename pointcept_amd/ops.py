"""Raw (non-differentiable) Python wrappers over the C-ABI of libptcore.so.

Every function takes CUDA tensors, allocates outputs / workspaces as torch tensors, and enqueues
the kernels on torch's current stream.  CPU tensors raise PtcoreError -- there is no fallback.
Differentiable versions live in pointcept_amd/functional.py.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import PtcoreError, check, dtype_code, lib, ptr, require_cuda, stream_ptr


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# serialization
# ------------------------------------------------------------------------------------------------
def serialize_encode(grid_coord: torch.Tensor, batch: Optional[torch.Tensor], depth: int,
                     orders: Sequence[str]) -> torch.Tensor:
    """serialization.encode for all `orders` at once -> code [k,N] int64
    (pointcept/models/utils/serialization/default.py:8-24)."""
    require_cuda(grid_coord, batch)
    if grid_coord.dtype not in (torch.int64, torch.int32):
        raise PtcoreError(f"grid_coord must be int32/int64, got {grid_coord.dtype}")
    gc = grid_coord.contiguous()
    n = gc.shape[0]
    b = None if batch is None else batch.to(torch.int64).contiguous()
    k = len(orders)
    oc = (ctypes.c_int * k)(*[_lib.ORDER_CODES[o] for o in orders])
    out = torch.empty((k, n), dtype=torch.int64, device=gc.device)
    check(lib().ptc_serialize_encode(ptr(gc), 1 if gc.dtype == torch.int64 else 0, ptr(b), n, int(depth),
                                     ctypes.cast(oc, ctypes.c_void_p), k, ptr(out), stream_ptr()),
          "ptc_serialize_encode")
    return out


def sort_keys(keys: torch.Tensor, begin_bit: int, end_bit: int, want_inverse: bool = True):
    """Stable argsort of every row of keys [k,N] int64 over bits [begin_bit,end_bit) ->
    (order [k,N] int64, inverse [k,N] int64 | None)   (structure.py:93-100)."""
    require_cuda(keys)
    if keys.dtype != torch.int64:
        raise PtcoreError("keys must be int64")
    squeeze = keys.dim() == 1
    k2 = keys.reshape(1, -1) if squeeze else keys
    k2 = k2.contiguous()
    k, n = k2.shape
    order = torch.empty_like(k2)
    inverse = torch.empty_like(k2) if want_inverse else None
    nbytes = lib().ptc_sort_keys_workspace_bytes(n, k)
    ws = _ws(nbytes, k2.device)
    check(lib().ptc_sort_keys(ptr(k2), n, k, int(begin_bit), int(end_bit), ptr(order), ptr(inverse), ptr(ws), nbytes,
                              stream_ptr()), "ptc_sort_keys")
    if squeeze:
        return order[0], (inverse[0] if want_inverse else None)
    return order, inverse


def exclusive_scan_i32(x: torch.Tensor) -> torch.Tensor:
    require_cuda(x)
    x = x.to(torch.int32).contiguous()
    n = x.numel()
    out = torch.empty(n, dtype=torch.int64, device=x.device)
    nbytes = lib().ptc_exclusive_scan_workspace_bytes(n)
    ws = _ws(nbytes, x.device)
    check(lib().ptc_exclusive_scan_i32(ptr(x), n, ptr(out), ptr(ws), nbytes, stream_ptr()), "ptc_exclusive_scan_i32")
    return out


# ------------------------------------------------------------------------------------------------
# patch padding maps
# ------------------------------------------------------------------------------------------------
def pad_sizes(offset_host: Sequence[int], patch: int):
    """(n, n_pad, n_seq) from a host copy of `offset` (ptv3m1:123-138,156-164)."""
    prev, n_pad, n_seq = 0, 0, 0
    for o in offset_host:
        ni = int(o) - prev
        prev = int(o)
        p = ((ni + patch - 1) // patch) * patch if ni > patch else ni
        n_pad += p
        n_seq += (p + patch - 1) // patch
    return prev, n_pad, n_seq


def patch_pad_maps(offset: torch.Tensor, offset_host: Sequence[int], patch: int):
    """SerializedAttention.get_padding_and_inverse (ptv3m1:114-170) in one launch.
    Returns pad [N'] i64, unpad [N] i64, cu_seqlens [n_seq+1] i32, dup [N] i64."""
    require_cuda(offset)
    off = offset.to(torch.int64).contiguous()
    n, n_pad, n_seq = pad_sizes(offset_host, patch)
    dev = off.device
    pad = torch.empty(n_pad, dtype=torch.int64, device=dev)
    unpad = torch.empty(n, dtype=torch.int64, device=dev)
    cu = torch.empty(n_seq + 1, dtype=torch.int32, device=dev)
    dup = torch.empty(n, dtype=torch.int64, device=dev)
    check(lib().ptc_patch_pad_maps(ptr(off), off.numel(), int(patch), n, n_pad, n_seq, ptr(pad), ptr(unpad), ptr(cu),
                                   ptr(dup), stream_ptr()), "ptc_patch_pad_maps")
    return pad, unpad, cu, dup


def attn_tables(order: torch.Tensor, inverse: torch.Tensor, pad: torch.Tensor, unpad: torch.Tensor, dup: torch.Tensor):
    """int32 gather tables of one serialization order (ptv3m1:184-188,216 and their backward) in one launch:
    (t_qkv_fwd [1,N'], t_qkv_bwd [2,N], t_proj_fwd [1,N], t_proj_bwd [1,N'])."""
    require_cuda(order, inverse, pad, unpad, dup)
    n, n_pad = order.numel(), pad.numel()
    dev = order.device
    t1 = torch.empty((1, n_pad), dtype=torch.int32, device=dev)
    t2 = torch.empty((2, n), dtype=torch.int32, device=dev)
    t3 = torch.empty((1, n), dtype=torch.int32, device=dev)
    t4 = torch.empty((1, n_pad), dtype=torch.int32, device=dev)
    check(lib().ptc_attn_tables(ptr(order.contiguous()), ptr(inverse.contiguous()), ptr(pad), ptr(unpad), ptr(dup), n, n_pad,
                                ptr(t1), ptr(t2), ptr(t3), ptr(t4), stream_ptr()), "ptc_attn_tables")
    return t1, t2, t3, t4


# ------------------------------------------------------------------------------------------------
# pooling maps
# ------------------------------------------------------------------------------------------------
def pool_level_counts(code0: torch.Tensor, order0: torch.Tensor, batch_shift: int, n_batch: int, shifts: Sequence[int]):
    """counts [len(shifts), n_batch] int64 (device): distinct values of code0 >> shifts[l] per scene -- the point counts
    of every pooled level (ptv3m1:384-390), one launch, no sync (the caller fetches them with its own single copy)."""
    require_cuda(code0, order0)
    import ctypes

    counts = torch.empty((len(shifts), n_batch), dtype=torch.int64, device=code0.device)
    arr = (ctypes.c_int * len(shifts))(*[int(v) for v in shifts])
    check(lib().ptc_pool_level_counts(ptr(code0.contiguous()), ptr(order0.contiguous()), code0.numel(), int(batch_shift), int(n_batch),
                                      ctypes.cast(arr, ctypes.c_void_p), len(shifts), ptr(counts), stream_ptr()),
          "ptc_pool_level_counts")
    return counts


def pool_maps(code0: torch.Tensor, order0: torch.Tensor, shift: int, n_cluster: Optional[int] = None):
    """cluster (= pooling_inverse), idx_ptr, head for SerializedPooling (ptv3m1:383-396).
    One host sync (the number of clusters sizes the outputs, as torch.unique does in the reference) unless the caller
    already knows `n_cluster` (pool_level_counts)."""
    require_cuda(code0, order0)
    code0 = code0.contiguous()
    order0 = order0.contiguous()
    n = code0.numel()
    dev = code0.device
    cluster = torch.empty(n, dtype=torch.int64, device=dev)
    ncl = torch.empty(1, dtype=torch.int64, device=dev)
    nbytes = lib().ptc_pool_maps_workspace_bytes(n)
    ws = _ws(nbytes, dev)
    check(lib().ptc_pool_maps_count(ptr(code0), ptr(order0), n, int(shift), ptr(cluster), ptr(ncl), ptr(ws), nbytes,
                                    stream_ptr()), "ptc_pool_maps_count")
    if n_cluster is None:
        n_cluster = int(ncl.item())
    idx_ptr = torch.empty(n_cluster + 1, dtype=torch.int64, device=dev)
    head = torch.empty(n_cluster, dtype=torch.int64, device=dev)
    check(lib().ptc_pool_maps_fill(ptr(order0), ptr(cluster), n, n_cluster, ptr(idx_ptr), ptr(head), stream_ptr()),
          "ptc_pool_maps_fill")
    return cluster, idx_ptr, head


def pool_child_codes(code: torch.Tensor, head: torch.Tensor, shift: int) -> torch.Tensor:
    require_cuda(code, head)
    code = code.contiguous()
    head = head.contiguous()
    k, n = code.shape
    nc = head.numel()
    out = torch.empty((k, nc), dtype=torch.int64, device=code.device)
    check(lib().ptc_pool_child_codes(ptr(code), n, k, ptr(head), nc, int(shift), ptr(out), stream_ptr()),
          "ptc_pool_child_codes")
    return out


# ------------------------------------------------------------------------------------------------
# voxelisation (GridSample front end)
# ------------------------------------------------------------------------------------------------
def voxel_keys(coord: torch.Tensor, grid_size: float):
    """floor(coord / grid_size) (float64 division), min-shifted, and the FNV-64 key of every point
    (pointcept/datasets/transform.py:867-875,997-1011) -> (grid_coord [N,3] i64, min_coord [3] i64, key [N] i64)."""
    require_cuda(coord)
    if coord.dtype != torch.float32 or coord.dim() != 2 or coord.shape[1] != 3:
        raise PtcoreError("coord must be float32 [N,3]")
    c = coord.contiguous()
    n = c.shape[0]
    grid = torch.empty((n, 3), dtype=torch.int64, device=c.device)
    mn = torch.empty(3, dtype=torch.int64, device=c.device)
    key = torch.empty(n, dtype=torch.int64, device=c.device)
    check(lib().ptc_voxel_keys(ptr(c), n, float(grid_size), ptr(grid), ptr(mn), ptr(key), stream_ptr()), "ptc_voxel_keys")
    return grid, mn, key


# ------------------------------------------------------------------------------------------------
# pointops subset
# ------------------------------------------------------------------------------------------------
def _xyz(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 3:
        raise PtcoreError(f"{name} must be float32 [N,3]")
    return t.contiguous()


def knn_query(nsample: int, xyz, offset, new_xyz, new_offset):
    """-> (idx [m, nsample] int32, dist [m, nsample] fp32), libs/pointops/functions/query.py:7-26."""
    require_cuda(xyz, offset, new_xyz, new_offset)
    x, q = _xyz(xyz, "xyz"), _xyz(new_xyz, "new_xyz")
    off, noff = offset.to(torch.int32).contiguous(), new_offset.to(torch.int32).contiguous()
    if off.numel() != noff.numel() or off.numel() == 0:
        raise PtcoreError("offset / new_offset must list the same (non-zero) number of scenes")
    m = q.shape[0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=x.device)
    dist = torch.empty((m, nsample), dtype=torch.float32, device=x.device)
    check(lib().ptc_knn_query(ptr(x), ptr(off), ptr(q), ptr(noff), off.numel(), x.shape[0], m, int(nsample), ptr(idx), ptr(dist),
                              stream_ptr()), "ptc_knn_query")
    return idx, dist


def ball_query(nsample: int, max_radius: float, min_radius: float, xyz, offset, new_xyz, new_offset, order=None):
    """-> (idx [m, nsample] int32 (-1 = none), dist [m, nsample] fp32 (1e5 = none)), libs/pointops/functions/query.py:29-113.
    order = None: ball_query (sorted by distance, uniformly sub-sampled); order = int32 permutation of every scene's points:
    random_ball_query (first nsample in-range points in that order)."""
    require_cuda(xyz, offset, new_xyz, new_offset, order)
    if not float(min_radius) < float(max_radius):
        raise PtcoreError("min_radius must be smaller than max_radius")      # query.py:45,93
    x, q = _xyz(xyz, "xyz"), _xyz(new_xyz, "new_xyz")
    off, noff = offset.to(torch.int32).contiguous(), new_offset.to(torch.int32).contiguous()
    if off.numel() != noff.numel() or off.numel() == 0:
        raise PtcoreError("offset / new_offset must list the same (non-zero) number of scenes")
    if order is not None:
        order = order.to(torch.int32).contiguous()
        if order.numel() != x.shape[0]:
            raise PtcoreError("order must be a permutation of the source points")
    m = q.shape[0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=x.device)
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=x.device)
    check(lib().ptc_ball_query(ptr(x), ptr(off), ptr(q), ptr(noff), ptr(order), off.numel(), x.shape[0], m, int(nsample), float(min_radius),
                               float(max_radius), ptr(idx), ptr(d2), stream_ptr()), "ptc_ball_query")
    return idx, torch.sqrt(d2)


def farthest_point_sampling(xyz, offset, new_offset):
    """-> idx [new_offset[-1]] int32, libs/pointops/functions/sampling.py:7-24 (one host sync for the output size)."""
    require_cuda(xyz, offset, new_offset)
    x = _xyz(xyz, "xyz")
    off, noff = offset.to(torch.int32).contiguous(), new_offset.to(torch.int32).contiguous()
    if off.numel() != noff.numel() or off.numel() == 0:
        raise PtcoreError("offset / new_offset must list the same (non-zero) number of scenes")
    idx = torch.zeros(int(noff[-1].item()), dtype=torch.int32, device=x.device)
    tmp = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(lib().ptc_farthest_point_sampling(ptr(x), ptr(off), ptr(noff), off.numel(), x.shape[0], ptr(tmp), ptr(idx), stream_ptr()),
          "ptc_farthest_point_sampling")
    return idx


# ------------------------------------------------------------------------------------------------
# edge-list operators of libs/pointops (grouping / interpolation / aggregation / subtraction): csrc/pointops_edges.hip
# ------------------------------------------------------------------------------------------------
def _f32_rows(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise PtcoreError(f"{name} must be float32 (the reference kernels are fp32-only)")
    return t.contiguous()


def _edge_idx(idx: torch.Tensor):
    if idx.dim() != 2:
        raise PtcoreError("idx must be [m, nsample]")
    return idx.to(torch.int32).contiguous()


def edge_rows(mode: int, src, a, idx, out=None, out_col0: int = 0):
    """Per-edge rows (mode 0 src[j], 1 a[t] - src[j], 2 src[j] - a[t] with zeros for empty slots) -> out [m, nsample, c], or written
    into the column window [out_col0, out_col0 + c) of a caller's wider `out` [m, nsample, width]."""
    require_cuda(src, a, idx, out)
    src, idx = _f32_rows(src, "src"), _edge_idx(idx)
    a = None if a is None else _f32_rows(a, "a")
    m, ns = idx.shape
    c = src.shape[1]
    if out is None:
        out = torch.empty((m, ns, c), dtype=torch.float32, device=src.device)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.shape[:2] != (m, ns):
        raise PtcoreError("edge_rows: out must be a contiguous fp32 [m, nsample, width] tensor")
    check(lib().ptc_edge_rows_fwd(int(mode), ptr(src), ptr(a), ptr(idx), m * ns, ns, c, src.shape[0], ptr(out), out.shape[2], int(out_col0),
                                  stream_ptr()), "ptc_edge_rows_fwd")
    return out


def edge_reduce(mode: int, src, pos, w, idx, m: int, nsample: int, c: int, w_c: int = 1, pos_stride: int = 0, pos_col0: int = 0):
    """Per-target sums over the nsample edges of a row -> [m, c] (mode 0 interpolation, 1 aggregation, 2 plain row sums of `pos`)."""
    require_cuda(src, pos, w, idx)
    out = torch.empty((m, c), dtype=torch.float32, device=(pos if src is None else src).device)
    n_src = 0 if src is None else src.shape[0]
    check(lib().ptc_edge_reduce_fwd(int(mode), ptr(src), ptr(pos), int(pos_stride), int(pos_col0), ptr(w), ptr(idx), int(m), int(nsample),
                                    int(c), int(w_c), n_src, ptr(out), stream_ptr()), "ptc_edge_reduce_fwd")
    return out


class EdgeCSR:
    """The edges of idx [m, nsample] sorted by source row (stable: ascending edge index inside a row) + the CSR pointer: what the
    segmented, atomics-free gradients of the gathered operands walk.  Built on first use: the engine's radix sort over
    bit_length(n_src) bits and two small kernels."""

    def __init__(self, idx: torch.Tensor, n_src: int):
        require_cuda(idx)
        self.idx, self.n_src = _edge_idx(idx), int(n_src)
        self.order = self.indptr = None

    def build(self):
        if self.order is None:
            e = self.idx.numel()
            keys = torch.empty(e, dtype=torch.int64, device=self.idx.device)
            check(lib().ptc_edge_csr_keys(ptr(self.idx), e, self.n_src, ptr(keys), stream_ptr()), "ptc_edge_csr_keys")
            self.order, _ = sort_keys(keys, 0, max(1, int(self.n_src).bit_length()), want_inverse=False) if e else (keys, None)
            self.indptr = torch.empty(self.n_src + 1, dtype=torch.int64, device=self.idx.device)
            check(lib().ptc_edge_csr_ptr(ptr(keys), ptr(self.order), e, self.n_src, ptr(self.indptr), stream_ptr()), "ptc_edge_csr_ptr")
        return self


def edge_scatter_bwd(mode: int, csr: EdgeCSR, g, w, nsample: int, c: int, w_c: int = 1, g_col0: int = 0):
    """grad of the gathered operand [n_src, c]: segmented sum over the edges of every source row (fixed order, no atomics).
    g: [E or m, width] rows (fp32, contiguous); the window [g_col0, g_col0 + c) of each row is summed."""
    require_cuda(g, w)
    csr.build()
    g = _f32_rows(g, "grad").reshape(-1, g.shape[-1])
    out = torch.empty((csr.n_src, c), dtype=torch.float32, device=g.device)
    check(lib().ptc_edge_scatter_bwd(int(mode), ptr(csr.order), ptr(csr.indptr), ptr(g), g.shape[1], int(g_col0), ptr(w), int(nsample), int(c),
                                     int(w_c), csr.n_src, ptr(out), stream_ptr()), "ptc_edge_scatter_bwd")
    return out


def pair_dot_weighted(a, b, w, ia, ib) -> torch.Tensor:
    """out[m, g] = sum_c a[ia[m], g, c] b[ib[m], g, c] (w[c] | 1); a [n_a, g, c], b [n_b, g, c] fp32, ia / ib [m] int32"""
    require_cuda(a, b, w, ia, ib)
    a, b = a.float().contiguous(), b.float().contiguous()
    if a.dim() != 3 or b.dim() != 3 or a.shape[1:] != b.shape[1:]:
        raise PtcoreError(f"pair_dot_weighted: rows must be [n, g, c], got {tuple(a.shape)} / {tuple(b.shape)}")
    ia, ib = ia.reshape(-1).to(torch.int32).contiguous(), ib.reshape(-1).to(torch.int32).contiguous()
    if ia.numel() != ib.numel():
        raise PtcoreError("pair_dot_weighted: index lists differ in length")
    w = None if w is None else w.float().contiguous()
    _, g, c = a.shape
    if w is not None and w.numel() != c:
        raise PtcoreError(f"pair_dot_weighted: weight has {w.numel()} entries, rows have {c} channels")
    out = torch.empty((ia.numel(), g), dtype=torch.float32, device=a.device)
    check(lib().ptc_pair_dot_weighted(ptr(a), ptr(b), ptr(w), ptr(ia), ptr(ib), ia.numel(), a.shape[0], b.shape[0], g, c, ptr(out),
                                      stream_ptr()), "ptc_pair_dot_weighted")
    return out


def pair_segment_sum(s, b, w, self_rows, csr: "EdgeCSR", oidx, want_prod: bool = False):
    """A[n, g, c] = sum over the pairs e of row n (csr: the pairs by that row, ascending) of s[e, g] b[oidx[e], g, c];
    -> (A * (w | 1), self_rows * A | None).  s [m, g], b [n_b, g, c] fp32, oidx [m] int32."""
    require_cuda(s, b, w, self_rows, oidx)
    csr.build()
    s, b = s.float().contiguous(), b.float().contiguous()
    oidx = oidx.reshape(-1).to(torch.int32).contiguous()
    _, g, c = b.shape
    if s.shape != (oidx.numel(), g):
        raise PtcoreError(f"pair_segment_sum: coefficients {tuple(s.shape)} for {oidx.numel()} pairs of {g} groups")
    w = None if w is None else w.float().contiguous()
    out = torch.empty((csr.n_src, g, c), dtype=torch.float32, device=b.device)
    prod = None
    if want_prod:
        self_rows = self_rows.float().contiguous()
        if tuple(self_rows.shape) != (csr.n_src, g, c):
            raise PtcoreError("pair_segment_sum: self rows do not match the output")
        prod = torch.empty_like(out)
    check(lib().ptc_pair_segment_sum(ptr(s), ptr(b), ptr(w), ptr(self_rows) if want_prod else None, ptr(csr.order), ptr(csr.indptr), ptr(oidx),
                                     csr.n_src, b.shape[0], g, c, ptr(out), ptr(prod), stream_ptr()), "ptc_pair_segment_sum")
    return out, prod


def aggregation_edge_bwd(src, pos, w, idx, g):
    """-> (grad_position [m, nsample, c], grad_weight [m, nsample, w_c]) of libs/pointops aggregation."""
    require_cuda(src, pos, w, idx, g)
    m, ns, c = pos.shape
    gp, gw = torch.empty_like(pos), torch.empty_like(w)
    check(lib().ptc_aggregation_edge_bwd(ptr(src), ptr(pos), ptr(w), ptr(idx), ptr(g), m, ns, c, w.shape[-1], src.shape[0], ptr(gp), ptr(gw),
                                         stream_ptr()), "ptc_aggregation_edge_bwd")
    return gp, gw


# ------------------------------------------------------------------------------------------------
# rows
# ------------------------------------------------------------------------------------------------
def gather_rows(src: torch.Tensor, idx: torch.Tensor, idx2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = src[idx[i]] (+ src[idx2[i]] where idx2[i] >= 0); idx[i] < 0 -> zeros."""
    require_cuda(src, idx, idx2)
    if src.dim() != 2:
        raise PtcoreError("gather_rows expects [N,C] features")
    src = src.contiguous()
    idx = idx.to(torch.int64).contiguous()
    if idx2 is not None:
        idx2 = idx2.to(torch.int64).contiguous()
    n_out, c = idx.numel(), src.shape[1]
    out = torch.empty((n_out, c), dtype=src.dtype, device=src.device)
    check(lib().ptc_gather_rows(ptr(src), src.shape[0], ptr(idx), ptr(idx2), n_out, c, dtype_code(src), ptr(out),
                                stream_ptr()), "ptc_gather_rows")
    return out


def gather_rows_add_supported(src: torch.Tensor, addend: torch.Tensor) -> bool:
    return (src.dim() == 2 and addend.dim() == 2 and src.dtype == addend.dtype and src.shape[1] == addend.shape[1]
            and src.dtype in (torch.float32, torch.bfloat16, torch.float16) and src.shape[1] % (4 if src.dtype == torch.float32 else 8) == 0)


def gather_rows_add(src: torch.Tensor, idx: torch.Tensor, addend: torch.Tensor) -> torch.Tensor:
    """out[i] = addend[i] + src[idx[i]] in one pass (SerializedUnpooling, ptv3m1:478)."""
    require_cuda(src, idx, addend)
    if not gather_rows_add_supported(src, addend) or addend.shape[0] != idx.numel():
        raise PtcoreError(f"gather_rows_add: src {tuple(src.shape)} {src.dtype}, addend {tuple(addend.shape)} {addend.dtype}, idx {tuple(idx.shape)}")
    src, addend = src.contiguous(), addend.contiguous()
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty_like(addend)
    check(lib().ptc_gather_rows_add(ptr(src), src.shape[0], ptr(idx), ptr(addend), idx.numel(), src.shape[1], dtype_code(src), ptr(out), stream_ptr()),
          "ptc_gather_rows_add")
    return out


def segment_csr_fwd(src: torch.Tensor, perm: Optional[torch.Tensor], indptr: torch.Tensor, reduce: str):
    """out[s] = reduce_{r in [indptr[s], indptr[s+1])} src[perm[r]]; returns (out, arg|None)."""
    require_cuda(src, perm, indptr)
    src = src.contiguous()
    indptr = indptr.to(torch.int64).contiguous()
    if perm is not None:
        perm = perm.to(torch.int64).contiguous()
    n_seg, c = indptr.numel() - 1, src.shape[1]
    out = torch.empty((n_seg, c), dtype=src.dtype, device=src.device)
    arg = torch.empty((n_seg, c), dtype=torch.int32, device=src.device) if reduce in ("max", "min") else None
    check(lib().ptc_segment_csr_fwd(ptr(src), ptr(perm), ptr(indptr), n_seg, c, dtype_code(src),
                                    _lib.REDUCE_CODES[reduce], ptr(out), ptr(arg), stream_ptr()), "ptc_segment_csr_fwd")
    return out, arg


def segment_csr_bwd(grad_out: torch.Tensor, perm: Optional[torch.Tensor], indptr: torch.Tensor,
                    arg: Optional[torch.Tensor], n_src: int, reduce: str, covers_all: bool = False) -> torch.Tensor:
    require_cuda(grad_out, perm, indptr, arg)
    grad_out = grad_out.contiguous()
    n_seg, c = indptr.numel() - 1, grad_out.shape[1]
    # rows outside every segment (none when perm is a full permutation) must read as zero
    covered = int(n_src)
    alloc = torch.empty if covers_all else torch.zeros      # covers_all: every row belongs to a segment and gets written
    gsrc = alloc((covered, c), dtype=grad_out.dtype, device=grad_out.device)
    check(lib().ptc_segment_csr_bwd(ptr(grad_out), ptr(perm), ptr(indptr), ptr(arg), n_seg, covered, c,
                                    dtype_code(grad_out), _lib.REDUCE_CODES[reduce], ptr(gsrc), stream_ptr()),
          "ptc_segment_csr_bwd")
    return gsrc


# ------------------------------------------------------------------------------------------------
# rulebooks
# ------------------------------------------------------------------------------------------------
VOX_MAX = 1 << 18
BATCH_MAX = 1023


def check_coord_range(coord_max_host, batch_size: int, spatial_shape=None) -> None:
    """The voxel hash packs (batch, x, y, z) into 10 + 3 x 18 bits (csrc/voxel_hash.h): coordinates outside [0, 2^18) --
    ptc_coord_max reports a negative coordinate as a negative maximum -- or outside the declared spatial_shape, or more
    than BATCH_MAX batch items, would alias other voxels' keys and give silently wrong neighbour maps.  Host integers
    only (they ride in the one host sync of the forward): no device work."""
    for a, m in enumerate(coord_max_host):
        m = int(m)
        if m < 0 or m >= VOX_MAX:
            raise PtcoreError(f"grid_coord axis {a}: coordinates must lie in [0, {VOX_MAX}) (max reported {m}; negative = a negative "
                              "coordinate in the input)")
        if spatial_shape is not None and m >= int(spatial_shape[a]):
            raise PtcoreError(f"grid_coord axis {a}: maximum {m} outside the declared spatial_shape {list(spatial_shape)}")
    if int(batch_size) > BATCH_MAX:
        raise PtcoreError(f"batch size {batch_size} exceeds the {BATCH_MAX} batch items the voxel key can hold")


class HashTable:
    def __init__(self, indices: torch.Tensor):
        require_cuda(indices)
        if indices.dtype != torch.int32 or indices.dim() != 2 or indices.shape[1] != 4:
            raise PtcoreError("indices must be int32 [N,4] (batch,x,y,z)")
        self.indices = indices.contiguous()
        n = self.indices.shape[0]
        self.size = int(lib().ptc_hash_table_size(n))            # buckets of 2x2x2 voxels, 64 bytes each
        self.nbytes = int(lib().ptc_hash_table_bytes(n))
        self.buf = torch.empty(self.nbytes // 8, dtype=torch.int64, device=indices.device)   # torch allocations are >= 256 B aligned
        check(lib().ptc_hash_build(ptr(self.indices), n, ptr(self.buf), self.nbytes, stream_ptr()), "ptc_hash_build")


def rulebook_subm(indices: torch.Tensor, ksize: int, table: Optional[HashTable] = None) -> torch.Tensor:
    """Gather table nbr [ksize^3, N] int32 of a submanifold convolution."""
    if table is None:
        table = HashTable(indices)
    ind = table.indices
    n = ind.shape[0]
    nbr = torch.empty((ksize ** 3, n), dtype=torch.int32, device=ind.device)
    check(lib().ptc_rulebook_subm(ptr(ind), n, int(ksize), ptr(table.buf), table.nbytes, ptr(nbr), stream_ptr()),
          "ptc_rulebook_subm")
    return nbr


BLOCK_BM, BLOCK_HCAP = 128, 416     # = C7_BM, C7_HCAP of csrc/conv7.h


class BlockTables:
    """Block-local form of a submanifold 3^3 gather table `nbr` (csrc/blocks.hip): per block of 128 consecutive rows the ascending list
    of distinct input rows (`hid` [n_blocks, hcap], `hcnt` [n_blocks]; -1 = does not fit) and the uint16 tables `tab`
    [2, n_blocks, 28, 32, 4] of LDS byte offsets of those rows (0: 64-channel rows, 1: 32-channel rows; include/ptcore.h).  Consumed by
    spconv_fwd(..., blk=...) (csrc/conv7.h)."""

    def __init__(self, nbr: torch.Tensor, bm: int = BLOCK_BM, hcap: int = BLOCK_HCAP):
        require_cuda(nbr)
        if nbr.dtype != torch.int32 or nbr.dim() != 2 or nbr.shape[0] != 27:
            raise PtcoreError("nbr must be int32 [27, n]")
        self.nbr = nbr.contiguous()
        kv, n = self.nbr.shape
        self.bm, self.hcap = int(bm), int(hcap)
        nblk = max(1, (n + self.bm - 1) // self.bm)
        dev = nbr.device
        self.tab = torch.empty(int(lib().ptc_rulebook_blocks_tab_bytes(n)) // 2, dtype=torch.int16, device=dev).view(2, nblk, 28, 32, 4)
        self.hid = torch.empty((nblk, self.hcap), dtype=torch.int32, device=dev)
        self.hcnt = torch.empty(nblk, dtype=torch.int32, device=dev)
        self.n_overflow = torch.empty(1, dtype=torch.int32, device=dev)
        check(lib().ptc_rulebook_blocks(ptr(self.nbr), kv, n, self.bm, self.hcap, ptr(self.tab), ptr(self.hid), ptr(self.hcnt),
                                        ptr(self.n_overflow), stream_ptr()), "ptc_rulebook_blocks")


def block_plan(c_in: int, c_out: int, kv: int, dtype: torch.dtype, n_rows: int = 1 << 30):
    """(bm, hcap) of the block-local tables for this shape, or None when the shape stays on the global-gather kernels.  Two consumers:
    the LDS-staged, register-weight convolution (csrc/conv7.h conv7_supported: 16-bit, 3^3 table, c_in = c_out in {32, 64}, >= 4096
    rows) and the accumulator-stationary weight gradient (csrc/wgrad7.h: the same shapes, plus -- round 4 -- every c_in % 32 == 0,
    c_out % 32 == 0 shape (the 128 .. 512-channel stages, SpUNet's 96-channel decoder) as (32 x 64 | 32)-channel slices; the forward of those shapes takes the tables and
    falls back to the global-gather kernel inside ptc_spconv_fwd_blk)."""
    if dtype == torch.float32 or kv != 27:
        return None
    if c_in == c_out and c_in in (32, 64):
        return (BLOCK_BM, BLOCK_HCAP) if n_rows >= 4096 else None
    if c_in % 32 == 0 and c_out % 32 == 0 and c_in <= 1024 and c_out <= 1024 and (c_in // (64 if c_in % 64 == 0 else 32)) * (c_out // 32) <= 256 \
            and n_rows >= 1024:
        return (BLOCK_BM, BLOCK_HCAP)
    return None


class BlockProvider:
    """lazily built BlockTables of ONE gather table; lives next to the table in the rulebook cache."""

    def __init__(self, nbr: torch.Tensor):
        self.nbr = nbr
        self.tables = {}

    def get(self, c_in: int, c_out: int, dtype: torch.dtype, conv: bool = False):
        """conv=True: asked by a forward / input-gradient convolution -- only the shapes conv7 serves (c_in = c_out in {32, 64}) get
        tables there; the sliced weight gradient of the wider shapes asks with conv=False from the backward, so inference, eval and
        PTC_WGRAD_BLK=0 never pay the table build (one launch + ~112 B per row) for a kernel that cannot use it."""
        plan = block_plan(c_in, c_out, self.nbr.shape[0], dtype, self.nbr.shape[1])
        from . import config

        conv_kernel = (c_in == c_out and c_in in (32, 64)) or (config.CONV8 and c_in >= 96)      # conv7 | conv8 (round 6: c_in % 32 == 0 from 96 up)
        if plan is None or not self.nbr.is_cuda or (conv and not conv_kernel):
            return None
        t = self.tables.get(plan)
        if t is None:
            t = self.tables[plan] = BlockTables(self.nbr, *plan)
        return t


def rulebook_down(indices: torch.Tensor, coord_bits: int, batch_bits: int):
    """k=2,s=2 strided conv maps: out_indices [M,4] i32, nbr_down [8,M] i32, nbr_up [8,N] i32."""
    require_cuda(indices)
    ind = indices.contiguous()
    n = ind.shape[0]
    dev = ind.device
    out_of_in = torch.empty(n, dtype=torch.int32, device=dev)
    n_out_dev = torch.empty(1, dtype=torch.int64, device=dev)
    nbytes = lib().ptc_rulebook_down_workspace_bytes(n)
    ws = _ws(nbytes, dev)
    check(lib().ptc_rulebook_down_count(ptr(ind), n, int(coord_bits), int(batch_bits), ptr(out_of_in), ptr(n_out_dev),
                                        ptr(ws), nbytes, stream_ptr()), "ptc_rulebook_down_count")
    n_out = int(n_out_dev.item())
    out_indices = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
    nbr_down = torch.empty((8, n_out), dtype=torch.int32, device=dev)
    nbr_up = torch.empty((8, n), dtype=torch.int32, device=dev)
    check(lib().ptc_rulebook_down_fill(ptr(ind), n, ptr(out_of_in), n_out, ptr(out_indices), ptr(nbr_down), ptr(nbr_up),
                                       stream_ptr()), "ptc_rulebook_down_fill")
    return out_indices, nbr_down, nbr_up


# ------------------------------------------------------------------------------------------------
# sparse conv compute
# ------------------------------------------------------------------------------------------------
def spconv_fwd(feat: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
               nbr: Optional[torch.Tensor], blk: Optional["BlockTables"] = None) -> torch.Tensor:
    """out[o] = bias + sum_k W[:,k,:] . feat[nbr[k][o]].  weight [C_out, kv, C_in] in feat.dtype,
    bias fp32.  C_in % 8 == 0 and C_out % 16 == 0 (callers pad).  nbr=None (kv == 1): identity
    table, i.e. the dense row-wise GEMM out = feat @ W[:,0,:]^T + bias.  blk = BlockTables of `nbr`: the
    LDS-staged kernel (same result up to the fp32 rounding of its summation order)."""
    require_cuda(feat, weight, bias, nbr)
    feat = feat.contiguous()
    weight = weight.contiguous()
    if weight.dtype != feat.dtype:
        raise PtcoreError("weight dtype must match feature dtype")
    c_out, kv, c_in = weight.shape
    if nbr is not None:
        nbr = nbr.contiguous()
        if nbr.shape[0] != kv or nbr.dtype != torch.int32:
            raise PtcoreError(f"table mismatch: weight {tuple(weight.shape)} nbr {tuple(nbr.shape)} {nbr.dtype}")
    elif kv != 1:
        raise PtcoreError("nbr=None needs kv == 1")
    if feat.shape[1] != c_in:
        raise PtcoreError(f"shape mismatch: feat {tuple(feat.shape)} weight {tuple(weight.shape)}")
    if bias is not None:
        bias = bias.to(torch.float32).contiguous()
    n_out = nbr.shape[1] if nbr is not None else feat.shape[0]
    out = torch.empty((n_out, c_out), dtype=feat.dtype, device=feat.device)
    if blk is not None and nbr is not None:
        if blk.nbr.data_ptr() != nbr.data_ptr() or tuple(blk.nbr.shape) != tuple(nbr.shape):
            raise PtcoreError("blk was built from another gather table")
        check(lib().ptc_spconv_fwd_blk(ptr(feat), feat.shape[0], ptr(weight), ptr(bias), ptr(nbr), ptr(blk.tab), ptr(blk.hid),
                                       ptr(blk.hcnt), blk.bm, blk.hcap, n_out, kv, c_in, c_out, dtype_code(feat), ptr(out),
                                       stream_ptr()), "ptc_spconv_fwd_blk")
        return out
    check(lib().ptc_spconv_fwd(ptr(feat), feat.shape[0], ptr(weight), ptr(bias), ptr(nbr), n_out, kv, c_in, c_out,
                               dtype_code(feat), ptr(out), stream_ptr()), "ptc_spconv_fwd")
    return out


def spconv_wgrad(feat: torch.Tensor, dout: torch.Tensor, nbr: Optional[torch.Tensor], want_bias: bool = False,
                 blk: Optional["BlockTables"] = None):
    """dw [C_out, kv, C_in] fp32 = sum_o dout[o]^T (x) feat[nbr[k][o]]  (nbr=None: identity, kv=1);
    with want_bias also dbias [C_out] fp32 = column sums of dout (fused).  Returns dw or (dw, dbias).  blk = BlockTables of `nbr`
    (and no bias gradient): the block-staged, accumulator-stationary kernel (csrc/wgrad7.h; same result up to the fp32 rounding of
    its summation order)."""
    require_cuda(feat, dout, nbr)
    feat = feat.contiguous()
    dout = dout.contiguous()
    if feat.dtype != dout.dtype:
        raise PtcoreError("feat / dout dtype mismatch")
    if nbr is not None:
        nbr = nbr.contiguous()
        kv, n_out = nbr.shape
    else:
        kv, n_out = 1, dout.shape[0]
    c_in, c_out = feat.shape[1], dout.shape[1]
    dw = torch.empty((c_out, kv, c_in), dtype=torch.float32, device=feat.device)
    if blk is not None and nbr is not None and not want_bias:
        if blk.nbr.data_ptr() != nbr.data_ptr() or tuple(blk.nbr.shape) != tuple(nbr.shape):
            raise PtcoreError("blk was built from another gather table")
        nbytes = lib().ptc_spconv_wgrad_blk_workspace_bytes(n_out, kv, c_in, c_out)
        ws = _ws(nbytes, feat.device)
        check(lib().ptc_spconv_wgrad_blk(ptr(feat), feat.shape[0], ptr(dout), ptr(nbr), ptr(blk.tab), ptr(blk.hid), ptr(blk.hcnt),
                                         ptr(blk.n_overflow), blk.bm, blk.hcap, n_out, kv, c_in, c_out, dtype_code(feat), ptr(dw), ptr(ws),
                                         nbytes, stream_ptr()), "ptc_spconv_wgrad_blk")
        return dw
    db = torch.empty(c_out, dtype=torch.float32, device=feat.device) if want_bias else None
    nbytes = lib().ptc_spconv_wgrad_workspace_bytes(n_out, kv, c_in, c_out)
    ws = _ws(nbytes, feat.device)
    check(lib().ptc_spconv_wgrad(ptr(feat), feat.shape[0], ptr(dout), ptr(nbr), n_out, kv, c_in, c_out,
                                 dtype_code(feat), ptr(dw), ptr(db), ptr(ws), nbytes, stream_ptr()), "ptc_spconv_wgrad")
    return (dw, db) if want_bias else dw


# ------------------------------------------------------------------------------------------------
# evaluation tail: arg-max + inverse gather + three class histograms in one pass
# ------------------------------------------------------------------------------------------------
def seg_eval_hist(logits: Optional[torch.Tensor], target: torch.Tensor, k: int, ignore_index: int = -1,
                  inverse: Optional[torch.Tensor] = None, pred: Optional[torch.Tensor] = None):
    """(area_intersection, area_union, area_target), each int64 [k]: pointcept/utils/misc.py:57-69 applied to
    pred = logits.max(1)[1][inverse] (pointcept/engines/hooks/evaluator.py:139-147), or to given predictions."""
    require_cuda(logits, target, inverse, pred)
    if (logits is None) == (pred is None):
        raise PtcoreError("give logits or pred")
    target = target.reshape(-1).to(torch.int64).contiguous()
    m = target.numel()
    hist = torch.empty((3, int(k)), dtype=torch.int64, device=target.device)
    if pred is not None:
        pred = pred.reshape(-1).to(torch.int64).contiguous()
        if pred.numel() != m:
            raise PtcoreError("pred / target size mismatch")            # misc.py:60
        check(lib().ptc_seg_eval_hist(0, 0, 0, 0, ptr(pred), 0, ptr(target), m, 0, int(k), int(ignore_index), ptr(hist), stream_ptr()),
              "ptc_seg_eval_hist")
    else:
        if logits.dim() != 2 or logits.stride(1) != 1:
            raise PtcoreError("logits must be [N, C] with unit column stride")
        if inverse is not None:
            inverse = inverse.reshape(-1).to(torch.int64).contiguous()
            if inverse.numel() != m:
                raise PtcoreError("inverse / target size mismatch")
        elif logits.shape[0] != m:
            raise PtcoreError("logits / target size mismatch")
        check(lib().ptc_seg_eval_hist(ptr(logits), dtype_code(logits), logits.stride(0), logits.shape[1], 0, ptr(inverse), ptr(target), m,
                                      logits.shape[0], int(k), int(ignore_index), ptr(hist), stream_ptr()), "ptc_seg_eval_hist")
    return hist[0], hist[1] + hist[2] - hist[0], hist[2]


# ------------------------------------------------------------------------------------------------
# 3-axis rotary embedding (libs/pointrope)
# ------------------------------------------------------------------------------------------------
def rope3d_(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> torch.Tensor:
    """in place on tokens [..., H, D] (contiguous, D % 6 == 0) with positions [..., 3] int64 (libs/pointrope/kernels.cu:19-100)"""
    require_cuda(tokens, positions)
    if tokens.dim() < 3 or not tokens.is_contiguous():
        raise PtcoreError("tokens must be contiguous [..., H, D]")       # kernels.cu:83: "tokens are not contiguous"
    if positions.dtype != torch.int64 or not positions.is_contiguous() or positions.shape[-1] != 3:
        raise PtcoreError("positions must be contiguous int64 [..., 3]")
    H, D = tokens.shape[-2], tokens.shape[-1]
    n = tokens.numel() // (H * D) if H * D > 0 else 0
    if positions.numel() != 3 * n:
        raise PtcoreError(f"positions {tuple(positions.shape)} do not match tokens {tuple(tokens.shape)}")
    check(lib().ptc_rope3d(ptr(tokens), dtype_code(tokens), ptr(positions), n, H, D, float(base), float(fwd), stream_ptr()), "ptc_rope3d")
    return tokens


def rope3d_xyz(qkv: torch.Tensor, xyz: torch.Tensor, inv_freq: torch.Tensor, rot_slabs: int, sign: float,
               out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """PT-v3m3 Point3DRoPE on packed rows: qkv [n, S, H, D] -> same shape in `out_dtype` (default: qkv's), the first `rot_slabs`
    slabs rotated with xyz [n, 3] fp32 and inv_freq [D/6] fp32, the rest converted (point_transformer_v3m3_utonia.py:43-102,274-323).
    sign = +1 forward / -1 gradient."""
    require_cuda(qkv, xyz, inv_freq)
    if qkv.dim() != 4 or not qkv.is_contiguous():
        raise PtcoreError("rope3d_xyz: qkv must be contiguous [n, slabs, H, D]")
    n, S, H, D = qkv.shape
    if D % 6 != 0:
        raise PtcoreError(f"rope3d_xyz: head dim {D} must be a multiple of 6")
    if xyz.dtype != torch.float32 or tuple(xyz.shape) != (n, 3) or not xyz.is_contiguous():
        raise PtcoreError("rope3d_xyz: xyz must be contiguous fp32 [n, 3]")
    if inv_freq.dtype != torch.float32 or inv_freq.numel() != D // 6 or not inv_freq.is_contiguous():
        raise PtcoreError(f"rope3d_xyz: inv_freq must be contiguous fp32 [{D // 6}]")
    out = torch.empty(qkv.shape, dtype=out_dtype or qkv.dtype, device=qkv.device)
    check(lib().ptc_rope3d_xyz(ptr(qkv), dtype_code(qkv), ptr(out), dtype_code(out), ptr(xyz), ptr(inv_freq), n, S, int(rot_slabs), H, D,
                               float(sign), stream_ptr()), "ptc_rope3d_xyz")
    return out


# ------------------------------------------------------------------------------------------------
# layer norm
# ------------------------------------------------------------------------------------------------
_DT = {torch.float32: _lib.PTC_F32, torch.float16: _lib.PTC_F16, torch.bfloat16: _lib.PTC_BF16}


def layer_norm_supported(c: int) -> bool:
    """the power-of-two instances (C = 32 .. 512): the widths the GEMM-epilogue joints and the Block executor are built for"""
    return lib().ptc_layer_norm_supported(int(c)) == 1


def layer_norm_joint_available(c: int) -> bool:
    """the fused residual joints (add_norm_fwd / _bwd) take this width: every width LayerNorm takes (round 5: generic even widths too)"""
    return lib().ptc_layer_norm_supported(int(c)) != 0


def layer_norm_available(c: int) -> bool:
    """layer_norm_fwd / _bwd take this width (the instances above, or the wave-per-row form: any even C <= 1024)"""
    return lib().ptc_layer_norm_supported(int(c)) != 0


def layer_norm_fwd(x: torch.Tensor, gamma, beta, eps: float, out_dtype: torch.dtype):
    """y [N,C] (out_dtype), mean [N], rstd [N] (fp32).  gamma / beta fp32 or None."""
    require_cuda(x, gamma, beta)
    x = x.contiguous()
    n, c = x.shape
    g = None if gamma is None else gamma.to(torch.float32).contiguous()
    b = None if beta is None else beta.to(torch.float32).contiguous()
    y = torch.empty((n, c), dtype=out_dtype, device=x.device)
    mean = torch.empty(n, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n, dtype=torch.float32, device=x.device)
    check(lib().ptc_layer_norm_fwd(ptr(x), n, c, dtype_code(x), ptr(g), ptr(b), float(eps), ptr(y), _DT[out_dtype],
                                   ptr(mean), ptr(rstd), stream_ptr()), "ptc_layer_norm_fwd")
    return y, mean, rstd


def layer_norm_bwd(dy: torch.Tensor, x: torch.Tensor, mean, rstd, gamma, want_affine: bool = True):
    """dx (x.dtype), dgamma, dbeta (fp32 or None)."""
    require_cuda(dy, x, mean, rstd, gamma)
    dy = dy.contiguous()
    x = x.contiguous()
    n, c = x.shape
    g = None if gamma is None else gamma.to(torch.float32).contiguous()
    dx = torch.empty_like(x)
    dg = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
    db = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
    nbytes = lib().ptc_layer_norm_bwd_workspace_bytes(n, c)
    ws = _ws(nbytes, x.device)
    check(lib().ptc_layer_norm_bwd(ptr(dy), dtype_code(dy), ptr(x), dtype_code(x), ptr(mean), ptr(rstd), ptr(g), n, c,
                                   ptr(dx), ptr(dg), ptr(db), ptr(ws), nbytes, stream_ptr()), "ptc_layer_norm_bwd")
    return dx, dg, db


def add_norm_fwd(u, a, row_scale, norm_a, norm_b, y_dtype):
    """z = a + row_scale * f(u) (fp32), y = g(z) (y_dtype or None).  norm_a / norm_b: (gamma, beta, eps) or
    None (identity).  Returns (z, y, statA, statB)."""
    require_cuda(u, a, row_scale)
    u = u.contiguous()
    a = a.contiguous()
    if a.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise PtcoreError("add_norm: the residual operand `a` must be fp32, bf16 or f16")
    n, c = u.shape
    dev = u.device
    z = torch.empty((n, c), dtype=torch.float32, device=dev)
    y = torch.empty((n, c), dtype=y_dtype, device=dev) if y_dtype is not None else None
    st_a = torch.empty((2, n), dtype=torch.float32, device=dev) if norm_a is not None else None
    st_b = torch.empty((2, n), dtype=torch.float32, device=dev) if (norm_b is not None and y is not None) else None
    ga, ba, ea = norm_a if norm_a is not None else (None, None, 0.0)
    gb, bb, eb = norm_b if norm_b is not None else (None, None, 0.0)
    rs = None if row_scale is None else row_scale.to(torch.float32).contiguous()
    check(lib().ptc_add_norm_fwd(ptr(u), dtype_code(u), ptr(a), dtype_code(a), ptr(rs), n, c, ptr(ga), ptr(ba), float(ea),
                                 int(norm_a is not None), ptr(gb), ptr(bb), float(eb), int(norm_b is not None), ptr(z), ptr(y),
                                 _DT[y_dtype] if y_dtype is not None else 0, ptr(st_a), ptr(st_b), stream_ptr()),
          "ptc_add_norm_fwd")
    return z, y, st_a, st_b


def linear_joint_supported(c_in: int, c_out: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and bool(lib().ptc_linear_joint_supported(int(c_in), int(c_out), _DT[dtype]))


def linear_joint_fwd(x, weight, bias, table, a, row_scale, norm_b, y_dtype, n_out=None):
    """z = a + row_scale * (x[table] @ weight^T + bias) (fp32), y = LN_B(z) | cast(z) -- a Linear of the Block with the residual joint behind
    it in its epilogue (csrc/fwd2_joint.h; bit-identical to spconv_fwd + add_norm_fwd).  weight [c_out, c_in] in x's 16-bit dtype, table
    None or the kv = 1 gather table [1, n_out] / [n_out] int32 (the inverse serialization table of `proj`), a [n_out, c_out] fp32.
    Returns (z, y, statB)."""
    require_cuda(x, weight, bias, table, a, row_scale)
    x, weight, a = x.contiguous(), weight.contiguous(), a.contiguous()
    c_out, c_in = weight.shape[0], weight.shape[-1]
    if a.dtype != torch.float32 or x.dtype != weight.dtype or not linear_joint_supported(c_in, c_out, x.dtype):
        raise PtcoreError("linear_joint_fwd: unsupported shape / dtype")
    n = int(a.shape[0]) if n_out is None else int(n_out)
    tab = None if table is None else table.reshape(-1).to(torch.int32).contiguous()
    dev = x.device
    z = torch.empty((n, c_out), dtype=torch.float32, device=dev)
    y = torch.empty((n, c_out), dtype=y_dtype, device=dev) if y_dtype is not None else None
    st_b = torch.empty((2, n), dtype=torch.float32, device=dev) if (norm_b is not None and y is not None) else None
    gb, bb, eb = norm_b if norm_b is not None else (None, None, 0.0)
    rs = None if row_scale is None else row_scale.to(torch.float32).contiguous()
    b = None if bias is None else bias.float().contiguous()
    check(lib().ptc_linear_joint_fwd(ptr(x), x.shape[0], ptr(weight), ptr(b), ptr(tab), n, c_in, c_out, dtype_code(x), ptr(a), ptr(rs), ptr(gb), ptr(bb),
                                     float(eb), int(norm_b is not None), ptr(z), ptr(y), ptr(st_b), stream_ptr()), "ptc_linear_joint_fwd")
    return z, y, st_b


def linear_norm_joint_fwd(x, weight, bias, norm_a, a, norm_b, y_dtype):
    """u = x @ weight^T + bias (x's dtype), z = a + LN_A(u) (fp32), y = LN_B(z) | cast(z): the Linear of the positional encoding with its
    joint in the epilogue (ptv3m1:285, 318-323; bit-identical to spconv_fwd + add_norm_fwd with normA).  a fp32 or x's dtype.
    Returns (u, z, y, statA, statB)."""
    require_cuda(x, weight, bias, a)
    x, weight, a = x.contiguous(), weight.contiguous(), a.contiguous()
    c_out, c_in = weight.shape[0], weight.shape[-1]
    if x.dtype != weight.dtype or c_in != c_out or not linear_joint_supported(c_in, c_out, x.dtype) or a.dtype not in (torch.float32, x.dtype):
        raise PtcoreError("linear_norm_joint_fwd: unsupported shape / dtype")
    n, dev = int(a.shape[0]), x.device
    u = torch.empty((n, c_out), dtype=x.dtype, device=dev)
    z = torch.empty((n, c_out), dtype=torch.float32, device=dev)
    y = torch.empty((n, c_out), dtype=y_dtype, device=dev) if y_dtype is not None else None
    st_a = torch.empty((2, n), dtype=torch.float32, device=dev)
    st_b = torch.empty((2, n), dtype=torch.float32, device=dev) if (norm_b is not None and y is not None) else None
    ga, ba, ea = norm_a
    gb, bb, eb = norm_b if norm_b is not None else (None, None, 0.0)
    b = None if bias is None else bias.float().contiguous()
    check(lib().ptc_linear_norm_joint_fwd(ptr(x), x.shape[0], ptr(weight), ptr(b), n, c_in, c_out, dtype_code(x), ptr(ga), ptr(ba), float(ea), ptr(a),
                                          dtype_code(a), ptr(gb), ptr(bb), float(eb), int(norm_b is not None), ptr(u), ptr(z), ptr(y), ptr(st_a), ptr(st_b),
                                          stream_ptr()), "ptc_linear_norm_joint_fwd")
    return u, z, y, st_a, st_b


def add_norm_bwd(dz_in, dy, z, u, row_scale, g_a, st_a, g_b, st_b, want_affine_a: bool, want_affine_b: bool,
                 da_dtype: torch.dtype = torch.float32):
    """-> (da (da_dtype: the dtype of the forward's `a`), du (u.dtype), dgA, dbA, dgB, dbB)"""
    require_cuda(dz_in, dy, z, u, row_scale)
    n, c = u.shape
    dev = u.device
    dz_in = None if dz_in is None else dz_in.to(torch.float32).contiguous()
    dy = None if dy is None else dy.contiguous()
    da = torch.empty((n, c), dtype=da_dtype, device=dev)
    du = torch.empty_like(u)
    mk = lambda want: torch.empty(c, dtype=torch.float32, device=dev) if want else None  # noqa: E731
    dga, dba, dgb, dbb = mk(want_affine_a), mk(want_affine_a), mk(want_affine_b), mk(want_affine_b)
    nbytes = lib().ptc_add_norm_bwd_workspace_bytes(n, c)
    ws = _ws(nbytes, dev)
    check(lib().ptc_add_norm_bwd(ptr(dz_in), ptr(dy), dtype_code(dy) if dy is not None else 0, ptr(z), ptr(u), dtype_code(u),
                                 ptr(row_scale), n, c, ptr(g_a), ptr(st_a), int(st_a is not None), ptr(g_b), ptr(st_b),
                                 int(st_b is not None), ptr(da), dtype_code(da), ptr(du), ptr(dga), ptr(dba), ptr(dgb), ptr(dbb), ptr(ws),
                                 nbytes, stream_ptr()), "ptc_add_norm_bwd")
    return da, du, dga, dba, dgb, dbb


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def attn_hd_supported(head_dim: int, max_seqlen: int) -> bool:
    """True when (head_dim, max_seqlen) is served by the MFMA window-attention kernels: head_dim 16 up to 1024 keys,
    17..32 up to 1024, ..48 up to 672, ..64 up to 512 (the window's operands stay in LDS)."""
    if head_dim == 16:
        return 1 <= int(max_seqlen) <= 1024
    return bool(lib().ptc_attn_varlen_hd_supported(int(head_dim), int(max_seqlen)))


def attn_varlen_fwd(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, softmax_scale: float, dropout_p: float = 0.0, seed: int = 0):
    """qkv [T,3,H,D] bf16 -> (out [T,H,D] bf16, lse [H,T] fp32); D = 16 (attention.hip) or 17..64 (attention_hd.h).
    D = 16 also takes f16 qkv: the SAME bf16 arithmetic with the reference's qkv.to(bfloat16) / feat.to(qkv.dtype) casts
    (ptv3m1:209,215) done in the kernel's load / store paths -- out comes back f16, bit for bit what the two cast passes produce.
    D = 17..64 with f16 qkv: f16 operands (f16 MFMAs, P rounded to f16), LitePT's call site."""
    require_cuda(qkv, cu_seqlens)
    if qkv.dtype not in (torch.bfloat16, torch.float16) or qkv.dim() != 4 or qkv.shape[1] != 3:
        raise PtcoreError(f"qkv must be bf16 or f16 [T,3,H,D], got {qkv.dtype} {tuple(qkv.shape)}")
    qkv = qkv.contiguous()
    cu = cu_seqlens.to(torch.int32).contiguous()
    T, _, H, D = qkv.shape
    out = torch.empty((T, H, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((H, T), dtype=torch.float32, device=qkv.device)
    if dropout_p > 0.0:       # attention dropout (csrc/attention_drop.h): head_dim 16; the mask is a function of `seed`, regenerated by the backward
        if D != 16:
            raise PtcoreError("attention dropout is implemented for head_dim 16")
        check(lib().ptc_attn_varlen_dropout_fwd(ptr(qkv), ptr(cu), cu.numel() - 1, T, H, int(max_seqlen), float(softmax_scale), dtype_code(qkv),
                                                float(dropout_p), int(seed), ptr(out), ptr(lse), stream_ptr()), "ptc_attn_varlen_dropout_fwd")
    elif D == 16:
        check(lib().ptc_attn_varlen_fwd(ptr(qkv), ptr(cu), cu.numel() - 1, T, H, int(max_seqlen), float(softmax_scale),
                                        dtype_code(qkv), ptr(out), ptr(lse), stream_ptr()), "ptc_attn_varlen_fwd")
    else:
        check(lib().ptc_attn_varlen_hd_fwd(ptr(qkv), ptr(cu), cu.numel() - 1, T, H, D, int(max_seqlen), float(softmax_scale),
                                           dtype_code(qkv), ptr(out), ptr(lse), stream_ptr()), "ptc_attn_varlen_hd_fwd")
    return out, lse


def attn_varlen_bwd(qkv, out, dout, lse, cu_seqlens, max_seqlen: int, softmax_scale: float, dropout_p: float = 0.0, seed: int = 0) -> torch.Tensor:
    require_cuda(qkv, out, dout, lse, cu_seqlens)
    qkv = qkv.contiguous()
    out = out.contiguous()
    dout = dout.to(qkv.dtype).contiguous()       # (f16 qkv, D = 16: f16 out / dout / dqkv, the casts live in the kernels, see attn_varlen_fwd)
    cu = cu_seqlens.to(torch.int32).contiguous()
    T, _, H, D = qkv.shape
    dqkv = torch.empty_like(qkv)
    nbytes = lib().ptc_attn_varlen_bwd_workspace_bytes(T, H)
    ws = _ws(nbytes, qkv.device)
    if dropout_p > 0.0:
        check(lib().ptc_attn_varlen_dropout_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(cu), cu.numel() - 1, T, H, int(max_seqlen),
                                                float(softmax_scale), dtype_code(qkv), float(dropout_p), int(seed), ptr(dqkv), ptr(ws), nbytes,
                                                stream_ptr()), "ptc_attn_varlen_dropout_bwd")
    elif D == 16:
        check(lib().ptc_attn_varlen_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(cu), cu.numel() - 1, T, H,
                                        int(max_seqlen), float(softmax_scale), dtype_code(qkv), ptr(dqkv), ptr(ws), nbytes,
                                        stream_ptr()), "ptc_attn_varlen_bwd")
    else:
        check(lib().ptc_attn_varlen_hd_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(cu), cu.numel() - 1, T, H, D,
                                           int(max_seqlen), float(softmax_scale), dtype_code(qkv), ptr(dqkv), ptr(ws), nbytes,
                                           stream_ptr()), "ptc_attn_varlen_hd_bwd")
    return dqkv


def attn_rope_supported(head_dim: int, max_seqlen: int) -> bool:
    """window attention with the 3-D rotary embedding fused into its prologue / epilogue (csrc/attention_hd.h, ROPE): head_dim 18"""
    return bool(lib().ptc_attn_varlen_hd_rope_supported(int(head_dim), int(max_seqlen)))


def attn_rope_fwd(qkv, xyz, inv_freq, cu_seqlens, max_seqlen: int, softmax_scale: float):
    """qkv [T,3,H,18] bf16 | f16, UN-rotated; xyz [T,3] fp32 positions of the (padded, serialized) rows; inv_freq [3] fp32
    -> (out [T,H,18], lse [H,T] fp32) of the attention over the rotated q / k."""
    require_cuda(qkv, xyz, inv_freq, cu_seqlens)
    if qkv.dtype not in (torch.bfloat16, torch.float16) or qkv.dim() != 4 or qkv.shape[1] != 3:
        raise PtcoreError(f"qkv must be bf16 or f16 [T,3,H,18], got {qkv.dtype} {tuple(qkv.shape)}")
    qkv, xyz, inv_freq = qkv.contiguous(), xyz.float().contiguous(), inv_freq.float().contiguous()
    T, _, H, D = qkv.shape
    if xyz.shape != (T, 3) or inv_freq.numel() * 6 != D:
        raise PtcoreError("attn_rope_fwd: xyz must be [T, 3] and inv_freq [head_dim / 6]")
    cu = cu_seqlens.to(torch.int32).contiguous()
    out = torch.empty((T, H, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((H, T), dtype=torch.float32, device=qkv.device)
    check(lib().ptc_attn_varlen_hd_rope_fwd(ptr(qkv), ptr(cu), ptr(xyz), ptr(inv_freq), cu.numel() - 1, T, H, D, int(max_seqlen),
                                            float(softmax_scale), dtype_code(qkv), ptr(out), ptr(lse), stream_ptr()), "ptc_attn_varlen_hd_rope_fwd")
    return out, lse


def attn_rope_bwd(qkv, out, dout, lse, xyz, inv_freq, cu_seqlens, max_seqlen: int, softmax_scale: float) -> torch.Tensor:
    """-> dqkv: gradient of the UN-rotated rows (the inverse rotation of dq / dk happens in the kernels' epilogue)"""
    require_cuda(qkv, out, dout, lse, xyz, inv_freq, cu_seqlens)
    qkv, out = qkv.contiguous(), out.contiguous()
    dout = dout.to(qkv.dtype).contiguous()
    xyz, inv_freq = xyz.float().contiguous(), inv_freq.float().contiguous()
    cu = cu_seqlens.to(torch.int32).contiguous()
    T, _, H, D = qkv.shape
    dqkv = torch.empty_like(qkv)
    nbytes = lib().ptc_attn_varlen_bwd_workspace_bytes(T, H)
    ws = _ws(nbytes, qkv.device)
    check(lib().ptc_attn_varlen_hd_rope_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(cu), ptr(xyz), ptr(inv_freq), cu.numel() - 1, T, H, D,
                                            int(max_seqlen), float(softmax_scale), dtype_code(qkv), ptr(dqkv), ptr(ws), nbytes, stream_ptr()),
          "ptc_attn_varlen_hd_rope_bwd")
    return dqkv


def attn_rpe_supported(head_dim: int, max_seqlen: int, pos_bnd: int) -> bool:
    """RPE attention kernels (attention_rpe.h): head_dim 16, windows whose images + coordinates + table fit LDS."""
    lp = (int(max_seqlen) + 31) & ~31
    r = 2 * int(pos_bnd) + 1
    return head_dim == 16 and 1 <= max_seqlen <= 1024 and lp * 72 + lp * 8 + ((3 * r + 3) & ~3) * 8 <= 163840


def attn_rpe_fwd(qkv, cu_seqlens, max_seqlen: int, softmax_scale: float, grid_coord, rpe_table, pos_bnd: int):
    """qkv [T,3,H,16] bf16 | f16, grid_coord [T,3] int32 (same row order), rpe_table [3(2B+1),H] fp32 -> (out [T,H,16] like qkv, lse [H,T]).
    f16 tensors (the reference's fp16 AMP) ride around the same bf16 arithmetic: the casts sit in the kernels' load / store paths."""
    require_cuda(qkv, cu_seqlens, grid_coord, rpe_table)
    if qkv.dtype not in (torch.bfloat16, torch.float16) or qkv.dim() != 4 or qkv.shape[1] != 3 or qkv.shape[3] != 16:
        raise PtcoreError(f"qkv must be bf16 / f16 [T,3,H,16], got {qkv.dtype} {tuple(qkv.shape)}")
    T, _, H, _ = qkv.shape
    if grid_coord.dtype != torch.int32 or tuple(grid_coord.shape) != (T, 3):
        raise PtcoreError(f"grid_coord must be int32 [{T},3], got {grid_coord.dtype} {tuple(grid_coord.shape)}")
    if rpe_table.dtype != torch.float32 or tuple(rpe_table.shape) != (3 * (2 * int(pos_bnd) + 1), H):
        raise PtcoreError(f"rpe_table must be fp32 [{3 * (2 * int(pos_bnd) + 1)},{H}], got {rpe_table.dtype} {tuple(rpe_table.shape)}")
    qkv, gc, tab = qkv.contiguous(), grid_coord.contiguous(), rpe_table.contiguous()
    cu = cu_seqlens.to(torch.int32).contiguous()
    out = torch.empty((T, H, 16), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((H, T), dtype=torch.float32, device=qkv.device)
    check(lib().ptc_attn_rpe_fwd(ptr(qkv), ptr(cu), ptr(gc), ptr(tab), int(pos_bnd), cu.numel() - 1, T, H, int(max_seqlen),
                                 float(softmax_scale), dtype_code(qkv), ptr(out), ptr(lse), stream_ptr()), "ptc_attn_rpe_fwd")
    return out, lse


def attn_rpe_bwd(qkv, out, dout, lse, cu_seqlens, max_seqlen: int, softmax_scale: float, grid_coord, rpe_table, pos_bnd: int):
    """-> (dqkv like qkv, d_rpe_table fp32 like rpe_table)"""
    require_cuda(qkv, out, dout, lse, cu_seqlens, grid_coord, rpe_table)
    qkv, out = qkv.contiguous(), out.contiguous()
    dout = dout.to(qkv.dtype).contiguous()
    gc, tab = grid_coord.contiguous(), rpe_table.contiguous()
    cu = cu_seqlens.to(torch.int32).contiguous()
    T, _, H, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    dtab = torch.empty_like(tab)
    nbytes = lib().ptc_attn_rpe_bwd_workspace_bytes(T, H, int(pos_bnd))
    ws = _ws(nbytes, qkv.device)
    check(lib().ptc_attn_rpe_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(cu), ptr(gc), ptr(tab), int(pos_bnd), cu.numel() - 1,
                                 T, H, int(max_seqlen), float(softmax_scale), dtype_code(qkv), ptr(dqkv), ptr(dtab), ptr(ws), nbytes,
                                 stream_ptr()), "ptc_attn_rpe_bwd")
    return dqkv, dtab


# ------------------------------------------------------------------------------------------------
# ends of the step: coordinate maxima, cross-entropy
# ------------------------------------------------------------------------------------------------
def coord_max(grid_coord: torch.Tensor) -> torch.Tensor:
    """max over points of grid_coord per axis -> int64 [3] on the device (structure.py:74,136-138)."""
    require_cuda(grid_coord)
    if grid_coord.dtype not in (torch.int64, torch.int32) or grid_coord.dim() != 2 or grid_coord.shape[1] != 3:
        raise PtcoreError("grid_coord must be int32/int64 [N,3]")
    gc = grid_coord.contiguous()
    out = torch.empty(3, dtype=torch.int64, device=gc.device)
    check(lib().ptc_coord_max(ptr(gc), int(gc.dtype == torch.int64), gc.shape[0], ptr(out), stream_ptr()), "ptc_coord_max")
    return out


def _rows_view(t: torch.Tensor):
    """[N, C] tensor whose rows are contiguous (any row stride) -> (tensor, row_stride)"""
    if t.dim() != 2:
        raise PtcoreError("expected [N, C]")
    if t.shape[0] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else t.shape[1])


def cross_entropy_fwd(logits: torch.Tensor, target: torch.Tensor, ignore_index: int):
    """-> (loss_sum [] fp32, count [] fp32, lse [N] fp32); loss = loss_sum / count (CrossEntropyLoss, mean)."""
    require_cuda(logits, target)
    if target.dtype != torch.int64:
        raise PtcoreError("target must be int64")
    lg, rs = _rows_view(logits)
    n, c = lg.shape
    nb = lib().ptc_cross_entropy_partials(n)
    lse = torch.empty(n, dtype=torch.float32, device=lg.device)
    partial = torch.empty((nb, 2), dtype=torch.float32, device=lg.device)
    check(lib().ptc_cross_entropy_fwd(ptr(lg), rs, ptr(target.contiguous()), n, c, dtype_code(lg), int(ignore_index), ptr(lse),
                                      ptr(partial), stream_ptr()), "ptc_cross_entropy_fwd")
    tot = partial.sum(0)   # fixed-order tree reduction of <= N/256 partials: deterministic
    return tot[0], tot[1], lse


def cross_entropy_bwd(logits: torch.Tensor, target: torch.Tensor, lse: torch.Tensor, scale: torch.Tensor, ignore_index: int):
    """dlogits [N, C] (logits' dtype) = scale * (softmax - onehot) on counted rows; `scale` device scalar fp32."""
    require_cuda(logits, target, lse, scale)
    lg, rs = _rows_view(logits)
    n, c = lg.shape
    out = torch.empty((n, c), dtype=lg.dtype, device=lg.device)
    check(lib().ptc_cross_entropy_bwd(ptr(lg), rs, ptr(target.contiguous()), ptr(lse), ptr(scale.reshape(1).float().contiguous()),
                                      n, c, dtype_code(lg), int(ignore_index), ptr(out), c, stream_ptr()), "ptc_cross_entropy_bwd")
    return out


def lovasz_softmax(logits: torch.Tensor, target: torch.Tensor, ignore_index: int):
    """Lovasz-Softmax (multiclass, classes present, whole batch: pointcept/models/losses/lovasz.py:118-146) ->
    (loss [] fp32, dlogits [N, C] fp32): one segmented radix sort of the [C, N] error matrix."""
    require_cuda(logits, target)
    if target.dtype != torch.int64:
        raise PtcoreError("target must be int64")
    lg, rs = _rows_view(logits)
    n, c = lg.shape
    nbytes = lib().ptc_lovasz_softmax_workspace_bytes(n, c)
    if nbytes == 0:
        raise PtcoreError(f"lovasz_softmax: unsupported shape [{n}, {c}] (at most 64 classes)")
    ws = _ws(nbytes, lg.device)
    loss = torch.empty((), dtype=torch.float32, device=lg.device)
    dlogits = torch.empty((n, c), dtype=torch.float32, device=lg.device)
    check(lib().ptc_lovasz_softmax(ptr(lg), rs, ptr(target.contiguous()), n, c, dtype_code(lg), int(ignore_index), ptr(loss),
                                   ptr(dlogits), ptr(ws), nbytes, stream_ptr()), "ptc_lovasz_softmax")
    return loss, dlogits


# ------------------------------------------------------------------------------------------------
# BatchNorm1d + activation
# ------------------------------------------------------------------------------------------------
ACT_CODES = {"none": 0, "gelu": 1, "relu": 2}


def batch_norm_supported(c: int, dtype: torch.dtype) -> bool:
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return False
    code = {torch.float32: _lib.PTC_F32, torch.float16: _lib.PTC_F16, torch.bfloat16: _lib.PTC_BF16}[dtype]
    return bool(lib().ptc_batch_norm_supported(int(c), code))


def batch_norm_act_fwd(x, gamma, beta, running_mean, running_var, training: bool, momentum: float, eps: float, act: str,
                       out_dtype: Optional[torch.dtype] = None, res: Optional[torch.Tensor] = None):
    """-> (y [N,C] out_dtype, save_mean [C] f32, save_rstd [C] f32); running statistics updated in place.  res [N,C] (x's dtype):
    y = act(BN(x) + res), the tail of a residual block in the same apply pass (ptc_batch_norm_add_act_fwd)."""
    require_cuda(x, gamma, beta, running_mean, running_var, res)
    x = x.contiguous()
    if res is not None:
        if res.shape != x.shape or res.dtype != x.dtype:
            raise PtcoreError("batch_norm_act_fwd: the residual must have x's shape and dtype")
        res = res.contiguous()
    n, c = x.shape
    out_dtype = out_dtype or x.dtype
    y = torch.empty((n, c), dtype=out_dtype, device=x.device)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = lib().ptc_batch_norm_workspace_bytes(n, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    g = None if gamma is None else gamma.float().contiguous()
    b = None if beta is None else beta.float().contiguous()
    for t in (running_mean, running_var):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise PtcoreError("running statistics must be contiguous fp32")
    if res is not None:
        check(lib().ptc_batch_norm_add_act_fwd(ptr(x), ptr(res), n, c, dtype_code(x), ptr(g), ptr(b), float(eps), float(momentum),
                                               int(bool(training)), ptr(running_mean), ptr(running_var), ACT_CODES[act], ptr(y),
                                               dtype_code(y), ptr(mean), ptr(rstd), ptr(ws), nbytes, stream_ptr()),
              "ptc_batch_norm_add_act_fwd")
        return y, mean, rstd
    check(lib().ptc_batch_norm_act_fwd(ptr(x), n, c, dtype_code(x), ptr(g), ptr(b), float(eps), float(momentum), int(bool(training)),
                                       ptr(running_mean), ptr(running_var), ACT_CODES[act], ptr(y), dtype_code(y), ptr(mean),
                                       ptr(rstd), ptr(ws), nbytes, stream_ptr()), "ptc_batch_norm_act_fwd")
    return y, mean, rstd


def batch_norm_act_bwd(dy, x, gamma, beta, mean, rstd, training: bool, act: str, want_affine: bool = True, res: Optional[torch.Tensor] = None):
    """-> (dx [N,C] x.dtype, dgamma [C] f32 | None, dbeta [C] f32 | None); with res (the forward's residual): a fourth result dres [N,C]"""
    require_cuda(dy, x, mean, rstd, res)
    dy, x = dy.contiguous(), x.contiguous()
    n, c = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
    db = torch.empty(c, dtype=torch.float32, device=x.device) if want_affine else None
    nbytes = lib().ptc_batch_norm_workspace_bytes(n, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    g = None if gamma is None else gamma.float().contiguous()
    b = None if beta is None else beta.float().contiguous()
    if res is not None:
        res = res.contiguous()
        dres = torch.empty_like(x)
        check(lib().ptc_batch_norm_add_act_bwd(ptr(dy), dtype_code(dy), ptr(x), ptr(res), dtype_code(x), ptr(g), ptr(b), ptr(mean), ptr(rstd), n, c,
                                               int(bool(training)), ACT_CODES[act], ptr(dx), ptr(dres), ptr(dg), ptr(db), ptr(ws), nbytes,
                                               stream_ptr()), "ptc_batch_norm_add_act_bwd")
        return dx, dg, db, dres
    check(lib().ptc_batch_norm_act_bwd(ptr(dy), dtype_code(dy), ptr(x), dtype_code(x), ptr(g), ptr(b), ptr(mean), ptr(rstd), n, c,
                                       int(bool(training)), ACT_CODES[act], ptr(dx), ptr(dg), ptr(db), ptr(ws), nbytes, stream_ptr()),
          "ptc_batch_norm_act_bwd")
    return dx, dg, db


def column_sum(x: torch.Tensor) -> torch.Tensor:
    """x [N, C] (f32 / bf16 / f16) -> fp32 [C] = x.float().sum(0) in one read of x (bias gradients)."""
    require_cuda(x)
    if x.dim() != 2:
        raise PtcoreError("column_sum expects [N, C]")
    if not batch_norm_supported(x.shape[1], x.dtype):
        return x.float().sum(0)          # odd channel counts: ATen on the GPU
    x = x.contiguous()
    n, c = x.shape
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = lib().ptc_batch_norm_workspace_bytes(n, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib().ptc_column_sum(ptr(x), n, c, dtype_code(x), ptr(out), ptr(ws), nbytes, stream_ptr()), "ptc_column_sum")
    return out


def linear_supported_ex(c_in: int, c_out: int, dtype: torch.dtype) -> bool:
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    code = _lib.PTC_F16 if dtype == torch.float16 else _lib.PTC_BF16
    return bool(lib().ptc_linear_supported_ex(int(c_in), int(c_out), code))


def linear_gelu_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """(h, a) = (x W^T + b, GELU(h)) in one kernel; x [N, C_in], weight [C_out, C_in] (both bf16 / f16)."""
    require_cuda(x, weight, bias)
    x, weight = x.contiguous(), weight.contiguous()
    n, c_in = x.shape
    c_out = weight.shape[0]
    h = torch.empty((n, c_out), dtype=x.dtype, device=x.device)
    a = torch.empty((n, c_out), dtype=x.dtype, device=x.device)
    b = None if bias is None else bias.to(torch.float32).contiguous()
    check(lib().ptc_linear_fwd_ex(ptr(x), n, ptr(weight), ptr(b), c_in, c_out, dtype_code(x), 1, 0, ptr(h), ptr(a), stream_ptr()),
          "ptc_linear_fwd_ex")
    return h, a


def linear_gelu_bwd_input(g: torch.Tensor, weight_t: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """dh = (g W) * GELU'(h) in one kernel; g [N, C_out2], weight_t [hidden, C_out2] (= W2^T), h [N, hidden]."""
    require_cuda(g, weight_t, h)
    g, weight_t, h = g.contiguous(), weight_t.contiguous(), h.contiguous()
    n, c_in = g.shape
    c_out = weight_t.shape[0]
    out = torch.empty((n, c_out), dtype=g.dtype, device=g.device)
    check(lib().ptc_linear_fwd_ex(ptr(g), n, ptr(weight_t), 0, c_in, c_out, dtype_code(g), 2, ptr(h), ptr(out), 0, stream_ptr()),
          "ptc_linear_fwd_ex")
    return out


def mlp_supported(c: int, dtype: torch.dtype) -> bool:
    """the one-kernel MLP (csrc/mlp.hip): C = 32 | 64 (hidden 4 C), bf16 / f16"""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    return bool(lib().ptc_mlp_supported(int(c), _lib.PTC_F16 if dtype == torch.float16 else _lib.PTC_BF16))


def mlp_fwd(x, w1, b1, w2, b2, a=None, row_scale=None, want_y=True):
    """m = GELU(x W1^T + b1) W2^T + b2 in one kernel.  a is None: -> m [N, C] (x's dtype).  a [N, C] fp32 (the residual stream):
    -> (z = a + row_scale * m fp32, y = cast(z) or None)."""
    require_cuda(x, w1, b1, w2, b2, a, row_scale)
    x, w1, w2 = x.contiguous(), w1.contiguous(), w2.contiguous()
    n, c = x.shape
    if tuple(w1.shape) != (4 * c, c) or tuple(w2.shape) != (c, 4 * c) or w1.dtype != x.dtype or w2.dtype != x.dtype:
        raise PtcoreError(f"mlp_fwd: x {tuple(x.shape)} {x.dtype}, w1 {tuple(w1.shape)} {w1.dtype}, w2 {tuple(w2.shape)} {w2.dtype}")
    b1 = None if b1 is None else b1.to(torch.float32).contiguous()
    b2 = None if b2 is None else b2.to(torch.float32).contiguous()
    if a is None:
        y = torch.empty((n, c), dtype=x.dtype, device=x.device)
        check(lib().ptc_mlp_fwd(ptr(x), n, c, dtype_code(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), 0, 0, 0, ptr(y), stream_ptr()), "ptc_mlp_fwd")
        return y
    if a.dtype != torch.float32 or tuple(a.shape) != (n, c):
        raise PtcoreError("mlp_fwd: the residual stream must be fp32 [N, C]")
    a = a.contiguous()
    rs = None if row_scale is None else row_scale.to(torch.float32).contiguous()
    z = torch.empty((n, c), dtype=torch.float32, device=x.device)
    y = torch.empty((n, c), dtype=x.dtype, device=x.device) if want_y else None
    check(lib().ptc_mlp_fwd(ptr(x), n, c, dtype_code(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(a), ptr(rs), ptr(z), ptr(y), stream_ptr()), "ptc_mlp_fwd")
    return z, y


def mlp_bwd(dm, x, w1, b1, w2t, want_b1=True, want_b2=True):
    """-> (dx [N, C] like x, dw1 [4C, C], db1 [4C] | None, dw2 [C, 4C], db2 [C] | None), fp32 parameter gradients; w2t = W2^T [4C, C]."""
    require_cuda(dm, x, w1, b1, w2t)
    dm, x, w1, w2t = dm.contiguous(), x.contiguous(), w1.contiguous(), w2t.contiguous()
    n, c = x.shape
    hid = 4 * c
    if dm.shape != x.shape or dm.dtype != x.dtype or tuple(w1.shape) != (hid, c) or tuple(w2t.shape) != (hid, c) or w1.dtype != x.dtype or w2t.dtype != x.dtype:
        raise PtcoreError(f"mlp_bwd: dm {tuple(dm.shape)} {dm.dtype}, x {tuple(x.shape)} {x.dtype}, w1 {tuple(w1.shape)}, w2t {tuple(w2t.shape)}")
    b1 = None if b1 is None else b1.to(torch.float32).contiguous()
    dx = torch.empty_like(x)
    dw1 = torch.empty((hid, c), dtype=torch.float32, device=x.device)
    dw2 = torch.empty((c, hid), dtype=torch.float32, device=x.device)
    db1 = torch.empty((hid,), dtype=torch.float32, device=x.device) if want_b1 else None
    db2 = torch.empty((c,), dtype=torch.float32, device=x.device) if want_b2 else None
    nbytes = lib().ptc_mlp_bwd_workspace_bytes(n, c)
    ws = _ws(nbytes, x.device)
    check(lib().ptc_mlp_bwd(ptr(dm), ptr(x), n, c, dtype_code(x), ptr(w1), ptr(b1), ptr(w2t), ptr(dx), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(ws), nbytes,
                            stream_ptr()), "ptc_mlp_bwd")
    return dx, dw1, db1, dw2, db2


def weight_layouts(desc: torch.Tensor, prefix: torch.Tensor, n: int, total: int) -> None:
    """functional._CastCache: rewrite every backward-pass weight layout described by desc [n,6] / prefix [n+1] (device int64)."""
    require_cuda(desc, prefix)
    check(lib().ptc_weight_layouts(ptr(desc), ptr(prefix), int(n), int(total), stream_ptr()), "ptc_weight_layouts")


def cast_many(desc: torch.Tensor, prefix: torch.Tensor, n: int, total_units: int, dst_dtype: torch.dtype) -> None:
    """functional._CastCache: refresh the 16-bit shadows described by desc [n,3] / prefix [n+1] (device int64) from their fp32 weights."""
    require_cuda(desc, prefix)
    code = {torch.bfloat16: _lib.PTC_BF16, torch.float16: _lib.PTC_F16}[dst_dtype]
    check(lib().ptc_cast_many(ptr(desc), ptr(prefix), int(n), int(total_units), code, stream_ptr()), "ptc_cast_many")


# ------------------------------------------------------------------------------------------------
# libs/pointops2: pair-list attention operators (csrc/pointops2.hip), fp32
# ------------------------------------------------------------------------------------------------
def _p2_check(name, t, shape=None, dtype=torch.float32):
    if t is None:
        return
    if t.dtype != dtype or not t.is_contiguous():
        raise PtcoreError(f"{name}: expected a contiguous {dtype} tensor, got {t.dtype} (contiguous={t.is_contiguous()})")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise PtcoreError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def pair_dot_fwd(q, k, i0, i1, table_q, table_k, rel_idx, with_qk: bool) -> torch.Tensor:
    """out[m,h] = [with_qk] q[i0[m],h].k[i1[m],h] + [table_q] q[i0[m],h].Tq(m,h) + [table_k] k[i1[m],h].Tk(m,h)"""
    require_cuda(q, k, i0, i1, table_q, table_k, rel_idx)
    _, H, d = q.shape
    M = i0.numel()
    _p2_check("q", q)
    _p2_check("k", k, (k.shape[0], H, d) if k is not None else None)
    _p2_check("i0", i0, (M,), torch.int32)
    _p2_check("i1", i1, (M,), torch.int32)
    _p2_check("rel_idx", rel_idx, (M, 3), torch.int32)
    for nm, t in (("table_q", table_q), ("table_k", table_k)):
        _p2_check(nm, t, (t.shape[0], H, d, 3) if t is not None else None)
    out = torch.empty((M, H), dtype=torch.float32, device=q.device)
    check(lib().ptc_pair_dot_fwd(ptr(q), ptr(k), ptr(i0), ptr(i1), ptr(table_q), ptr(table_k), ptr(rel_idx), int(bool(with_qk)), M, H, d,
                                 ptr(out), stream_ptr()), "ptc_pair_dot_fwd")
    return out


def pair_dot_bwd(g, q, k, i0, offsets, i1, table_q, table_k, rel_idx, with_qk: bool, want_q=True, want_k=True, want_tq=True,
                 want_tk=True):
    require_cuda(g, q, k, i0, offsets, i1, table_q, table_k, rel_idx)
    Nq, H, d = q.shape
    M = i0.numel()
    _p2_check("grad_out", g, (M, H))
    if offsets is not None and offsets.numel() != Nq + 1:
        raise PtcoreError(f"offsets must have {Nq + 1} entries (one segment per query row), got {offsets.numel()}")
    Nk = k.shape[0] if k is not None else 0
    L = table_q.shape[0] if table_q is not None else (table_k.shape[0] if table_k is not None else 0)
    dq = torch.empty_like(q) if want_q else None
    dk = torch.empty_like(k) if (want_k and k is not None) else None
    dtq = torch.empty_like(table_q) if (want_tq and table_q is not None) else None
    dtk = torch.empty_like(table_k) if (want_tk and table_k is not None) else None
    check(lib().ptc_pair_dot_bwd(ptr(g), ptr(q), ptr(k), ptr(i0), ptr(offsets), ptr(i1), ptr(table_q), ptr(table_k), ptr(rel_idx),
                                 int(bool(with_qk)), M, Nq, Nk, L, H, d, ptr(dq), ptr(dk), ptr(dtq), ptr(dtk), stream_ptr()),
          "ptc_pair_dot_bwd")
    return dq, dk, dtq, dtk


def pair_aggregate_fwd(attn, v, i0, offsets, i1, table_v, rel_idx, n_q: int) -> torch.Tensor:
    """out[n,h,c] = sum_{m: i0[m] = n} attn[m,h] (v[i1[m],h,c] + [table_v] Tv(m,h,c)), out [n_q, H, d]"""
    require_cuda(attn, v, i0, offsets, i1, table_v, rel_idx)
    _, H, d = v.shape
    M = i1.numel()
    _p2_check("attn", attn, (M, H))
    _p2_check("v", v)
    _p2_check("i0", i0, (M,), torch.int32)
    _p2_check("i1", i1, (M,), torch.int32)
    _p2_check("rel_idx", rel_idx, (M, 3), torch.int32)
    _p2_check("table", table_v, (table_v.shape[0], H, d, 3) if table_v is not None else None)
    if offsets is not None and offsets.numel() != int(n_q) + 1:
        raise PtcoreError(f"offsets must have {int(n_q) + 1} entries, got {offsets.numel()}")
    out = torch.empty((int(n_q), H, d), dtype=torch.float32, device=v.device)
    check(lib().ptc_pair_aggregate_fwd(ptr(attn), ptr(v), ptr(i0), ptr(offsets), ptr(i1), ptr(table_v), ptr(rel_idx), M, int(n_q), H, d,
                                       ptr(out), stream_ptr()), "ptc_pair_aggregate_fwd")
    return out


def pair_aggregate_bwd(g, attn, v, i0, i1, table_v, rel_idx, want_attn=True, want_v=True, want_tv=True):
    require_cuda(g, attn, v, i0, i1, table_v, rel_idx)
    Nv, H, d = v.shape
    M = i1.numel()
    _p2_check("grad_out", g, (g.shape[0], H, d))
    L = table_v.shape[0] if table_v is not None else 0
    da = torch.empty_like(attn) if want_attn else None
    dv = torch.empty_like(v) if want_v else None
    dtv = torch.empty_like(table_v) if (want_tv and table_v is not None) else None
    check(lib().ptc_pair_aggregate_bwd(ptr(g), ptr(attn), ptr(v), ptr(i0), ptr(i1), ptr(table_v), ptr(rel_idx), M, Nv, L, H, d, ptr(da),
                                       ptr(dv), ptr(dtv), stream_ptr()), "ptc_pair_aggregate_bwd")
    return da, dv, dtv

