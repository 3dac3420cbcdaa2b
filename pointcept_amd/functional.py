"""Differentiable operators of the engine (torch.autograd.Function wrappers over pointcept_amd.ops).

Design rule: every backward is written in GATHER form over precomputed index maps, so no kernel
uses floating-point atomics and every result is bit-reproducible run to run:
  * backward of a row gather through a permutation-with-padding = a row gather through the inverse
    map (+ the duplicate slot),
  * backward of the unpooling gather = a segmented sum over the cluster CSR,
  * sparse-conv dgrad = the forward kernel on the transposed table / mirrored weights,
  * sparse-conv wgrad = per-split partial sums + a deterministic reduction.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, config, ops
from ._lib import PtcoreError


# ------------------------------------------------------------------------------------------------
# weight casts under autocast: ONE multi-tensor kernel per step instead of one cast kernel per layer
# ------------------------------------------------------------------------------------------------
# Every Linear / sparse-conv forward needs its fp32 master weight in the autocast dtype.  Casting at the
# point of use cost ~190 launches of 5 us per step (profiles/r01_k: bfloat16_copy_kernel).  The cache keeps
# one low-precision shadow per weight storage and, at the first miss after the optimizer step (version
# counters moved), refreshes ALL shadows with a single torch._foreach_copy_.
class _CastCache:
    """Entries die with the parameter they shadow (weak reference on the owning tensor: nothing here pins a discarded
    model's storage), are validated by the tensor's version counter, and can be dropped explicitly:
    `invalidate_weight_casts()` after writes that do not move the counter (`p.data.copy_()`, `p.data.mul_()` -- EMA /
    weight clipping code written against `.data`; optimizers, `load_state_dict` and `copy_` under `no_grad` do move it).
    A refresh between the forward and the backward of one live graph raises autograd's saved-tensor version error, as
    modifying the weight itself does in stock PyTorch."""

    def __init__(self, cuda_only: bool = True):
        self.entries = {}   # (data_ptr, numel, dtype) -> [weakref(owner), shadow (flat), version]
        self.cuda_only = cuda_only   # False only in the CPU unit test of this class
        # backward-pass layouts of the shadows ([c_in][taps][c_out], see layout()): shadow data_ptr -> entry, and
        # (shadow data_ptr, mode, taps_dst) -> [entry, buffer, shadow generation the buffer was made from, dims]
        self.by_shadow = {}
        self.layouts = {}
        self._desc = None   # (signature, desc tensor, prefix tensor, total)
        self._cast_desc = None   # the same for the one-launch cast (_cast_many)

    def invalidate(self) -> None:
        self._cast_desc = None
        self.entries.clear()
        self.by_shadow.clear()
        self.layouts.clear()
        self._desc = None

    _MODES = {"mirror": 0, "repeat": 1, "keep": 2}

    def layout(self, w: torch.Tensor, mode: str, taps_dst: int = 0) -> Optional[torch.Tensor]:
        """[c_in, taps', c_out] layout of a weight shadow for the input-gradient GEMMs: w = what get() returned, shape
        [c_out, c_in] or [c_out, taps, c_in]; mode "mirror" (taps reversed: submanifold dgrad; the plain transpose when
        taps = 1), "keep" (tap order kept) or "repeat" (taps = 1 matrix repeated taps_dst times).  Persistent buffers; every
        stale layout of the model is rewritten in ONE launch (ptc_weight_layouts) the first time one is asked for after the
        shadows were refreshed.  None when w is not a cache shadow (fp32 runs, padded copies): the caller permutes itself."""
        if not w.is_cuda or w.dim() not in (2, 3) or not w.is_contiguous() or w.element_size() != 2:
            return None
        e = self.by_shadow.get(w.data_ptr())
        if e is None or e[1].numel() != w.numel():
            return None
        co, ks, ci = (w.shape[0], 1, w.shape[1]) if w.dim() == 2 else tuple(w.shape)
        kd = int(taps_dst) if mode == "repeat" else ks
        if mode == "repeat" and ks != 1:
            return None
        key = (w.data_ptr(), mode, kd)
        lay = self.layouts.get(key)
        if lay is None:
            lay = self.layouts[key] = [e, torch.empty(ci * kd * co, dtype=w.dtype, device=w.device), None, (co, ks, ci, kd, self._MODES[mode])]
            self._desc = None
            self._refresh_layouts(w.device, only=key)      # first use: just this one (the table is rebuilt at the next full refresh)
        elif lay[2] != e[3]:
            self._refresh_layouts(w.device)
        return lay[1].view(ci, kd, co)

    def _refresh_layouts(self, device, only=None) -> None:
        live = [(k, v) for k, v in self.layouts.items() if v[0][0]() is not None and v[1].device == device and (only is None or k == only)]
        sig = tuple(k for k, _ in live)
        if only is not None or self._desc is None or self._desc[0] != sig:
            rows, prefix, total = [], [0], 0
            for _, (e, buf, _, (co, ks, ci, kd, mode)) in live:
                rows.append([e[1].data_ptr(), buf.data_ptr(), co, ks, ci, kd | (mode << 32)])
                total += ci * kd * co
                prefix.append(total)
            built = (sig, torch.tensor(rows, dtype=torch.int64, device=device), torch.tensor(prefix, dtype=torch.int64, device=device), total)
            if only is None:
                self._desc = built
        else:
            built = self._desc
        _, desc, prefix, total = built
        ops.weight_layouts(desc, prefix, len(live), total)
        for _, v in live:
            v[2] = v[0][3]

    def _cast_many(self, stale, srcs) -> bool:
        """ONE launch for the whole refresh (ptc_cast_many) instead of ~29 multi-tensor launches; False = not applicable."""
        if not stale or not all(s.is_cuda and s.dtype == torch.float32 for s in srcs):
            return False
        dts = {x[1].dtype for x in stale}
        if len(dts) != 1 or next(iter(dts)) not in (torch.bfloat16, torch.float16) or len({s.device for s in srcs}) != 1:
            return False
        sig = tuple((s.data_ptr(), x[1].data_ptr(), s.numel()) for x, s in zip(stale, srcs))
        dt = next(iter(dts))
        if self._cast_desc is None:
            self._cast_desc = {}
        if dt not in self._cast_desc or self._cast_desc[dt][0] != sig:
            rows, prefix, units = [], [0], 0
            for src_ptr, dst_ptr, numel in sig:
                rows.append([src_ptr, dst_ptr, numel])
                units += (numel + 7) // 8
                prefix.append(units)
            dev = srcs[0].device
            self._cast_desc[dt] = (sig, torch.tensor(rows, dtype=torch.int64, device=dev), torch.tensor(prefix, dtype=torch.int64, device=dev), units)
        _, desc, prefix, units = self._cast_desc[dt]
        ops.cast_many(desc, prefix, len(sig), units, dt)
        return True

    def get(self, w: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
        if w.dtype == dt:
            return w
        owner = w._base if w._base is not None else w     # conv weights arrive as views of the Parameter
        if not isinstance(owner, torch.nn.Parameter) and (owner.grad_fn is not None or not owner.is_leaf):   # a weight COMPUTED this step: nothing to cache
            return w.to(dt)
        if not w.is_contiguous() or (self.cuda_only and not w.is_cuda) or owner.numel() != w.numel() or not owner.is_contiguous():
            return w.to(dt)
        key = (w.data_ptr(), w.numel(), dt)
        e = self.entries.get(key)
        if e is not None and e[0]() is not owner:          # address reused by another tensor
            e = None
        if e is not None and e[2] == owner._version:
            return e[1].view(w.shape)
        if e is None:
            entries = self.entries

            def _drop(_ref, key=key, entries=entries, cache=self):
                cur = entries.get(key)
                if cur is not None and cur[0] is _ref:
                    del entries[key]
                    ptr = cur[1].data_ptr()
                    cache.by_shadow.pop(ptr, None)
                    for k in [k for k in cache.layouts if k[0] == ptr]:
                        del cache.layouts[k]
                    cache._desc = None

            e = [weakref.ref(owner, _drop), torch.empty(w.numel(), dtype=dt, device=w.device), -1, 0]   # ..., generation of the shadow
            self.entries[key] = e
            self.by_shadow[e[1].data_ptr()] = e
        # one multi-tensor copy refreshes every stale shadow (the first miss after an optimizer step)
        stale, srcs = [], []
        for x in list(self.entries.values()):
            o = x[0]()
            if o is not None and x[2] != o._version:
                stale.append(x)
                srcs.append(o.detach().reshape(-1))
        # per shadow dtype: a process that ran bf16 AND fp16 autocast holds both kinds, and torch._foreach_copy_ on a destination list
        # of mixed dtypes converts every tensor to the FIRST one's dtype on the CUDA fast path (seen on PyTorch 2.10 / ROCm 7: the
        # second dtype's shadows received the first dtype's bit patterns -- tools/fp16_stem_probe.py)
        with torch.no_grad():
            for dt_ in {x[1].dtype for x in stale}:
                grp = [(x, s_) for x, s_ in zip(stale, srcs) if x[1].dtype == dt_]
                gx, gs = [g_[0] for g_ in grp], [g_[1] for g_ in grp]
                if not self._cast_many(gx, gs):
                    torch._foreach_copy_([x[1] for x in gx], gs)
        for x, o in zip(stale, srcs):
            x[2] = x[0]()._version
            x[3] += 1
        return e[1].view(w.shape)


# The per-step refresh of the 16-bit weight shadows is ONE launch (csrc/rows.hip: cast_many_kernel; _CastCache._cast_many) where every
# stale shadow is a CUDA fp32 -> bf16 / f16 pair on one device, torch._foreach_copy_ (~29 multi-tensor launches at the bench config) otherwise.
# Timed in the step: 49.4 vs 49.8 ms (profiles/r03_a_knob_ab.txt) -- neutral on the GPU clock, 28 launches less on the host.
_cast_cache = _CastCache()


def invalidate_weight_casts() -> None:
    """Drop every low-precision weight shadow (call after `.data` writes to parameters, see _CastCache)."""
    _cast_cache.invalidate()


def _autocast_on() -> bool:
    return torch.is_autocast_enabled("cuda")


def _autocast_dtype(t: torch.Tensor) -> torch.dtype:
    if _autocast_on():
        return torch.get_autocast_dtype("cuda")
    return t.dtype


# ------------------------------------------------------------------------------------------------
# row gathers
# ------------------------------------------------------------------------------------------------
class _GatherRows(Function):
    @staticmethod
    def forward(ctx, src, idx, bwd_idx, bwd_idx2):
        ctx.save_for_backward(bwd_idx, bwd_idx2)
        return ops.gather_rows(src, idx)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        bwd_idx, bwd_idx2 = ctx.saved_tensors
        return ops.gather_rows(grad.contiguous(), bwd_idx, bwd_idx2), None, None, None


def gather_rows(src: torch.Tensor, idx: torch.Tensor, bwd_idx: torch.Tensor,
                bwd_idx2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = src[idx[i]].  `bwd_idx` [n_src] (and optional `bwd_idx2`) list, for every source row,
    the output row(s) it was copied to (-1 = none): grad_src[p] = g[bwd_idx[p]] + g[bwd_idx2[p]]."""
    return _GatherRows.apply(src, idx, bwd_idx, bwd_idx2)


class _GatherByCluster(Function):
    @staticmethod
    def forward(ctx, src, cluster, perm, indptr):
        ctx.save_for_backward(perm, indptr)
        return ops.gather_rows(src, cluster)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        perm, indptr = ctx.saved_tensors
        gsrc, _ = ops.segment_csr_fwd(grad.contiguous(), perm, indptr, "sum")
        return gsrc, None, None, None


class _GatherByClusterAdd(Function):
    """addend + src[cluster] in one kernel; the gradient of the addend is the incoming gradient itself"""

    @staticmethod
    def forward(ctx, src, cluster, perm, indptr, addend):
        ctx.save_for_backward(perm, indptr)
        return ops.gather_rows_add(src, cluster, addend)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        perm, indptr = ctx.saved_tensors
        grad = grad.contiguous()
        gsrc, _ = ops.segment_csr_fwd(grad, perm, indptr, "sum")
        return gsrc, None, None, None, grad


def gather_by_cluster_add(addend: torch.Tensor, src: torch.Tensor, cluster: torch.Tensor, perm: torch.Tensor, indptr: torch.Tensor):
    """addend + src[cluster] (SerializedUnpooling, ptv3m1:478: `parent.feat + point.feat[inverse]`) -- one pass where the kernel serves the
    shape (same 16-bit / fp32 dtype on both sides, rows of whole 16-byte lanes), the gather followed by torch's add otherwise."""
    if ops.gather_rows_add_supported(src, addend) and addend.shape[0] == cluster.numel():
        return _GatherByClusterAdd.apply(src, cluster, perm, indptr, addend)
    return addend + gather_by_cluster(src, cluster, perm, indptr)


def gather_by_cluster(src: torch.Tensor, cluster: torch.Tensor, perm: torch.Tensor, indptr: torch.Tensor):
    """out[p] = src[cluster[p]] (SerializedUnpooling, ptv3m1:478).  Backward = segmented sum over
    the cluster CSR (perm = points sorted by cluster, indptr = idx_ptr)."""
    return _GatherByCluster.apply(src, cluster, perm, indptr)


# ------------------------------------------------------------------------------------------------
# segment_csr
# ------------------------------------------------------------------------------------------------
class _SegmentCSR(Function):
    @staticmethod
    def forward(ctx, src, perm, indptr, reduce, covers_all):
        out, arg = ops.segment_csr_fwd(src, perm, indptr, reduce)
        ctx.reduce = reduce
        ctx.covers_all = covers_all
        ctx.n_src = src.shape[0]
        ctx.save_for_backward(perm, indptr, arg)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        perm, indptr, arg = ctx.saved_tensors
        g = ops.segment_csr_bwd(grad.contiguous(), perm, indptr, arg, ctx.n_src, ctx.reduce, ctx.covers_all)
        return g, None, None, None, None


def segment_csr(src: torch.Tensor, indptr: torch.Tensor, reduce: str = "sum", perm: Optional[torch.Tensor] = None,
                covers_all: bool = False):
    """torch_scatter.segment_csr(src[perm], indptr, reduce) with the gather fused (ptv3m1:416-421).
    covers_all: the caller guarantees that the segments partition ALL rows of src (indptr[0] = 0, indptr[-1] = N, perm a
    permutation -- the pooling layers build them that way): the backward then writes every row and skips the zero fill."""
    if src.dim() == 1:
        return _SegmentCSR.apply(src[:, None], perm, indptr, reduce, covers_all)[:, 0]
    return _SegmentCSR.apply(src, perm, indptr, reduce, covers_all)


# ------------------------------------------------------------------------------------------------
# sparse convolution
# ------------------------------------------------------------------------------------------------
def _pad_to(x: torch.Tensor, dim: int, mult: int) -> torch.Tensor:
    c = x.shape[dim]
    r = (-c) % mult
    if r == 0:
        return x
    pad = [0, 0] * (x.dim() - 1 - dim) + [0, r]
    return F.pad(x, pad)


# The duplicate merge is ONE segmented sum (ptc_segment_csr_fwd) over a CSR of the representatives that is built once per coordinate
# set -- no python loop, no host sync per conv backward; additions in ascending row order.
_dup_csr = {}   # id(rep) -> (weakref(rep), perm, indptr, keep)


def _dup_csr_of(rep: torch.Tensor):
    e = _dup_csr.get(id(rep))
    if e is not None and e[0]() is rep:
        return e[1:]
    n = rep.shape[0]
    perm = torch.sort(rep, stable=True).indices                       # rows grouped by representative, ascending row inside a group
    counts = torch.bincount(rep, minlength=n)
    indptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)])
    keep = counts == 0                                                # rows that are copies: nothing is summed INTO them
    key = id(rep)

    def _drop(ref, key=key):
        cur = _dup_csr.get(key)
        if cur is not None and cur[0] is ref:
            del _dup_csr[key]

    _dup_csr[key] = (weakref.ref(rep, _drop), perm, indptr, keep)
    return perm, indptr, keep


def _merge_duplicate_rows(g: torch.Tensor, rep: torch.Tensor) -> torch.Tensor:
    """g with, for every voxel listed more than once, the rows of all its copies summed into the representative row
    rep[i] (the lowest row of the voxel).  Only runs for inputs that carry duplicate coordinates (Mix3D batches); no atomics, fixed order."""
    perm, indptr, keep = _dup_csr_of(rep)
    merged, _ = ops.segment_csr_fwd(g.contiguous(), perm, indptr, "sum")   # row t: g[t] + its copies, ascending rows
    return torch.where(keep[:, None], g, merged.to(g.dtype))


_rev_index_cache = {}
def _reverse_index(n: int, device) -> torch.Tensor:
    key = (n, device)
    t = _rev_index_cache.get(key)
    if t is None:
        t = _rev_index_cache[key] = torch.arange(n - 1, -1, -1, device=device)
    return t


# (Weight gradients on a second HIP stream beside the input gradient were measured SLOWER in the step -- 52.6-53.6 against 51.1-51.8 ms,
#  profiles/r02_t_bench_ab.txt: two cross-stream waits per Function cost more than the small kernels gain -- and removed.)


# ---- cast twins of activations --------------------------------------------------------------------
# The fused residual joint writes the fp32 stream x AND its autocast-dtype copy in one pass (add_norm: y = cast(z)).  GEMM
# wrappers that would cast x again (pooling / unpooling projections, the segmentation head: 13 casts of [N, C] per step at
# the bench config, and 13 more in the backward) look the copy up here instead.  Same values (one RNE rounding of the same
# fp32 number); the gradient then reaches the joint through its `y` output, which the kernel adds in fp32.
_act_twins = {}   # id(x) -> (weakref(x), twin, x._version)


def register_cast_twin(x: torch.Tensor, twin: torch.Tensor) -> None:
    key = id(x)

    def _drop(ref, key=key):
        cur = _act_twins.get(key)
        if cur is not None and cur[0] is ref:
            del _act_twins[key]

    _act_twins[key] = (weakref.ref(x, _drop), twin, x._version)


def cast_twin(x: torch.Tensor, dt: torch.dtype) -> Optional[torch.Tensor]:
    e = _act_twins.get(id(x))
    if e is not None and e[0]() is x and e[2] == x._version and e[1].dtype == dt and e[1].shape == x.shape:
        return e[1]
    return None


class _SparseConv(Function):
    """out = conv(feat; weight [C_out, kv, C_in], bias) over gather table `nbr`;
    `nbr_t` is the table of the transposed map (SubM: the same table, with mirrored weights).

    Duplicate voxel coordinates (legal input: Mix3D) make the maps many-to-one -- every copy of a voxel reads
    the SAME neighbour rows (lowest row wins in the hash) and no row ever reads a non-lowest copy -- so the transposed
    gather needs two corrections to stay the exact adjoint: `dup_out` = representative row of every OUTPUT row (the
    incoming gradient of the copies is first summed into their representative), `dup_in` = representative row of every
    INPUT row (copies that nothing reads get a zero gradient).  Both None for duplicate-free coordinates."""

    @staticmethod
    def forward(ctx, feat, weight, bias, nbr, nbr_t, mirror, dup_out, dup_in, blocks):
        dt = _autocast_dtype(feat)
        c_out, kv, c_in = weight.shape
        # channels are padded to 16; a stem (c_in <= 8: 6 colour + normal channels, k = 5) only to 8: its gathered rows are then 16
        # bytes and conv3 packs four table rows into one MFMA step (csrc/conv3.h, KPC = 16)
        cpad = 8 if (c_in <= 8 and kv > 1 and dt != torch.float32) else 16
        opad = 16
        if (blocks is not None and kv == 27 and c_in == c_out and 16 < c_in < 64 and c_in != 32 and dt != torch.float32
                and ops.block_plan(64, 64, 27, dt, feat.shape[0]) is not None):
            # round 5: a square submanifold 3^3 convolution of 17 .. 63 channels (PT-v3m2's 48, LitePT's 36 at stage 0) on the block-staged,
            # register-weight kernels of the next width they exist for (conv7 / wgrad7: 32 | 64 channels), zero-padded: the 48-channel
            # stage-0 convolutions ran 403 us on conv2 against 145 us for conv7 at 64 (profiles/r05_m_m2_kernel_stats.csv)
            cpad = opad = 32 if c_in < 32 else 64
        # (measured and dropped, round 5: SpUNet's 96-channel level-0 convolutions zero-padded onto the 128-channel instances -- step 26.7 -> 31.2 ms)
        f = _pad_to(feat.to(dt), 1, cpad).contiguous()
        w = _pad_to(_pad_to(_cast_cache.get(weight, dt), 2, cpad), 0, opad).contiguous()
        b = None if bias is None else _pad_to(bias.float(), 0, opad)
        # (padding the 6 -> 16 channel stem further to 32 so that conv3 takes it was measured SLOWER than conv2:
        #  1.20 ms vs 0.74 ms for the 125-offset table, r01_u)
        # blocks = ops.BlockProvider of the (submanifold) table: the LDS-staged kernel where the shape allows it
        blk = None if blocks is None else blocks.get(f.shape[1], w.shape[0], dt, conv=True)
        out = ops.spconv_fwd(f, w, b, nbr, blk)
        ctx.save_for_backward(f, w, nbr, nbr_t, dup_out, dup_in)
        ctx.blocks = blocks if mirror else None    # SubM: dgrad runs over the same table
        ctx.mirror = mirror
        ctx.shape = (c_out, kv, c_in)
        ctx.in_dtype, ctx.w_dtype = feat.dtype, weight.dtype
        ctx.has_bias = bias is not None
        return out[:, :c_out] if out.shape[1] != c_out else out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        f, w, nbr, nbr_t, dup_out, dup_in = ctx.saved_tensors
        c_out, kv, c_in = ctx.shape
        g = _pad_to(grad.to(f.dtype), 1, w.shape[0] if w.shape[0] == f.shape[1] and w.shape[0] in (32, 64) else 16).contiguous()
        dfeat = dw = dbias = None
        if ctx.needs_input_grad[1]:
            # submanifold 3^3, 32 | 64 channels: the block-staged weight gradient over the tables the forward built (csrc/wgrad7.h)
            blkw = None if (ctx.blocks is None or not config.WGRAD_BLK) else ctx.blocks.get(f.shape[1], g.shape[1], f.dtype)
            dw = ops.spconv_wgrad(f, g, nbr, blk=blkw)[:c_out, :, :c_in].to(ctx.w_dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = ops.column_sum(grad)
        if ctx.needs_input_grad[0]:
            wt = _cast_cache.layout(w, "mirror" if ctx.mirror else "keep")   # all layers' layouts in one launch per step
            if wt is None:
                wt = w.permute(2, 1, 0)
                if ctx.mirror:   # W' = W.permute(ci, k, co).flip(k), contiguous, in ONE launch (flip + contiguous were two)
                    wt = wt.index_select(1, _reverse_index(wt.shape[1], wt.device))
                else:
                    wt = wt.contiguous()
            gm = g if dup_out is None else _merge_duplicate_rows(g, dup_out)
            blk = None if ctx.blocks is None else ctx.blocks.get(gm.shape[1], wt.shape[0], gm.dtype, conv=True)
            dfeat = ops.spconv_fwd(gm, wt, None, nbr_t, blk)[:, :c_in].to(ctx.in_dtype)
            if dup_in is not None:
                own = dup_in == torch.arange(dup_in.numel(), device=dup_in.device, dtype=dup_in.dtype)
                dfeat = dfeat * own[:, None].to(dfeat.dtype)
        return dfeat, dw, dbias, None, None, None, None, None, None


def sparse_conv(feat, weight, bias, nbr, nbr_t, mirror: bool, dup_out=None, dup_in=None, blocks=None):
    """weight: [C_out, kv, C_in] (a view of the spconv-layout parameter [C_out,k0,k1,k2,C_in]).  blocks: optional
    ops.BlockProvider of `nbr` (submanifold tables only)."""
    return _SparseConv.apply(feat, weight, bias, nbr, nbr_t, mirror, dup_out, dup_in, blocks)


# ------------------------------------------------------------------------------------------------
# dense row-wise GEMM (nn.Linear on [N,C] point features) on the sparse-conv MFMA kernels
# ------------------------------------------------------------------------------------------------
# Shape policy: contractions of <= 256 channels run on the persistent linear2 kernel, wider ones on the chunked implicit-GEMM kernel with
# an identity table (csrc/conv3.h, IDENT: multiples of 128 -- PTv3's fc2 / dgrad-of-fc1 / qkv / proj of the 128..512-channel stages --
# and, round 5, every multiple of 32 through its general chunking: the 288 .. 2304-wide MLPs of PT-v3m2 / m3 / LitePT), every weight
# gradient on the split-K kernel (wgrad2).  Operands are zero-padded to the kernels' granularity (`_gemm_pad`): 16 channels, 32 on
# both sides as soon as either side of the GEMM exceeds 256 (its input-gradient GEMM contracts over the OUTPUT width).  There is no
# library GEMM behind this file (rounds 2-4 sent what the kernels did not cover to hipBLASLt).
def _own_gemm(n_rows: int, k: int, dtype: torch.dtype, c_out: int = 32) -> bool:
    return dtype != torch.float32 and (k <= 256 or (k % 32 == 0 and c_out % 32 == 0))


def _gemm_pad(c_in: int, c_out: int, dtype: torch.dtype) -> int:
    return 32 if (dtype != torch.float32 and (c_in > 256 or c_out > 256)) else 16


class _Linear(Function):
    """out[o] = W x[tab[o]] + b.  tab_fwd [1, n_out] int32 (None = identity) folds a row gather into
    the GEMM; tab_bwd [k, n_in] int32 lists, for every input row, the output rows that read it
    (-1 = none), which makes the input gradient a gather too: dx[p] = sum_k dout[tab_bwd[k][p]] W."""

    @staticmethod
    def forward(ctx, x, weight, bias, tab_fwd, tab_bwd):
        dt = _autocast_dtype(x)
        c_out, c_in = weight.shape
        pad = _gemm_pad(c_in, c_out, dt)
        xp = _pad_to(x.to(dt), 1, pad).contiguous()
        wp = _pad_to(_pad_to(_cast_cache.get(weight, dt), 1, pad), 0, pad).contiguous()
        bp = None if bias is None else _pad_to(bias.float(), 0, pad)
        out = ops.spconv_fwd(xp, wp[:, None, :], bp, tab_fwd)        # linear2 / conv3 (identity table) / the fp32 kernels: no library GEMM
        ctx.pad = pad
        ctx.save_for_backward(xp, wp, tab_fwd, tab_bwd)
        ctx.shape = (c_out, c_in)
        ctx.in_dtype, ctx.w_dtype = x.dtype, weight.dtype
        ctx.b_dtype = None if bias is None else bias.dtype
        return out[:, :c_out] if out.shape[1] != c_out else out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        xp, wp, tab_fwd, tab_bwd = ctx.saved_tensors
        c_out, c_in = ctx.shape
        g = _pad_to(grad.to(xp.dtype), 1, ctx.pad).contiguous()
        dx = dw = db = None
        want_b = ctx.b_dtype is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            res = ops.spconv_wgrad(xp, g, tab_fwd, want_bias=want_b)
            dwp, dbp = res if want_b else (res, None)
            dw = dwp[:c_out, 0, :c_in].to(ctx.w_dtype)
            if want_b:
                db = dbp[:c_out].to(ctx.b_dtype)
        if ctx.needs_input_grad[0]:
            slots = tab_bwd.shape[0] if tab_bwd is not None else 1
            wt = _cast_cache.layout(wp, "repeat", slots) if slots > 1 else _cast_cache.layout(wp, "mirror")
            if wt is None:
                wt = wp.t().contiguous()[:, None, :]                       # [c_in, 1, c_out]
                if slots > 1:
                    wt = wt.expand(-1, slots, -1).contiguous()              # same W for every slot
            dx = ops.spconv_fwd(g, wt, None, tab_bwd)
            dx = dx[:, :c_in].to(ctx.in_dtype)
        return dx, dw, db, None, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           tab_fwd: Optional[torch.Tensor] = None, tab_bwd: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.linear for [N, C_in] point features: out = x W^T + b, autocast-aware (bf16 operands, fp32
    accumulate), forward / dgrad / wgrad (+ fused bias gradient) on the identity-table MFMA kernels.
    With (tab_fwd, tab_bwd) the GEMM also applies a row permutation-with-padding:
    out[o] = W x[tab_fwd[0][o]] + b  (= F.linear(x)[tab] = F.linear(x[tab]): a row-wise map commutes
    with a row gather), which removes the separate gather pass around serialized attention."""
    if x.dim() != 2:
        raise PtcoreError("linear expects [N, C] features")
    if (tab_fwd is None) != (tab_bwd is None):
        raise PtcoreError("linear: tab_fwd and tab_bwd go together")
    return _Linear.apply(x, weight, bias, tab_fwd, tab_bwd)


# ------------------------------------------------------------------------------------------------
# layer norm
# ------------------------------------------------------------------------------------------------
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        y, mean, rstd = ops.layer_norm_fwd(x, weight, bias, eps, out_dtype)
        ctx.save_for_backward(x, mean, rstd, weight)
        ctx.has_affine = weight is not None
        ctx.has_bias = bias is not None          # nn.LayerNorm(bias=False): no gradient may be returned for the absent input
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, mean, rstd, weight = ctx.saved_tensors
        dx, dg, db = ops.layer_norm_bwd(dy, x, mean, rstd, weight, want_affine=ctx.has_affine)
        if ctx.has_affine:
            dg, db = dg.to(weight.dtype), (db.to(weight.dtype) if ctx.has_bias else None)
        return dx, dg, db, None, None


def layer_norm(x: torch.Tensor, weight, bias, eps: float = 1e-5, out_dtype: Optional[torch.dtype] = None):
    """nn.LayerNorm over the last dim of [N,C].  Default output dtype follows PyTorch: fp32 under
    autocast (layer_norm is an fp32 autocast op) else x.dtype; `out_dtype=torch.bfloat16` lets the
    caller take the value already rounded for a following bf16 GEMM (same number the autocast
    cast would produce, one HBM pass less)."""
    if out_dtype is None:
        out_dtype = torch.float32 if (_autocast_on() or x.dtype == torch.float32) else x.dtype
    return _LayerNorm.apply(x, weight, bias, float(eps), out_dtype)


# ------------------------------------------------------------------------------------------------
# fused residual joint: z = a + row_scale * LN_A(u),  y = LN_B(z)
# ------------------------------------------------------------------------------------------------
class _AddNorm(Function):
    @staticmethod
    def forward(ctx, u, a, row_scale, ga, ba, eps_a, gb, bb, eps_b, has_a, has_b, y_dtype):
        norm_a = (None if ga is None else ga.float(), None if ba is None else ba.float(), eps_a) if has_a else None
        norm_b = (None if gb is None else gb.float(), None if bb is None else bb.float(), eps_b) if has_b else None
        z, y, st_a, st_b = ops.add_norm_fwd(u, a, row_scale, norm_a, norm_b, y_dtype)
        ctx.save_for_backward(u, z, row_scale, st_a, st_b, None if norm_a is None else norm_a[0],
                              None if norm_b is None else norm_b[0])
        ctx.has_a, ctx.has_b = has_a, has_b
        ctx.aff = (ga is not None, gb is not None)
        ctx.has_bias = (ba is not None, bb is not None)     # a weight-only LayerNorm gets no bias gradient back (autograd raises on one)
        ctx.a_dtype = a.dtype
        # an unused output (the cast copy of a stage's last block, which the pooling does not read) must not be
        # materialised as a zero gradient: that was 8 zero fills + 8 extra reads of [N, C] per step
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable()
        if y is None:
            y = z.new_empty(0)
            ctx.mark_non_differentiable(y)
        return z, y

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dy):
        u, z, row_scale, st_a, st_b, ga, gb = ctx.saved_tensors
        if dy is not None and dy.numel() == 0:
            dy = None
        if dz is None and dy is None:
            return (None,) * 12
        da, du, dga, dba, dgb, dbb = ops.add_norm_bwd(dz, dy, z, u, row_scale, ga, st_a, gb, st_b,
                                                      ctx.has_a and ctx.aff[0], ctx.has_b and ctx.aff[1] and dy is not None,
                                                      da_dtype=ctx.a_dtype)
        return du, da, None, dga, (dba if ctx.has_bias[0] else None), None, dgb, (dbb if ctx.has_bias[1] else None), None, None, None, None


def add_norm(u: torch.Tensor, a: torch.Tensor, row_scale: Optional[torch.Tensor] = None, norm_a=None, norm_b=None,
             y_dtype: Optional[torch.dtype] = None):
    """One pass over a residual joint of the PTv3 Block: z = a + row_scale[:,None] * f(u) (fp32 residual
    stream), y = g(z) as `y_dtype`.  f / g are nn.LayerNorm modules (norm_a / norm_b) or identity.
    Returns (z, y); y is None when y_dtype is None."""
    if a.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        a = a.float()
    ga, ba, ea = (norm_a.weight, norm_a.bias, norm_a.eps) if norm_a is not None else (None, None, 0.0)
    gb, bb, eb = (norm_b.weight, norm_b.bias, norm_b.eps) if norm_b is not None else (None, None, 0.0)
    z, y = _AddNorm.apply(u, a, row_scale, ga, ba, float(ea), gb, bb, float(eb), norm_a is not None, norm_b is not None,
                          y_dtype)
    return z, (y if y_dtype is not None else None)


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
class _AttnVarlen(Function):
    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_seqlen, softmax_scale, dropout_p, seed):
        out, lse = ops.attn_varlen_fwd(qkv, cu_seqlens, max_seqlen, softmax_scale, dropout_p, seed)
        ctx.save_for_backward(qkv, out, lse, cu_seqlens)
        ctx.max_seqlen, ctx.scale, ctx.drop = max_seqlen, softmax_scale, (dropout_p, seed)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, out, lse, cu = ctx.saved_tensors
        dqkv = ops.attn_varlen_bwd(qkv, out, dout.contiguous(), lse, cu, ctx.max_seqlen, ctx.scale, *ctx.drop)
        return dqkv, None, None, None, None, None


def attn_varlen_qkvpacked(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int,
                          softmax_scale: Optional[float] = None, dropout_p: float = 0.0, seed: Optional[int] = None) -> torch.Tensor:
    """flash_attn.flash_attn_varlen_qkvpacked_func semantics (ptv3m1:208-214),
    non-causal, qkv [T,3,H,16] bf16 -> [T,H,16] bf16.  f16 qkv with head_dim 16 = the reference's fp16-autocast call site INCLUDING its
    casts: `flash_attn(qkv.to(bfloat16)).to(qkv.dtype)` -- bf16 arithmetic, f16 tensors, the four cast passes (two forward, two
    backward) folded into the kernels' loads and stores; f16 qkv with head_dim 17..64 = f16 OPERANDS (f16 MFMAs, f16 P / dS, fp32
    accumulation: what flash-attn does with LitePT's fp16 tensors, litept_v1.py:259-265).  dropout_p > 0 (head_dim 16): attention dropout; the mask is a function of
    `seed` (default: drawn from torch's CPU generator, so torch.manual_seed reproduces a step) and is regenerated by the backward."""
    if qkv.dtype not in (torch.bfloat16, torch.float16):
        raise PtcoreError("attn_varlen_qkvpacked expects 16-bit operands (the reference casts with .to(torch.bfloat16), ptv3m1:209)")
    if softmax_scale is None:
        softmax_scale = qkv.shape[-1] ** -0.5
    dropout_p = float(dropout_p)
    if not 0.0 <= dropout_p < 1.0:
        raise PtcoreError("dropout_p must lie in [0, 1)")
    if dropout_p > 0.0 and seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())        # CPU generator: no device synchronisation
    return _AttnVarlen.apply(qkv, cu_seqlens, int(max_seqlen), float(softmax_scale), dropout_p, int(seed or 0))


class _AttnRpe(Function):
    @staticmethod
    def forward(ctx, qkv, rpe_table, cu_seqlens, grid_coord, max_seqlen, softmax_scale, pos_bnd):
        out, lse = ops.attn_rpe_fwd(qkv, cu_seqlens, max_seqlen, softmax_scale, grid_coord, rpe_table.float(), pos_bnd)
        ctx.save_for_backward(qkv, out, lse, cu_seqlens, grid_coord, rpe_table)
        ctx.args = (max_seqlen, softmax_scale, pos_bnd)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, out, lse, cu, gc, tab = ctx.saved_tensors
        max_seqlen, scale, pos_bnd = ctx.args
        dqkv, dtab = ops.attn_rpe_bwd(qkv, out, dout.contiguous(), lse, cu, max_seqlen, scale, gc, tab.float(), pos_bnd)
        return dqkv, dtab.to(tab.dtype), None, None, None, None, None


def attn_rpe_qkvpacked(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, softmax_scale: float,
                       grid_coord: torch.Tensor, rpe_table: torch.Tensor, pos_bnd: int) -> torch.Tensor:
    """The reference's dense attention branch with RPE (ptv3m1:29-48,190-206) on the window-attention kernels:
    softmax(scale q k^T + rpe(grid_coord_i - grid_coord_j)) v per window, qkv [T,3,H,16] bf16 or f16 in serialized order,
    grid_coord [T,3] int32 in the same order, rpe_table [3(2B+1), H] (differentiable)."""
    if qkv.dtype not in (torch.bfloat16, torch.float16):
        raise PtcoreError("attn_rpe_qkvpacked expects 16-bit qkv")
    return _AttnRpe.apply(qkv, rpe_table, cu_seqlens, grid_coord, int(max_seqlen), float(softmax_scale), int(pos_bnd))


# ------------------------------------------------------------------------------------------------
# PT-v3m3 Point3DRoPE on packed qkv rows
# ------------------------------------------------------------------------------------------------
class _RopeXYZ(Function):
    """ptc_rope3d_xyz: q / k slabs rotated, v converted, one pass, bf16 out; backward = the inverse rotation of the gradient."""

    @staticmethod
    def forward(ctx, qkv, xyz, inv_freq, out_dtype):
        ctx.save_for_backward(xyz, inv_freq)
        ctx.in_dtype = qkv.dtype
        return ops.rope3d_xyz(qkv.contiguous(), xyz, inv_freq, 2, 1.0, out_dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xyz, inv_freq = ctx.saved_tensors
        return ops.rope3d_xyz(g.contiguous(), xyz, inv_freq, 2, -1.0, ctx.in_dtype), None, None, None


def rope_xyz_torch(qkv: torch.Tensor, xyz: torch.Tensor, inv_freq: torch.Tensor) -> torch.Tensor:
    """The reference's arithmetic (point_transformer_v3m3_utonia.py:58-101,303-323) on the packed layout, in torch ops:
    qkv [n, 3, H, D] of any float dtype -> bf16; q and k are rotated in fp32 (bf16 * fp32 promotes there too), v is only cast."""
    n, _, H, D = qkv.shape
    Q = D // 6
    emb = xyz[:, :, None] * inv_freq[None, None, :]                       # [n, 3, Q]   (:62-67)
    cos, sin = emb.cos()[:, None, None, :, None, :], emb.sin()[:, None, None, :, None, :]   # over (slab, head, axis, half, i)
    t = qkv[:, :2].float().reshape(n, 2, H, 3, 2, Q)
    u, v = t[..., 0:1, :], t[..., 1:2, :]
    rot = torch.cat((u * cos + (-v) * sin, v * cos + u * sin), dim=-2)     # x cos + rotate_half(x) sin  (:75-77,91-92)
    return torch.cat((rot.reshape(n, 2, H, D).to(torch.bfloat16), qkv[:, 2:].to(torch.bfloat16)), dim=1)


def rope_xyz_qkvpacked(qkv: torch.Tensor, xyz: torch.Tensor, inv_freq: torch.Tensor, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """qkv [n, 3, H, D] (D % 6 == 0) -> bf16 [n, 3, H, D] with Point3DRoPE applied to q and k: what flash-attn receives at
    point_transformer_v3m3_utonia.py:319-323.  On ptc_rope3d_xyz (one pass over the packed rows, bf16 out); `rope_xyz_torch` is the same
    arithmetic in torch ops, kept as the reference of the parity tests."""
    if qkv.dim() != 4 or qkv.shape[1] != 3 or qkv.shape[3] % 6 != 0:
        raise PtcoreError(f"rope_xyz_qkvpacked: qkv {tuple(qkv.shape)} must be [n, 3, H, D] with D % 6 == 0")
    # out_dtype: bf16 = PT-v3m3's `qkv.to(torch.bfloat16)` in front of flash-attn; LitePT hands over its autocast dtype (f16 under the fp16 recipe)
    return _RopeXYZ.apply(qkv, xyz.float().contiguous(), inv_freq.float().contiguous(), out_dtype)


class _AttnRope(Function):
    @staticmethod
    def forward(ctx, qkv, xyz, inv_freq, cu_seqlens, max_seqlen, softmax_scale):
        out, lse = ops.attn_rope_fwd(qkv, xyz, inv_freq, cu_seqlens, max_seqlen, softmax_scale)
        ctx.save_for_backward(qkv, out, lse, xyz, inv_freq, cu_seqlens)
        ctx.args = (max_seqlen, softmax_scale)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, out, lse, xyz, inv_freq, cu = ctx.saved_tensors
        return ops.attn_rope_bwd(qkv, out, dout.contiguous(), lse, xyz, inv_freq, cu, *ctx.args), None, None, None, None, None


def attn_rope_qkvpacked(qkv: torch.Tensor, xyz: torch.Tensor, inv_freq: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int,
                        softmax_scale: Optional[float] = None, operand_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """Point3DRoPE / PointROPE on q and k, then flash_attn_varlen_qkvpacked_func (point_transformer_v3m3_utonia.py:303-323,353-359;
    litept_v1.py:239-265): qkv [n, 3, H, D] UN-rotated -> [n, H, D] in `operand_dtype`.  head_dim 18 with qkv already in the operand
    dtype: ONE operator -- the rotation happens in the attention kernels' prologue and its inverse in their backward epilogue
    (csrc/attention_hd.h ROPE; no rotated copy of q / k in memory).  Anything else (other head dims, an fp16 qkv that the call site
    casts to bf16 AFTER the rotation): the rotation pass ptc_rope3d_xyz followed by the attention kernels -- the same arithmetic."""
    if softmax_scale is None:
        softmax_scale = qkv.shape[-1] ** -0.5
    if (qkv.dtype == operand_dtype and operand_dtype in (torch.bfloat16, torch.float16) and qkv.dim() == 4
            and ops.attn_rope_supported(int(qkv.shape[3]), int(max_seqlen))):
        return _AttnRope.apply(qkv, xyz.float().contiguous(), inv_freq.float().contiguous(), cu_seqlens, int(max_seqlen), float(softmax_scale))
    return attn_varlen_qkvpacked(rope_xyz_qkvpacked(qkv, xyz, inv_freq, operand_dtype), cu_seqlens, max_seqlen, softmax_scale)


# ------------------------------------------------------------------------------------------------
# cross entropy
# ------------------------------------------------------------------------------------------------
class _CrossEntropy(Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        loss_sum, count, lse = ops.cross_entropy_fwd(logits, target, ignore_index)
        ctx.save_for_backward(logits, target, lse, count)
        ctx.ignore_index = ignore_index
        return loss_sum / count          # nan when nothing is counted, as nn.CrossEntropyLoss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, target, lse, count = ctx.saved_tensors
        d = ops.cross_entropy_bwd(logits, target, lse, g.float() / count, ctx.ignore_index)
        return d, None, None


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = -1) -> torch.Tensor:
    """nn.CrossEntropyLoss(ignore_index=ignore_index) (mean over counted points) on seg logits [N, C]
    (pointcept/models/losses/misc.py, pointcept/models/default.py:78-84): one forward and one backward
    kernel, fp32 log-sum-exp straight from the head's (bf16, possibly strided) output."""
    if logits.dim() != 2:
        raise PtcoreError("cross_entropy expects [N, C] logits")
    return _CrossEntropy.apply(logits, target, int(ignore_index))


class _LovaszSoftmax(Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        loss, dlogits = ops.lovasz_softmax(logits, target, ignore_index)
        ctx.save_for_backward(dlogits)
        ctx.dtype = logits.dtype
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return (dlogits * g.float()).to(ctx.dtype), None, None


def lovasz_softmax(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = -1) -> torch.Tensor:
    """LovaszLoss(mode="multiclass", ignore_index=ignore_index) on seg logits [N, C]
    (pointcept/models/losses/lovasz.py:209-260; second criterion of scannet/semseg-pt-v3m1-0-base.py:49-52):
    softmax, per-class errors, ONE segmented sort, exact Jaccard steps and the gradient, all on device."""
    if logits.dim() != 2:
        raise PtcoreError("lovasz_softmax expects [N, C] logits")
    return _LovaszSoftmax.apply(logits, target, int(ignore_index))


# ------------------------------------------------------------------------------------------------
# BatchNorm1d + activation
# ------------------------------------------------------------------------------------------------
class _BatchNormAct(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, act, res):
        # dense rows for the kernels -- AND for the tensor the backward reads: a Linear / conv output with a channel count that is not a
        # multiple of 16 arrives as a column slice of the padded GEMM output (36 of 48 columns: LitePT; found on hardware in round 3)
        x = x.contiguous()
        if res is not None:
            res = res.contiguous()
        y, mean, rstd = ops.batch_norm_act_fwd(x, weight, bias, running_mean, running_var, training, momentum, eps, act, res=res)
        ctx.save_for_backward(x, weight, bias, mean, rstd, res)
        ctx.training, ctx.act = training, act
        ctx.mark_non_differentiable()
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, bias, mean, rstd, res = ctx.saved_tensors
        out = ops.batch_norm_act_bwd(dy, x, weight, bias, mean, rstd, ctx.training, ctx.act, want_affine=weight is not None, res=res)
        dx, dg, db = out[:3]
        if weight is not None:
            dg, db = dg.to(weight.dtype), db.to(weight.dtype)
        return dx, dg, db, None, None, None, None, None, None, (out[3] if res is not None else None)


def batch_norm_act(x: torch.Tensor, weight, bias, running_mean, running_var, training: bool, momentum: float, eps: float,
                   act: str = "none", residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(F.batch_norm(x, ...) [+ residual]) over the rows of [N, C] (act in {"none", "gelu", "relu"}), statistics in
    fp32/fp64, output in x's dtype; backward recomputes the pre-activation (nothing but x and the residual is saved).  With
    `residual` (x's shape and dtype): the tail of a residual block, spconv_unet_v1m1_base.py:79-83, in the BatchNorm's apply pass."""
    return _BatchNormAct.apply(x, weight, bias, running_mean, running_var, bool(training), float(momentum), float(eps), act, residual)


# ------------------------------------------------------------------------------------------------
# one PT-v3m1 Block = ONE autograd Function = one C call per direction (csrc/block_exec.hip)
# ------------------------------------------------------------------------------------------------
_BLK_PARAMS = ("W_CONV", "B_CONV", "W_LIN", "B_LIN", "G_CPE", "BE_CPE", "G_N1", "BE_N1", "W_QKV", "B_QKV", "W_PROJ", "B_PROJ", "G_N2",
               "BE_N2", "W_FC1", "B_FC1", "W_FC2", "B_FC2")
_BLK_WEIGHTS = ("W_CONV", "W_LIN", "W_QKV", "W_PROJ", "W_FC1", "W_FC2")


def _blk_layout(items, elem_bytes):
    """[(name, rows, cols)] -> ({name: (byte offset, rows, cols)}, total elements): consecutive 256-byte aligned pieces of one allocation"""
    unit, off, out = 256 // elem_bytes, 0, {}
    for name, rows, cols in items:
        out[name] = (off * elem_bytes, int(rows), int(cols))
        off += (int(rows) * int(cols) + unit - 1) // unit * unit
    return out, max(off, 1)


_blk_plans = {}


def _blk_plan(n, npad, c, heads, x0_f32, dt=torch.bfloat16):
    """pointer-table layout of one Block call, cached per shape: byte offsets of every saved activation / scratch gradient inside
    the slabs, as (table index, offset) lists -- the per-call work is then two allocations and ~45 integer stores per direction"""
    # round 6: where the executor runs the MLP on csrc/mlp.hip (C = 32 | 64) the hidden-width tensors do not exist: no H / ACT / M in the
    # saved slab (2 x N x 4C x 2 bytes per Block: 0.84 GB at N = 819200, C = 64), no DH in the backward's scratch
    fused = bool(ops.lib().ptc_ptv3_block_mlp_fused(int(c), _lib.PTC_F16 if dt == torch.float16 else _lib.PTC_BF16))
    key = (n, npad, c, heads, x0_f32, fused)
    p = _blk_plans.get(key)
    if p is not None:
        return p
    E = _lib.block_enums()
    hid = 4 * c
    l16, t16 = _blk_layout([("CONV", n, c), ("LIN", n, c), ("Y1", n, c), ("QKV", npad, 3 * c), ("ATT", npad, c), ("A", n, c), ("Y2", n, c)] +
                           ([] if fused else [("H", n, hid), ("ACT", n, hid), ("M", n, c)]), 2)
    l32, t32 = _blk_layout([("X1", n, c), ("X2", n, c), ("ST_CPE", 2, n), ("ST_N1", 2, n), ("ST_N2", 2, n), ("LSE", heads, npad)], 4)
    shapes = {"W_CONV": (c, 27, c), "B_CONV": (c,), "W_LIN": (c, c), "B_LIN": (c,), "G_CPE": (c,), "BE_CPE": (c,), "G_N1": (c,), "BE_N1": (c,),
              "W_QKV": (3 * c, c), "B_QKV": (3 * c,), "W_PROJ": (c, c), "B_PROJ": (c,), "G_N2": (c,), "BE_N2": (c,), "W_FC1": (hid, c), "B_FC1": (hid,),
              "W_FC2": (c, hid), "B_FC2": (c,)}
    # the parameter gradients in their OWN small slab (~12 C^2 floats): AccumulateGrad keeps the returned views -- and with them their
    # storage -- alive as long as .grad lives; sharing a slab with the 3 N C floats of fp32 scratch pinned ~77 MB per stage-0 Block
    # (ADVICE r3).  The fp32 scratch gradients (and dx0 of an fp32 stream) live in a second, transient slab.
    lg, tg = _blk_layout([("G_" + k, 1, int(torch.Size(shapes[k]).numel())) for k in _BLK_PARAMS], 4)
    gs_items = [("S_DX2", n, c), ("S_DX1", n, c)]
    if x0_f32:
        gs_items.append(("G_X0", n, c))
    lgs, tgs = _blk_layout(gs_items, 4)
    s_items = [("S_DM", n, c)] + ([] if fused else [("S_DH", n, hid)]) + [("S_DY2", n, c), ("S_DA", n, c), ("S_DATT", npad, c), ("S_DQKV", npad, 3 * c), ("S_DY1", n, c),
               ("S_DLIN", n, c), ("S_DCONV", n, c), ("G_XC", n, c)]
    if not x0_f32:
        s_items.append(("G_X0", n, c))
    ls, ts = _blk_layout(s_items, 2)
    p = dict(E=E, t16=t16, t32=t32, tg=tg, tgs=tgs, ts=ts,
             o16=[(E["O_" + k], v[0]) for k, v in l16.items()], o32=[(E["O_" + k], v[0]) for k, v in l32.items()],
             g32=[(E[k], v[0]) for k, v in lg.items()], g32s=[(E[k], v[0]) for k, v in lgs.items()], g16=[(E[k], v[0]) for k, v in ls.items()],
             gparam={k: (lg["G_" + k][0] // 4, int(torch.Size(shapes[k]).numel())) for k in _BLK_PARAMS},
             gx0=(lgs["G_X0"][0] // 4 if x0_f32 else ls["G_X0"][0] // 2), gxc=ls["G_XC"][0] // 2,
             ws=None)
    if len(_blk_plans) > 256:
        _blk_plans.clear()
    _blk_plans[key] = p
    return p


def _blk_tables(E, x0, meta):
    import ctypes

    n, c = x0.shape
    blk = meta["blk"]
    iv = (ctypes.c_int64 * E["I_COUNT"])()
    iv[E["I_ABI"]], iv[E["I_N"]], iv[E["I_NPAD"]], iv[E["I_NSEQ"]], iv[E["I_C"]] = E["ABI"], n, meta["n_pad"], meta["n_seq"], c
    iv[E["I_HEADS"]], iv[E["I_DTYPE"]], iv[E["I_A_DTYPE"]], iv[E["I_PATCH"]] = (meta["heads"], _lib.PTC_F16 if meta["dt"] == torch.float16 else _lib.PTC_BF16,
                                                                              _lib.dtype_code(x0), meta["patch"])
    iv[E["I_BLK_BM"]], iv[E["I_BLK_HCAP"]] = (0, 0) if blk is None else (blk.bm, blk.hcap)
    fv = (ctypes.c_float * E["F_COUNT"])()
    fv[E["F_SCALE"]], fv[E["F_EPS_CPE"]], fv[E["F_EPS_N1"]], fv[E["F_EPS_N2"]] = meta["scale"], meta["eps_cpe"], meta["eps_n1"], meta["eps_n2"]
    pin = (ctypes.c_void_p * E["P_COUNT"])()
    tabs = meta["tabs"]
    pin[E["P_NBR"]], pin[E["P_CU"]] = meta["nbr"].data_ptr(), meta["cu"].data_ptr()
    pin[E["P_T_QKV_FWD"]], pin[E["P_T_QKV_BWD"]] = tabs[0].data_ptr(), tabs[1].data_ptr()
    pin[E["P_T_PROJ_FWD"]], pin[E["P_T_PROJ_BWD"]] = tabs[2].data_ptr(), tabs[3].data_ptr()
    if blk is not None:
        pin[E["P_BLK_TAB"]], pin[E["P_BLK_HID"]], pin[E["P_BLK_HCNT"]] = blk.tab.data_ptr(), blk.hid.data_ptr(), blk.hcnt.data_ptr()
        if config.WGRAD_BLK:      # (NULL: the executor's convolution weight gradient stays on the global-gather kernel)
            pin[E["P_BLK_NOVF"]] = blk.n_overflow.data_ptr()
    return iv, fv, pin


class _BlockFn(Function):
    """(x3, xb3) = Block(x0, xc): the residual stream after the block (fp32) and its bf16 copy (the operand of the next
    convolution / Linear).  `meta`: sizes, eps, tables (python object, not differentiated); `params`: the block's 18 parameter
    tensors in _BLK_PARAMS order (qkv bias may be None)."""

    @staticmethod
    def forward(ctx, x0, xc, rs1, rs2, meta, *params):
        dt = meta["dt"]            # the autocast dtype of the step: bf16, or f16 (the reference's fp16 recipe)
        n, c = x0.shape
        npad, heads = meta["n_pad"], meta["heads"]
        dev = x0.device
        x0 = x0.contiguous()
        xc = xc.contiguous()
        plan = _blk_plan(n, npad, c, heads, x0.dtype == torch.float32, dt)
        E = plan["E"]
        iv, fv, pin = _blk_tables(E, x0, meta)
        keep = []                                            # tensors the pointer table names (alive until saved / returned)
        for k, p in zip(_BLK_PARAMS, params):
            if p is None:
                continue
            if k in _BLK_WEIGHTS:                            # 16-bit shadows [c_out][taps][c_in] (conv) / [c_out][c_in] (Linear)
                t = _cast_cache.get(p.reshape(p.shape[0], -1, p.shape[-1]) if p.dim() == 5 else p, dt)
            else:
                t = p if p.dtype == torch.float32 else p.float()
            if not t.is_contiguous():
                t = t.contiguous()
            keep.append(t)
            pin[E["P_" + k]] = t.data_ptr()
        pin[E["P_X0"]], pin[E["P_XC"]] = x0.data_ptr(), xc.data_ptr()
        if rs1 is not None:
            pin[E["P_RS1"]] = rs1.data_ptr()
        if rs2 is not None:
            pin[E["P_RS2"]] = rs2.data_ptr()
        buf16 = torch.empty(plan["t16"], dtype=dt, device=dev)
        buf32 = torch.empty(plan["t32"], dtype=torch.float32, device=dev)
        x3 = torch.empty((n, c), dtype=torch.float32, device=dev)
        xb3 = torch.empty((n, c), dtype=dt, device=dev)
        import ctypes

        pout = (ctypes.c_void_p * E["O_COUNT"])()
        b16, b32 = buf16.data_ptr(), buf32.data_ptr()
        for idx, off in plan["o16"]:
            pout[idx] = b16 + off
        for idx, off in plan["o32"]:
            pout[idx] = b32 + off
        pout[E["O_X3"]], pout[E["O_XB3"]] = x3.data_ptr(), xb3.data_ptr()
        _lib.check(ops.lib().ptc_ptv3_block_fwd(iv, fv, pin, pout, ops.stream_ptr()), "ptc_ptv3_block_fwd")
        ctx.save_for_backward(x0, xc, rs1, rs2, buf16, buf32, x3, xb3, *keep)
        ctx.meta, ctx.plan = meta, plan
        ctx.present = [p is not None for p in params]
        ctx.param_dtypes = [None if p is None else p.dtype for p in params]
        ctx.param_shapes = [None if p is None else p.shape for p in params]
        ctx.set_materialize_grads(False)
        return x3, xb3

    @staticmethod
    @once_differentiable
    def backward(ctx, dz3, dyb3):
        import ctypes

        n_par = len(_BLK_PARAMS)
        if dz3 is None and dyb3 is None:
            return (None,) * (5 + n_par)
        sv = ctx.saved_tensors
        x0, xc, rs1, rs2, buf16, buf32, x3, xb3 = sv[:8]
        meta, plan = ctx.meta, ctx.plan
        E = plan["E"]
        dt = meta["dt"]
        n, c = x0.shape
        npad, heads = meta["n_pad"], meta["heads"]
        dev = x0.device
        iv, fv, pin = _blk_tables(E, x0, meta)
        it = iter(sv[8:])
        sh = {}
        keep_wt = []
        for k, present in zip(_BLK_PARAMS, ctx.present):
            if present:
                t = next(it)
                pin[E["P_" + k]] = t.data_ptr()
                if k in _BLK_WEIGHTS:
                    sh[k] = t
        # transposed weight layouts of the input-gradient GEMMs (all layers' layouts are refreshed in one launch per step)
        for k, mode, slots in (("CONV", "mirror", 0), ("LIN", "mirror", 0), ("QKV", "repeat", 2), ("PROJ", "mirror", 0), ("FC1", "mirror", 0),
                               ("FC2", "mirror", 0)):
            w = sh["W_" + k]
            wt = _cast_cache.layout(w, mode, slots)
            if wt is None:
                # the shadow is not (or no longer) a cache entry -- a non-leaf / re-parametrised weight, or the cache was invalidated
                # between forward and backward: build this one layout here (the composed path does the same), never fail the step
                w3 = w if w.dim() == 3 else w[:, None, :]
                wt = (w3.flip(1).permute(2, 1, 0) if mode == "mirror" else w3.permute(2, 1, 0).expand(-1, slots, -1)).contiguous()
                keep_wt.append(wt)
            pin[E["P_WT_" + k]] = wt.data_ptr()
        pin[E["P_X0"]], pin[E["P_XC"]] = x0.data_ptr(), xc.data_ptr()
        if rs1 is not None:
            pin[E["P_RS1"]] = rs1.data_ptr()
        if rs2 is not None:
            pin[E["P_RS2"]] = rs2.data_ptr()
        dz3c = None if dz3 is None else (dz3 if dz3.dtype == torch.float32 and dz3.is_contiguous() else dz3.float().contiguous())
        dyb3c = None if dyb3 is None else (dyb3 if dyb3.dtype == dt and dyb3.is_contiguous() else dyb3.to(dt).contiguous())
        if dz3c is not None:
            pin[E["P_DZ3"]] = dz3c.data_ptr()
        if dyb3c is not None:
            pin[E["P_DYB3"]] = dyb3c.data_ptr()
        psv = (ctypes.c_void_p * E["O_COUNT"])()
        b16, b32 = buf16.data_ptr(), buf32.data_ptr()
        for idx, off in plan["o16"]:
            psv[idx] = b16 + off
        for idx, off in plan["o32"]:
            psv[idx] = b32 + off
        psv[E["O_X3"]], psv[E["O_XB3"]] = x3.data_ptr(), xb3.data_ptr()
        gbuf = torch.empty(plan["tg"], dtype=torch.float32, device=dev)          # parameter gradients only (small, long-lived)
        gscr = torch.empty(plan["tgs"], dtype=torch.float32, device=dev)         # fp32 scratch gradients (+ dx0 of an fp32 stream)
        sbuf = torch.empty(plan["ts"], dtype=dt, device=dev)
        pg = (ctypes.c_void_p * E["GS_COUNT"])()
        bg, bgs, bs = gbuf.data_ptr(), gscr.data_ptr(), sbuf.data_ptr()
        for idx, off in plan["g32"]:
            pg[idx] = bg + off
        for idx, off in plan["g32s"]:
            pg[idx] = bgs + off
        for idx, off in plan["g16"]:
            pg[idx] = bs + off
        for k, present in zip(_BLK_PARAMS, ctx.present):
            if not present:
                pg[E["G_" + k]] = None
        if plan["ws"] is None:
            plan["ws"] = int(ops.lib().ptc_ptv3_block_workspace_bytes(n, npad, c, heads))
        nbytes = plan["ws"]
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(ops.lib().ptc_ptv3_block_bwd(iv, fv, pin, psv, pg, ws.data_ptr(), nbytes, ops.stream_ptr()), "ptc_ptv3_block_bwd")
        dx0 = (gscr if x0.dtype == torch.float32 else sbuf)[plan["gx0"]:plan["gx0"] + n * c].view(n, c)
        dxc = sbuf[plan["gxc"]:plan["gxc"] + n * c].view(n, c)
        grads = []
        for k, present, pdt, shp in zip(_BLK_PARAMS, ctx.present, ctx.param_dtypes, ctx.param_shapes):
            if not present:
                grads.append(None)
                continue
            off, cnt = plan["gparam"][k]
            g = gbuf[off:off + cnt].view(shp)
            grads.append(g if pdt == torch.float32 else g.to(pdt))
        return (dx0, dxc, None, None, None, *grads)


def ptv3_block(x0, xc, rs1, rs2, meta, params):
    """One PT-v3m1 Block through csrc/block_exec.hip -> (x3 fp32, xb3 bf16).  See point_transformer_v3.Block._forward_exec."""
    return _BlockFn.apply(x0, xc, rs1, rs2, meta, *params)


# ------------------------------------------------------------------------------------------------
# MLP (fc1 -> GELU -> fc2) with the activation fused into the GEMM epilogues
# ------------------------------------------------------------------------------------------------
class _MLP(Function):
    """out = GELU(x W1^T + b1) W2^T + b2 (ptv3m1:225-248, dropout p = 0).  GELU rides in fc1's epilogue (h and
    GELU(h) leave the kernel together) and GELU' in the epilogue of fc2's input gradient (the gradient reaches
    HBM already multiplied through the activation): the two elementwise kernels of the unfused form, and the
    dA tensor, disappear.  Weight gradients run on the split-K kernels as in `linear`."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        dt = torch.get_autocast_dtype("cuda") if _autocast_on() else x.dtype
        xp = x.to(dt).contiguous()
        w1c, w2c = _cast_cache.get(w1, dt).contiguous(), _cast_cache.get(w2, dt).contiguous()
        ctx.dtypes = (x.dtype, w1.dtype, None if b1 is None else b1.dtype, w2.dtype, None if b2 is None else b2.dtype)
        ctx.one_kernel = config.MLP_ONE_KERNEL and w1.shape[0] == 4 * w1.shape[1] and w2.shape[0] == w1.shape[1] and ops.mlp_supported(w1.shape[1], dt)
        if ctx.one_kernel:      # round 6 (csrc/mlp.hip): the hidden tensor never reaches memory, the backward recomputes it; same bits as below
            ctx.save_for_backward(xp, w1c, w2c, b1)
            return ops.mlp_fwd(xp, w1c, b1, w2c, b2)
        h, a = ops.linear_gelu_fwd(xp, w1c, b1)
        out = ops.spconv_fwd(a, w2c[:, None, :], None if b2 is None else b2.float(), None)      # shapes: mlp_gelu_supported
        ctx.save_for_backward(xp, h, a, w1c, w2c)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x_dt, w1_dt, b1_dt, w2_dt, b2_dt = ctx.dtypes
        if ctx.one_kernel:
            xp, w1c, w2c, b1 = ctx.saved_tensors
            w2t = _cast_cache.layout(w2c, "mirror")
            dx, dw1, db1, dw2, db2 = ops.mlp_bwd(dout.to(xp.dtype).contiguous(), xp, w1c, b1, w2c.t().contiguous() if w2t is None else w2t[:, 0, :],
                                                 want_b1=b1_dt is not None, want_b2=b2_dt is not None)
            return (dx.to(x_dt) if ctx.needs_input_grad[0] else None, dw1.to(w1_dt), None if db1 is None else db1.to(b1_dt), dw2.to(w2_dt),
                    None if db2 is None else db2.to(b2_dt))
        xp, h, a, w1c, w2c = ctx.saved_tensors
        g = dout.to(xp.dtype).contiguous()
        n = g.shape[0]
        # fc2: weight / bias gradients, then the input gradient THROUGH the activation
        res = ops.spconv_wgrad(a, g, None, want_bias=b2_dt is not None)
        dw2, db2 = res if b2_dt is not None else (res, None)
        dw2 = dw2[:, 0, :]
        dw2 = dw2.to(w2_dt)
        db2 = None if db2 is None else db2.to(b2_dt)
        w2t = _cast_cache.layout(w2c, "mirror")
        dh = ops.linear_gelu_bwd_input(g, w2c.t().contiguous() if w2t is None else w2t[:, 0, :], h)
        # fc1: the same split
        res = ops.spconv_wgrad(xp, dh, None, want_bias=b1_dt is not None)
        dw1, db1 = res if b1_dt is not None else (res, None)
        dw1 = dw1[:, 0, :]
        dw1 = dw1.to(w1_dt)
        db1 = None if db1 is None else db1.to(b1_dt)
        dx = None
        if ctx.needs_input_grad[0]:
            w1t = _cast_cache.layout(w1c, "mirror")
            dx = ops.spconv_fwd(dh, w1c.t().contiguous()[:, None, :] if w1t is None else w1t, None, None).to(x_dt)
        return dx, dw1, db1, dw2, db2


def mlp_gelu_supported(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dim() == 2 and x.shape[0] > 0):
        return False
    dt = torch.get_autocast_dtype("cuda") if _autocast_on() else x.dtype
    hidden, c = w1.shape
    # (fc2 and fc1's input gradient contract over `hidden`: beyond linear2's 256 channels they run on the identity-table kernel, which
    #  wants 32-channel multiples on both sides -- other widths take the unfused Linear -> GELU -> Linear, whose operands are padded)
    return (dt in (torch.bfloat16, torch.float16) and c % 16 == 0 and hidden % 16 == 0 and w2.shape[0] % 16 == 0
            and ops.linear_supported_ex(c, hidden, dt) and ops.linear_supported_ex(w2.shape[0], hidden, dt)
            and _own_gemm(0, hidden, dt, w2.shape[0]) and _own_gemm(0, hidden, dt, c))


def mlp_gelu(x, w1, b1, w2, b2):
    return _MLP.apply(x, w1, b1, w2, b2)
