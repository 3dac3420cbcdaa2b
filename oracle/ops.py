"""TEST INFRASTRUCTURE (oracle).  CPU restatements (numpy / plain torch fp32) of the third-party
operators on the hot path.  spconv, flash_attn and torch_scatter are NOT vendored under
/root/reference (environment.yml:44,47,51, all unpinned) -> "parity unpinned" against the real
libraries; the definitions below follow SURVEY 8(a) A7/A8/A12/A16 and are cross-checked against
dense F.conv3d / F.conv_transpose3d, SDPA and brute-force loops in tests/test_oracle_ops.py.

Conventions (shared with include/ptcore.h):
  * indices [N,4] int32 = (batch, x, y, z); conv weight [C_out, k0, k1, k2, C_in] (spconv 2.x layout),
    (k0,k1,k2) applied to (x,y,z); cross-correlation: out[p] = sum_k W[k] . in[p + (k - r)].
  * gather tables nbr[kv][n_out] int32, -1 = no input; duplicate voxels: lowest row index wins.
"""
from __future__ import annotations

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------
# rulebooks (numpy)
# ------------------------------------------------------------------------------------------------
def _linear_key(b, x, y, z, S):
    return ((b.astype(np.int64) * S + x) * S + y) * S + z


def subm_rulebook(indices: np.ndarray, ksize: int) -> np.ndarray:
    """nbr[k][i] = row whose coordinate is coord_i + delta_k (same batch), k lexicographic in
    (d0,d1,d2), or -1.  Sort + searchsorted; lowest row index wins among duplicates."""
    ind = np.asarray(indices, dtype=np.int64)
    n = ind.shape[0]
    r = ksize // 2
    S = int(ind[:, 1:].max()) + 2 * r + 2 if n else 1
    key = _linear_key(ind[:, 0], ind[:, 1] + r, ind[:, 2] + r, ind[:, 3] + r, S)
    ukeys, first = np.unique(key, return_index=True)  # first occurrence = lowest index
    nbr = np.full((ksize ** 3, n), -1, dtype=np.int32)
    k = 0
    for d0 in range(-r, r + 1):
        for d1 in range(-r, r + 1):
            for d2 in range(-r, r + 1):
                q = _linear_key(ind[:, 0], ind[:, 1] + r + d0, ind[:, 2] + r + d1, ind[:, 3] + r + d2, S)
                valid = (ind[:, 1] + d0 >= 0) & (ind[:, 2] + d1 >= 0) & (ind[:, 3] + d2 >= 0)
                pos = np.searchsorted(ukeys, q)
                pos = np.minimum(pos, len(ukeys) - 1)
                hit = valid & (ukeys[pos] == q)
                nbr[k, hit] = first[pos[hit]].astype(np.int32)
                k += 1
    return nbr


def _spread3(v: np.ndarray) -> np.ndarray:
    """bit i of v (21 bits) -> bit 3 i"""
    out = np.zeros_like(v, dtype=np.int64)
    for i in range(21):
        out |= ((v >> i) & 1) << (3 * i)
    return out


def down_rulebook(indices: np.ndarray):
    """k=2 s=2: coarse sites = unique (b, x>>1, y>>1, z>>1), numbered by ascending (b, Morton code of the coarse coordinate: bit i of
    x at 3 i + 2, of y at 3 i + 1, of z at 3 i).  spconv numbers its output sites in hash-insertion order (SparseConv3d behind
    spconv_unet_v1m1_base.py:137-144): nothing downstream may depend on the numbering, so the restatement picks the one whose
    consecutive rows are spatial neighbours (round 6; before: the lexicographic key).
    Returns out_indices [n_out,4], out_of_in [n_in], nbr_down [8][n_out], nbr_up [8][n_in]."""
    ind = np.asarray(indices, dtype=np.int64)
    n = ind.shape[0]
    c = np.stack([ind[:, 0], ind[:, 1] >> 1, ind[:, 2] >> 1, ind[:, 3] >> 1], axis=1)
    assert n == 0 or (int(c[:, 1:].max()) < (1 << 16) and int(c[:, 0].max()) < (1 << 15))
    key = (c[:, 0].astype(np.int64) << 48) | (_spread3(c[:, 1]) << 2) | (_spread3(c[:, 2]) << 1) | _spread3(c[:, 3])
    ukeys, first, out_of_in = np.unique(key, return_index=True, return_inverse=True)
    n_out = len(ukeys)
    out_indices = c[first].astype(np.int32)
    k = ((ind[:, 1] & 1) << 2) | ((ind[:, 2] & 1) << 1) | (ind[:, 3] & 1)
    nbr_down = np.full((8, n_out), -1, dtype=np.int32)
    for j in range(n - 1, -1, -1):  # descending so that the lowest row index is written last
        nbr_down[k[j], out_of_in[j]] = j
    nbr_up = np.full((8, n), -1, dtype=np.int32)
    nbr_up[k, np.arange(n)] = out_of_in.astype(np.int32)
    return out_indices, out_of_in.astype(np.int32), nbr_down, nbr_up


# ------------------------------------------------------------------------------------------------
# gather-table convolution (torch, differentiable through plain autograd)
# ------------------------------------------------------------------------------------------------
def gather_conv(feat: torch.Tensor, weight: torch.Tensor, bias, nbr) -> torch.Tensor:
    """out[o] = bias + sum_k W[:,k,:] . feat[nbr[k][o]]   (weight [C_out, k0,k1,k2, C_in] or [C_out,kv,C_in])."""
    nbr_t = torch.as_tensor(np.asarray(nbr), dtype=torch.long)
    kv, n_out = nbr_t.shape
    c_out, c_in = weight.shape[0], weight.shape[-1]
    w = weight.reshape(c_out, kv, c_in)
    fpad = torch.cat([feat, feat.new_zeros(1, c_in)], dim=0)  # row -1 -> zeros
    out = feat.new_zeros(n_out, c_out)
    for k in range(kv):
        idx = nbr_t[k]
        if bool((idx >= 0).any()):
            out = out + fpad[idx] @ w[:, k, :].t()
    if bias is not None:
        out = out + bias
    return out


# ------------------------------------------------------------------------------------------------
# segment_csr  (torch_scatter.segment_csr semantics; ptv3m1:416-421)
# ------------------------------------------------------------------------------------------------
def segment_csr(src: torch.Tensor, indptr: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
    """CSR segmented reduce over dim 0.  max/min: gradient goes to the FIRST arg-max row
    (differentiable: implemented with gather of the arg rows)."""
    indptr = torch.as_tensor(indptr, dtype=torch.long)
    n_seg = indptr.numel() - 1
    counts = indptr[1:] - indptr[:-1]
    seg_id = torch.repeat_interleave(torch.arange(n_seg), counts)
    c = src.shape[1:]
    if reduce in ("sum", "mean"):
        out = src.new_zeros((n_seg,) + tuple(c)).index_add(0, seg_id, src[: seg_id.numel()])
        if reduce == "mean":
            out = out / counts.clamp(min=1).to(src.dtype).reshape((-1,) + (1,) * len(c))
        return out
    # max / min with first-arg semantics
    with torch.no_grad():
        s = src.detach()[: seg_id.numel()]
        flat = s.reshape(s.shape[0], -1)
        fill = float("-inf") if reduce == "max" else float("inf")
        best = torch.full((n_seg, flat.shape[1]), fill, dtype=flat.dtype)
        best = best.scatter_reduce(0, seg_id[:, None].expand_as(flat), flat, "amax" if reduce == "max" else "amin")
        is_best = flat == best[seg_id]
        rows = torch.arange(flat.shape[0])[:, None].expand_as(flat)
        big = flat.shape[0]
        cand = torch.where(is_best, rows, torch.full_like(rows, big))
        arg = torch.full((n_seg, flat.shape[1]), big, dtype=torch.long).scatter_reduce(
            0, seg_id[:, None].expand_as(flat), cand, "amin")
        empty = counts == 0
        arg[empty] = 0
    flat_src = src.reshape(src.shape[0], -1)
    out = torch.gather(flat_src, 0, arg.clamp(max=max(flat_src.shape[0] - 1, 0)))
    out = torch.where(empty[:, None], torch.zeros_like(out), out)
    return out.reshape((n_seg,) + tuple(c))


class _AttnKernelRounding(torch.autograd.Function):
    """attention_varlen with the 16-bit roundings of the gfx950 kernels (csrc/attention.hip) placed where THEY round,
    forward and backward -- used to show that the engine-vs-oracle gradient differences are those roundings and nothing
    else (tests/test_gpu_fullsize.py).  fp32 math otherwise.
      forward : P = exp(s - max) rounded to bf16 BEFORE the P V product; the denominator is the sum of the same rounded
                values (it comes out of the same MFMA); output rounded to bf16 (flash-attn returns bf16 too).
      backward: P recomputed in fp32 from lse; delta = rowsum(dO * O) with the bf16 output; dV = bf16(P)^T dO;
                dS = P * (dP - delta) rounded to bf16 before dQ = dS K and dK = dS^T Q; dQ / dK / dV rounded to bf16."""

    @staticmethod
    def forward(ctx, qkv, cu, scale):
        T, _, H, D = qkv.shape
        out = qkv.new_zeros(T, H, D)
        lse = qkv.new_zeros(H, T)
        for a, b in zip(cu[:-1], cu[1:]):
            q, k, v = (qkv[a:b, j].transpose(0, 1).float() for j in range(3))
            s = (q * scale) @ k.transpose(1, 2)
            m = s.max(dim=-1, keepdim=True).values
            p16 = torch.exp(s - m).to(torch.bfloat16).float()
            l = p16.sum(-1, keepdim=True)
            out[a:b] = ((p16 @ v) / l).transpose(0, 1).to(torch.bfloat16).float()
            lse[:, a:b] = (m + torch.log(l))[..., 0]
        ctx.save_for_backward(qkv, out, lse)
        ctx.cu, ctx.scale = cu, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        dqkv = torch.zeros_like(qkv)
        do16 = dout.to(torch.bfloat16).float()
        for a, b in zip(ctx.cu[:-1], ctx.cu[1:]):
            q, k, v = (qkv[a:b, j].transpose(0, 1).float() for j in range(3))
            do, o = do16[a:b].transpose(0, 1), out[a:b].transpose(0, 1)
            s = (q * ctx.scale) @ k.transpose(1, 2)
            p = torch.exp(s - lse[:, a:b, None])
            delta = (do * o).sum(-1, keepdim=True)
            dp = do @ v.transpose(1, 2)
            ds16 = (p * (dp - delta)).to(torch.bfloat16).float()
            p16 = p.to(torch.bfloat16).float()
            dqkv[a:b, 0] = ((ds16 @ k) * ctx.scale).transpose(0, 1).to(torch.bfloat16).float()
            dqkv[a:b, 1] = ((ds16.transpose(1, 2) @ q) * ctx.scale).transpose(0, 1).to(torch.bfloat16).float()
            dqkv[a:b, 2] = (p16.transpose(1, 2) @ do).transpose(0, 1).to(torch.bfloat16).float()
        return dqkv, None, None


def attention_varlen_kernel_rounding(qkv: torch.Tensor, cu_seqlens, softmax_scale: float) -> torch.Tensor:
    """see _AttnKernelRounding; qkv holds bf16-representable values (callers round first, ptv3m1:209)"""
    return _AttnKernelRounding.apply(qkv, [int(v) for v in cu_seqlens], float(softmax_scale))


# ------------------------------------------------------------------------------------------------
# variable-length attention (flash_attn.flash_attn_varlen_qkvpacked_func semantics; ptv3m1:208-214)
# ------------------------------------------------------------------------------------------------
def attn_dropout_keep(seed: int, unit: int, lq: int, lk: int, p: float) -> torch.Tensor:
    """keep mask [lq, lk] (bool) of attention dropout inside the (sequence, head) unit `unit` = sequence * H + head: the integer hash
    of pointcept_amd/csrc/attention_drop.h (ad_unit_key / ad_keep) restated in numpy uint32 arithmetic.  TEST INFRASTRUCTURE: the
    random stream of attention dropout is an implementation detail of the attention library (flash-attn's Philox stream in the
    reference); parity of the kernels is stated against softmax-then-drop with THIS mask."""
    M = 0xFFFFFFFF
    seed_lo, seed_hi = seed & M, (seed >> 32) & M
    h = (seed_lo ^ ((unit * 0x9E3779B1) & M)) & M
    h ^= h >> 16; h = (h * 0x7FEB352D) & M; h ^= h >> 15; h = (h * 0x846CA68B) & M; h ^= h >> 16
    h ^= (seed_hi * 0xC2B2AE3D) & M
    h ^= h >> 15; h = (h * 0x2C1B3C6D) & M; h ^= h >> 12
    uk = np.uint64(h)
    q = np.arange(lq, dtype=np.uint64)[:, None]
    k = np.arange(lk, dtype=np.uint64)[None, :]
    m64 = np.uint64(M)
    x = (((q << np.uint64(10)) | k) ^ uk) & m64
    x = (x * np.uint64(0x9E3779B1)) & m64; x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x85EBCA77)) & m64; x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE3D)) & m64; x ^= x >> np.uint64(16)
    t = float(np.float32(p)) * 4294967296.0          # the C side computes the threshold from the float it is handed
    thresh = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    return torch.from_numpy(x >= np.uint64(thresh))


def attention_varlen(qkv: torch.Tensor, cu_seqlens, softmax_scale: float, return_lse: bool = False, dropout_p: float = 0.0, seed: int = 0):
    """qkv [T,3,H,D]; per sequence [cu[i],cu[i+1]) and head: softmax(scale q k^T) v, non-causal,
    fp32 math on the given values (callers round qkv to bf16 first to mirror ptv3m1:209).  dropout_p > 0: flash-attn's attention
    dropout (softmax over all keys, then drop + rescale by 1 / (1 - p); lse of the undropped scores) with the mask of attn_dropout_keep."""
    cu = [int(v) for v in cu_seqlens]
    T, _, H, D = qkv.shape
    outs, lses = [], []
    for si, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
        q, k, v = (qkv[a:b, j].transpose(0, 1).float() for j in range(3))  # [H, L, D]
        s = (q * softmax_scale) @ k.transpose(1, 2)
        lses.append(torch.logsumexp(s, dim=-1))  # [H, L]
        if dropout_p > 0.0:
            pm = torch.softmax(s, dim=-1)
            keep = torch.stack([attn_dropout_keep(seed, si * H + h, b - a, b - a, dropout_p) for h in range(H)]).to(pm.dtype)
            outs.append(((pm * keep / (1.0 - float(np.float32(dropout_p)))) @ v).transpose(0, 1))
            continue
        outs.append((torch.softmax(s, dim=-1) @ v).transpose(0, 1))  # [L, H, D]
    out = torch.cat(outs, dim=0) if outs else qkv.new_zeros(0, H, D).float()
    if return_lse:
        return out, torch.cat(lses, dim=1) if lses else qkv.new_zeros(H, 0).float()
    return out
