"""CPU tests of the ORACLE itself (not gpu): the third-party operators it restates (spconv,
flash_attn, torch_scatter) are absent from /root/reference, so they are pinned here against
independent known-answer constructions: dense F.conv3d / F.conv_transpose3d on small grids, SDPA,
brute-force loops.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ops as oops


def _random_voxels(seed, n, extent, batch=2):
    rng = np.random.default_rng(seed)
    c = rng.integers(0, extent, size=(n * 2, 3))
    b = rng.integers(0, batch, size=(n * 2, 1))
    ind = np.unique(np.concatenate([b, c], axis=1), axis=0)
    ind = ind[rng.permutation(len(ind))][:n]
    return ind.astype(np.int32)


def _dense(feat, ind, shape, batch):
    d = torch.zeros(batch, feat.shape[1], *shape, dtype=feat.dtype)
    d[ind[:, 0], :, ind[:, 1], ind[:, 2], ind[:, 3]] = feat
    return d


def test_subm_conv_matches_dense_conv3d():
    for ks in (3, 5):
        ind = _random_voxels(ks, 400, 9)
        li = torch.from_numpy(ind.astype(np.int64))
        g = torch.Generator().manual_seed(ks)
        feat = torch.randn(len(ind), 4, generator=g, dtype=torch.float64)
        w = torch.randn(6, ks, ks, ks, 4, generator=g, dtype=torch.float64)
        bias = torch.randn(6, generator=g, dtype=torch.float64)
        nbr = oops.subm_rulebook(ind, ks)
        out = oops.gather_conv(feat, w, bias, nbr)
        dense = _dense(feat, li, (9, 9, 9), 2)
        ref = F.conv3d(dense, w.permute(0, 4, 1, 2, 3), bias, padding=ks // 2)
        ref_s = ref[li[:, 0], :, li[:, 1], li[:, 2], li[:, 3]]
        assert torch.allclose(out, ref_s, atol=1e-10)


def test_down_and_inverse_conv_match_dense():
    ind = _random_voxels(1, 500, 10)
    li = torch.from_numpy(ind.astype(np.int64))
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(len(ind), 3, generator=g, dtype=torch.float64)
    w = torch.randn(5, 2, 2, 2, 3, generator=g, dtype=torch.float64)
    out_ind, out_of_in, nbr_down, nbr_up = oops.down_rulebook(ind)
    out = oops.gather_conv(feat, w, None, nbr_down)
    dense = _dense(feat, li, (10, 10, 10), 2)
    ref = F.conv3d(dense, w.permute(0, 4, 1, 2, 3), None, stride=2)
    lo = torch.from_numpy(out_ind.astype(np.int64))
    assert torch.allclose(out, ref[lo[:, 0], :, lo[:, 1], lo[:, 2], lo[:, 3]], atol=1e-10)
    # coarse sites: ascending (batch, Morton code), unique, exactly the occupied parents
    def spread(v):
        return sum(((v >> i) & 1) << (3 * i) for i in range(8))
    key = (lo[:, 0] << 30) | (spread(lo[:, 1]) << 2) | (spread(lo[:, 2]) << 1) | spread(lo[:, 3])
    assert bool((key[1:] > key[:-1]).all())
    occ = F.max_pool3d(_dense(torch.ones(len(ind), 1, dtype=torch.float64), li, (10, 10, 10), 2), 2)
    assert int(occ.sum()) == len(out_ind)
    # inverse conv == conv_transpose3d sampled at the fine sites
    wi = torch.randn(3, 2, 2, 2, 5, generator=g, dtype=torch.float64)  # [C_out_fine, k, C_in_coarse]
    up = oops.gather_conv(out, wi, None, nbr_up)
    dense_c = _dense(out, lo, (5, 5, 5), 2)
    ref_up = F.conv_transpose3d(dense_c, wi.permute(4, 0, 1, 2, 3), stride=2)
    assert torch.allclose(up, ref_up[li[:, 0], :, li[:, 1], li[:, 2], li[:, 3]], atol=1e-10)


def test_subm_rulebook_duplicates_lowest_index_wins():
    ind = _random_voxels(3, 150, 6, batch=1)
    n = len(ind)
    ind2 = np.concatenate([ind, ind[:50]], axis=0)
    nbr = oops.subm_rulebook(ind2, 3)
    assert nbr.max() < n  # duplicates (rows >= n) are never referenced
    assert np.array_equal(nbr[13, :n], np.arange(n))    # centre tap of the first copy = itself
    assert np.array_equal(nbr[13, n:], np.arange(50))   # centre tap of a duplicate = the lowest row


def test_segment_csr_brute_force():
    g = torch.Generator().manual_seed(0)
    counts = torch.randint(0, 6, (50,), generator=g)
    indptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    src = torch.randn(int(indptr[-1]), 5, generator=g)
    src[3] = src[4]  # a tie inside some segment
    for red in ("sum", "mean", "max", "min"):
        out = oops.segment_csr(src, indptr, red)
        for s in range(50):
            seg = src[indptr[s]:indptr[s + 1]]
            if len(seg) == 0:
                ref = torch.zeros(5)
            else:
                ref = dict(sum=seg.sum(0), mean=seg.mean(0), max=seg.max(0).values, min=seg.min(0).values)[red]
            assert torch.allclose(out[s], ref, atol=1e-6), (red, s)
    # max backward goes to the FIRST arg-max
    x = torch.tensor([[1.0], [3.0], [3.0], [2.0]], requires_grad=True)
    oops.segment_csr(x, torch.tensor([0, 4]), "max").sum().backward()
    assert x.grad.flatten().tolist() == [0.0, 1.0, 0.0, 0.0]


def test_attention_varlen_matches_sdpa():
    g = torch.Generator().manual_seed(1)
    lens = [5, 32, 1, 17]
    cu = [0] + list(np.cumsum(lens))
    qkv = torch.randn(sum(lens), 3, 4, 16, generator=g)
    out, lse = oops.attention_varlen(qkv, cu, 0.25, return_lse=True)
    for a, b in zip(cu[:-1], cu[1:]):
        q, k, v = (qkv[a:b, j].transpose(0, 1)[None] for j in range(3))
        ref = F.scaled_dot_product_attention(q, k, v, scale=0.25)[0].transpose(0, 1)
        assert torch.allclose(out[a:b], ref, atol=1e-5)
        s = (q[0] * 0.25) @ k[0].transpose(1, 2)
        assert torch.allclose(lse[:, a:b], torch.logsumexp(s, -1), atol=1e-5)


def test_pointops2_oracle_against_the_kernels_index_arithmetic():
    """oracle/pointops2.py (the torch formulations of the reference's operator tests) against a literal loop over the flat
    index arithmetic of the CUDA kernels (table[r * C * 3 + h * d * 3 + i * 3 + a], q[n * C + h * d + i]:
    libs/pointops2/src/rpe_v2/relative_pos_encoding_cuda_kernel_v2.cu:248-282,398-437) on a small case."""
    from oracle import pointops2 as orc

    g = torch.Generator().manual_seed(0)
    N, M, h, d, L = 7, 23, 2, 4, 5
    q, k, v = (torch.rand(N, h, d, generator=g, dtype=torch.float64) for _ in range(3))
    tq, tk, tv = (torch.rand(L, h, d, 3, generator=g, dtype=torch.float64) for _ in range(3))
    i0 = torch.sort(torch.randint(0, N, (M,), generator=g)).values
    i1 = torch.randint(0, N, (M,), generator=g)
    rel = torch.randint(0, L, (M, 3), generator=g)
    attn = torch.rand(M, h, generator=g, dtype=torch.float64)
    C = h * d
    qf, kf, vf, tqf, tkf, tvf = (t.reshape(-1) for t in (q, k, v, tq, tk, tv))
    out1 = torch.zeros(M, h, dtype=torch.float64)
    out3 = torch.zeros(M, h, dtype=torch.float64)
    agg = torch.zeros(N, h, d, dtype=torch.float64)
    for m in range(M):
        r1, r2, r3 = (int(x) for x in rel[m])
        for hh in range(h):
            for i in range(d):
                t_q = tqf[r1 * C * 3 + hh * d * 3 + i * 3] + tqf[r2 * C * 3 + hh * d * 3 + i * 3 + 1] + tqf[r3 * C * 3 + hh * d * 3 + i * 3 + 2]
                t_k = tkf[r1 * C * 3 + hh * d * 3 + i * 3] + tkf[r2 * C * 3 + hh * d * 3 + i * 3 + 1] + tkf[r3 * C * 3 + hh * d * 3 + i * 3 + 2]
                t_v = tvf[r1 * C * 3 + hh * d * 3 + i * 3] + tvf[r2 * C * 3 + hh * d * 3 + i * 3 + 1] + tvf[r3 * C * 3 + hh * d * 3 + i * 3 + 2]
                qv, kv = qf[int(i0[m]) * C + hh * d + i], kf[int(i1[m]) * C + hh * d + i]
                out1[m, hh] += qv * kv
                out3[m, hh] += qv * t_q + kv * t_k
                agg[int(i0[m]), hh, i] += attn[m, hh] * (vf[int(i1[m]) * C + hh * d + i] + t_v)
    assert torch.allclose(orc.attention_step1(q, k, i0, i1), out1, atol=1e-12)
    assert torch.allclose(orc.dot_prod_with_idx_v3(q, i0, k, i1, tq, tk, rel), out3, atol=1e-12)
    assert torch.allclose(orc.attention_step2(attn, v, i0, i1, N, tv, rel), agg, atol=1e-12)
    off = orc.offsets_of(i0, N)
    assert off.numel() == N + 1 and int(off[-1]) == M and all(int(off[n]) <= int(off[n + 1]) for n in range(N))

