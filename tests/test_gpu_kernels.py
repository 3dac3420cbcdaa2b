"""-m gpu parity tests proper: every libptcore.so kernel, called through the C-ABI (via
pointcept_amd.ops), against the CPU oracle on the same seeded inputs.

Bars: integer / index outputs bit-exact; fp32 features rtol 2e-5 (summation-order differences only);
bf16 / f16 features compared with an fp32 oracle fed the SAME rounded inputs, tolerance one output
rounding step (2^-8 relative for bf16, 2^-10 for f16) plus fp32 accumulation slack.
"""
import numpy as np
import pytest
import torch

from oracle import maps as omaps
from oracle import ops as oops
from oracle import sfc as osfc

pytestmark = pytest.mark.gpu

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


def _t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _close(name, got, ref, rtol, atol):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite values in kernel output"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; worst at flat {i}: "
            f"got {got.flatten()[i].item():.6g} ref {ref.flatten()[i].item():.6g} "
            f"(max abs err {err.max().item():.3g}, ref absmax {ref.abs().max().item():.3g})")


# ------------------------------------------------------------------------------------------------
# A. serialization
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("depth", [1, 3, 7, 8, 9, 12, 16])
@pytest.mark.parametrize("i64", [True, False])
def test_serialize_encode_bit_exact(cuda, depth, i64):
    from pointcept_amd import ops

    rng = np.random.default_rng(depth)
    n = 20000 + depth
    gc = rng.integers(0, 1 << depth, size=(n, 3), dtype=np.int64)
    b = rng.integers(0, 16, size=n, dtype=np.int64)
    ref = osfc.encode_c(gc, b, depth, ORDERS)
    got = ops.serialize_encode(_t(gc if i64 else gc.astype(np.int32), cuda), _t(b, cuda), depth, ORDERS)
    assert np.array_equal(got.cpu().numpy(), ref)
    got_nb = ops.serialize_encode(_t(gc, cuda), None, depth, ("hilbert", "z"))
    assert np.array_equal(got_nb.cpu().numpy(), osfc.encode_c(gc, None, depth, ("hilbert", "z")))


def test_serialize_and_sort_full_size(cuda):
    """BASELINE config 3 size: 8 x 102400 points, 4 orders: codes bit-exact, orders bit-exact (unique keys)."""
    from pointcept_amd import ops, synthetic

    batch = synthetic.indoor_batch(8, 102400)
    gc, off = batch["grid_coord"], batch["offset"]
    b = omaps.offset2batch(off)
    depth, code, order, inverse = osfc.serialization(gc, b, ORDERS)
    got = ops.serialize_encode(_t(gc, cuda), _t(b, cuda), depth, ORDERS)
    assert np.array_equal(got.cpu().numpy(), code)
    bits = 3 * depth + int(len(off)).bit_length()
    go, gi = ops.sort_keys(got, 0, bits)
    assert np.array_equal(go.cpu().numpy(), order)
    assert np.array_equal(gi.cpu().numpy(), inverse)
    # size-independent properties
    srt = torch.gather(got, 1, go)
    assert bool((srt[:, 1:] >= srt[:, :-1]).all())
    ar = torch.arange(got.shape[1], device=cuda).expand_as(go)
    assert bool((torch.gather(gi, 1, go) == ar).all())


# ------------------------------------------------------------------------------------------------
# B. sort / scan
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 4095, 4096, 4097, 50000])
def test_sort_keys_stable(cuda, n):
    from pointcept_amd import ops

    rng = np.random.default_rng(n)
    for k, bits, hi in [(1, 64, 1 << 62), (4, 27, 1 << 27), (3, 10, 1 << 10), (2, 39, 1 << 39)]:
        keys = rng.integers(0, hi, size=(k, n), dtype=np.int64)
        if n > 10:
            keys[:, n // 2:] = keys[:, : n - n // 2]  # force duplicates: stability must hold
        order, inv = ops.sort_keys(_t(keys, cuda), 0, bits if bits < 64 else 63)
        ref = np.argsort(keys, axis=1, kind="stable")
        assert np.array_equal(order.cpu().numpy(), ref), f"n={n} k={k} bits={bits}"
        ar = np.arange(n)
        for r in range(k):
            assert np.array_equal(inv.cpu().numpy()[r][ref[r]], ar)


def test_sort_keys_bit_window(cuda):
    from pointcept_amd import ops

    rng = np.random.default_rng(7)
    keys = rng.integers(0, 1 << 40, size=(2, 30000), dtype=np.int64)
    order, _ = ops.sort_keys(_t(keys, cuda), 8, 29)
    ref = np.argsort((keys >> 8) & ((1 << 21) - 1), axis=1, kind="stable")
    assert np.array_equal(order.cpu().numpy(), ref)
    order0, inv0 = ops.sort_keys(_t(keys, cuda), 5, 5)  # empty window -> identity
    assert np.array_equal(order0.cpu().numpy(), np.tile(np.arange(30000), (2, 1)))


@pytest.mark.parametrize("n", [1, 2, 5000, 8192, 8193])
def test_sort_keys_at_the_packing_boundary(cuda, n):
    """The radix passes carry the row index in the low ceil(log2 n) bits of the key word when key bits + index bits fit 64 bits and as a
    separate array otherwise (scan_sort.hip, round 4): the same stable order on both sides of that boundary, with key bits above
    end_bit present (they are ignored) and a window that does not start at bit 0."""
    from pointcept_amd import ops

    rng = np.random.default_rng(n + 3)
    idx_bits = max(1, int(n - 1).bit_length())
    for end_bit in (64 - idx_bits - 1, 64 - idx_bits, 64 - idx_bits + 1):            # packs, packs (exactly 64 bits), does not pack
        for begin_bit in (0, 5):
            keys = rng.integers(0, 1 << 62, size=(2, n), dtype=np.int64) | (rng.integers(0, 2, size=(2, n), dtype=np.int64) << 62)
            if n > 10:
                keys[:, n // 2:] = keys[:, : n - n // 2]                              # duplicates: stability must hold
            order, inv = ops.sort_keys(_t(keys, cuda), begin_bit, end_bit)
            window = (keys.astype(np.uint64) >> np.uint64(begin_bit)) & np.uint64((1 << (end_bit - begin_bit)) - 1)
            ref = np.argsort(window, axis=1, kind="stable")
            assert np.array_equal(order.cpu().numpy(), ref), (n, begin_bit, end_bit)
            for r in range(2):
                assert np.array_equal(inv.cpu().numpy()[r][ref[r]], np.arange(n))


@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 16383, 16384, 16385, 300001])   # <= 16384: the one-workgroup single-launch form
def test_exclusive_scan(cuda, n):
    from pointcept_amd import ops

    rng = np.random.default_rng(n)
    x = rng.integers(0, 1000, size=n).astype(np.int32)
    got = ops.exclusive_scan_i32(_t(x, cuda)).cpu().numpy()
    ref = np.concatenate([[0], np.cumsum(x.astype(np.int64))[:-1]])
    assert np.array_equal(got, ref)


# ------------------------------------------------------------------------------------------------
# C / D. index maps
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("counts,K", [([10, 3, 7], 4), ([1024, 1025, 5000, 1], 1024), ([48, 49, 100, 7], 48),
                                      ([330, 1425, 2048], 1024), ([102400] * 8, 1024), ([5], 1024),
                                      ([2047, 2049, 1023, 1024, 1], 1024)])
def test_patch_pad_maps(cuda, counts, K):
    from pointcept_amd import ops

    off = np.cumsum(counts).astype(np.int64)
    pad, unpad, cu = omaps.pad_maps(off, K)
    dup = omaps.dup_map(pad, unpad)
    gp, gu, gc_, gd = ops.patch_pad_maps(_t(off, cuda), off.tolist(), K)
    assert np.array_equal(gp.cpu().numpy(), pad)
    assert np.array_equal(gu.cpu().numpy(), unpad)
    assert np.array_equal(gc_.cpu().numpy(), cu)
    assert np.array_equal(gd.cpu().numpy(), dup)


@pytest.mark.parametrize("n_pts", [2000, 60000])
def test_pool_maps(cuda, n_pts):
    from pointcept_amd import ops, synthetic

    batch = synthetic.indoor_batch(2, n_pts)
    gc, off = batch["grid_coord"], batch["offset"]
    b = omaps.offset2batch(off)
    depth, code, order, inverse = osfc.serialization(gc, b, ORDERS)
    ref = omaps.pooling_maps(code, 2, depth)
    dcode, dorder = _t(code, cuda), _t(order, cuda)
    cluster, idx_ptr, head = ops.pool_maps(dcode[0], dorder[0], 3)
    assert np.array_equal(cluster.cpu().numpy(), ref["cluster"])
    assert np.array_equal(idx_ptr.cpu().numpy(), ref["idx_ptr"])
    child = ops.pool_child_codes(dcode, head, 3)
    assert np.array_equal(child.cpu().numpy(), ref["code"])  # head choice is immaterial for the child codes
    # members listed under idx_ptr (via order0) are exactly the cluster's members
    o0 = order[0]
    assert np.array_equal(ref["cluster"][o0], np.repeat(np.arange(len(ref["counts"])), ref["counts"]))
    assert np.array_equal(ref["cluster"][head.cpu().numpy()], np.arange(len(ref["counts"])))
    bits = 3 * (depth - 1) + int(len(off)).bit_length()
    corder, cinv = ops.sort_keys(child, 0, bits)
    assert np.array_equal(corder.cpu().numpy(), ref["order"])
    assert np.array_equal(cinv.cpu().numpy(), ref["inverse"])


# ------------------------------------------------------------------------------------------------
# E. rows
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c", [3, 16, 96, 512])
def test_gather_rows(cuda, dtype, c):
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(c)
    src = torch.randn(5000, c, generator=g).to(dtype)
    idx = torch.randint(0, 5000, (7777,), generator=g)
    idx[::17] = -1
    idx2 = torch.randint(0, 5000, (7777,), generator=g)
    idx2[::3] = -1
    got = ops.gather_rows(src.to(cuda), idx.to(cuda))
    ref = torch.where((idx >= 0)[:, None], src[idx.clamp(min=0)], torch.zeros((), dtype=dtype))
    assert torch.equal(got.cpu(), ref)  # pure data movement: bit-exact
    got2 = ops.gather_rows(src.to(cuda), idx.to(cuda), idx2.to(cuda))
    ref2 = ref.float() + torch.where((idx2 >= 0)[:, None], src[idx2.clamp(min=0)].float(), torch.zeros(()))
    assert torch.equal(got2.cpu(), ref2.to(dtype))  # one fp32 add, one rounding


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("reduce", ["max", "mean", "sum", "min"])
@pytest.mark.parametrize("c", [3, 64])
def test_segment_csr(cuda, dtype, reduce, c):
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(11)
    counts = torch.randint(1, 9, (3000,), generator=g)
    counts[5] = 0  # an empty segment
    indptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    n = int(indptr[-1])
    perm = torch.randperm(n, generator=g)
    src = torch.randn(n, c, generator=g).to(dtype)
    out, arg = ops.segment_csr_fwd(src.to(cuda), perm.to(cuda), indptr.to(cuda), reduce)
    ref = oops.segment_csr(src[perm].float(), indptr, reduce)
    if reduce in ("max", "min"):
        assert torch.equal(out.cpu().float(), ref)  # selection: exact
    else:
        _close(f"segment_{reduce}", out, ref, 1e-2 if dtype != torch.float32 else 1e-6, 1e-2 if dtype != torch.float32 else 1e-6)
    # backward against autograd through the oracle
    gout = torch.randn(indptr.numel() - 1, c, generator=g).to(dtype)
    src_ref = src[perm].float().requires_grad_(True)
    oops.segment_csr(src_ref, indptr, reduce).backward(gout.float())
    gref = torch.zeros(n, c)
    gref[perm] = src_ref.grad
    gsrc = ops.segment_csr_bwd(gout.to(cuda), perm.to(cuda), indptr.to(cuda), arg, n, reduce)
    _close(f"segment_{reduce}_bwd", gsrc, gref.to(dtype).float(), 1e-2 if dtype != torch.float32 else 1e-6, 1e-6)


# ------------------------------------------------------------------------------------------------
# F. rulebooks
# ------------------------------------------------------------------------------------------------
def _scene_indices(n_pts, batch=2, dup=False):
    from pointcept_amd import synthetic

    b = synthetic.indoor_batch(batch, n_pts)
    ind = np.concatenate([omaps.offset2batch(b["offset"])[:, None], b["grid_coord"]], axis=1).astype(np.int32)
    if dup:  # Mix3D-style duplicate voxels
        ind = np.concatenate([ind, ind[::7]], axis=0)
    return ind


@pytest.mark.parametrize("ksize", [3, 5])
@pytest.mark.parametrize("dup", [False, True])
def test_rulebook_subm(cuda, ksize, dup):
    from pointcept_amd import ops

    ind = _scene_indices(6000, dup=dup)
    ref = oops.subm_rulebook(ind, ksize)
    got = ops.rulebook_subm(_t(ind, cuda), ksize)
    assert np.array_equal(got.cpu().numpy(), ref)
    # symmetry property used by dgrad: nbr[k][i] = j  <=>  nbr[kv-1-k][j] = i  (unique voxels only)
    if not dup:
        kv = ksize ** 3
        g = got.cpu().numpy()
        k, i = np.nonzero(g >= 0)
        assert np.array_equal(g[kv - 1 - k, g[k, i]], i)


def test_rulebook_down(cuda):
    from pointcept_amd import ops

    ind = _scene_indices(9000)
    oi, ooi, nd, nu = oops.down_rulebook(ind)
    cb = int(ind[:, 1:].max() >> 1).bit_length()
    goi, gnd, gnu = ops.rulebook_down(_t(ind, cuda), max(cb, 1), 2)
    assert np.array_equal(goi.cpu().numpy(), oi)
    assert np.array_equal(gnd.cpu().numpy(), nd)
    assert np.array_equal(gnu.cpu().numpy(), nu)


# ------------------------------------------------------------------------------------------------
# G. sparse conv compute
# ------------------------------------------------------------------------------------------------
def _tols(dtype):
    if dtype == torch.float32:
        return 2e-5, 2e-5
    if dtype == torch.bfloat16:
        return 1.0 / 128, 2e-3
    return 1.0 / 512, 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,ksize", [(8, 32, 5), (32, 32, 3), (64, 64, 3), (96, 96, 3), (128, 48, 3),
                                            (192, 128, 3), (256, 256, 3), (16, 16, 3), (128, 96, 3), (96, 64, 3), (32, 192, 3),
                                            (8, 48, 5), (8, 64, 5), (16, 48, 3), (8, 48, 3),    # LitePT's 6 -> 36 stem family
                                            (48, 48, 3), (80, 80, 3), (144, 144, 3), (48, 80, 3), (80, 48, 3)])   # 36 / 72 / 144 channels padded to 16
def test_spconv_fwd_and_wgrad(cuda, dtype, cin, cout, ksize):
    from pointcept_amd import ops

    ind = _scene_indices(1500 if cin * cout > 20000 else 4000)
    n = ind.shape[0]
    nbr = oops.subm_rulebook(ind, ksize)
    kv = ksize ** 3
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    feat = (torch.randn(n, cin, generator=g) * 0.5).to(dtype)
    w = (torch.randn(cout, kv, cin, generator=g) / (kv * cin) ** 0.5 * 2).to(dtype)
    bias = torch.randn(cout, generator=g)
    rtol, atol = _tols(dtype)
    ref = oops.gather_conv(feat.float(), w.float(), bias, nbr)
    got = ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), _t(nbr, cuda))
    _close("spconv_fwd", got, ref, rtol, atol)
    got_nb = ops.spconv_fwd(feat.to(cuda), w.to(cuda), None, _t(nbr, cuda))
    _close("spconv_fwd_nobias", got_nb, ref - bias, rtol, atol)
    # wgrad (fp32 output, fp32 accumulation over rows)
    dout = (torch.randn(n, cout, generator=g) * 0.5).to(dtype)
    fr = feat.float()
    wr = w.float().requires_grad_(True)
    oops.gather_conv(fr, wr, None, nbr).backward(dout.float())
    dw = ops.spconv_wgrad(feat.to(cuda), dout.to(cuda), _t(nbr, cuda))
    _close("spconv_wgrad", dw, wr.grad, 1e-4, 1e-3 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("cin,cout,ksize,n_pts", [
    (32, 64, 3, 1300), (64, 64, 3, 1300), (64, 128, 3, 1300), (128, 128, 3, 1300), (256, 64, 3, 1300), (512, 128, 3, 1300),
    (64, 128, 2, 1300), (128, 64, 5, 1300), (32, 32, 3, 1300), (96, 96, 3, 1300), (128, 96, 3, 1300), (160, 32, 3, 1300),
    (192, 64, 2, 1300), (64, 32, 3, 1300), (64, 96, 3, 1300), (32, 96, 3, 1300), (32, 64, 5, 1300), (128, 64, 2, 1300),
    (128, 128, 3, 17000), (96, 96, 3, 34000), (128, 96, 3, 34000), (160, 32, 3, 34000), (256, 64, 3, 34000),
    (256, 256, 3, 34000), (128, 128, 3, 72000)])      # the last two: 128-column workgroups (conv3 NTILES = 8, round 4)
def test_spconv_fwd_chunked_pipeline(cuda, cin, cout, ksize, n_pts):
    """conv3 (double-buffered W chunks, fragment-order LDS, gather ring: c_in >= 96) and conv5 (whole-row coalesced gathers
    through swizzled wave-private tile images: c_in = 32 / 64): every chunking case (4 / 2 / 1 table rows per 128-channel
    chunk, multi-chunk rows, partial last chunk), both workgroup shapes of conv3 (128-row workgroups at the small scenes,
    256-row ones where 256-row blocks fill the chip: the last five cases), ragged row count, bf16 and f16, against the oracle in fp32."""
    from pointcept_amd import ops

    ind = _scene_indices(n_pts)
    n = ind.shape[0]
    if ksize == 2:
        _, _, nbr, _ = oops.down_rulebook(ind)
        kv = 8
    else:
        nbr = oops.subm_rulebook(ind, ksize)
        kv = ksize ** 3
    n_out = nbr.shape[1]
    g = torch.Generator().manual_seed(cin * 7 + cout + ksize)
    for dtype in (torch.bfloat16, torch.float16):
        feat = (torch.randn(n, cin, generator=g) * 0.5).to(dtype)
        w = (torch.randn(cout, kv, cin, generator=g) / (kv * cin) ** 0.5 * 2).to(dtype)
        bias = torch.randn(cout, generator=g)
        rtol, atol = _tols(dtype)
        ref = oops.gather_conv(feat.float(), w.float(), bias, nbr)
        got = ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), _t(nbr, cuda))
        assert got.shape == (n_out, cout)
        _close(f"conv3_{dtype}", got, ref, rtol, atol)
        again = ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), _t(nbr, cuda))
        assert torch.equal(got, again), "conv3 must be bit-reproducible"


def _curve_sorted_indices(n_pts, batch=2):
    """scene indices with rows in Hilbert order (what PTC_SORT_POINTS / the SpUNet entry sort give the kernels)"""
    from pointcept_amd import synthetic

    b = synthetic.indoor_batch(batch, n_pts)
    bt = omaps.offset2batch(b["offset"])
    gc = b["grid_coord"]
    depth = int(gc.max() + 1).bit_length()
    code = osfc.encode_c(gc, bt, depth, ("hilbert",))[0]
    o = np.argsort(code, kind="stable")
    return np.concatenate([bt[o, None], gc[o]], axis=1).astype(np.int32)


@pytest.mark.parametrize("ordered", [True, False])
def test_rulebook_blocks(cuda, ordered):
    """block-local rulebook (csrc/blocks.hip): halo lists ascending + distinct + exactly the rows the block names (padded with the last
    one to a multiple of 16), the uint16 table maps every entry to its position at [block][tap][row in tile][tile] and carries the
    per-tile tap-occupancy masks in its padding row; rows in no spatial
    order overflow and are flagged (count -1), not truncated."""
    from pointcept_amd import ops

    ind = _curve_sorted_indices(9000) if ordered else _scene_indices(9000)
    nbr = oops.subm_rulebook(ind, 3)
    n = nbr.shape[1]
    bt = ops.BlockTables(_t(nbr, cuda))
    bm, hcap = bt.bm, bt.hcap
    tab, hid, hcnt = bt.tab.cpu().numpy().astype(np.uint16), bt.hid.cpu().numpy(), bt.hcnt.cpu().numpy()
    nblk = (n + bm - 1) // bm
    assert hcnt.shape == (nblk,) and tab.shape == (2, nblk, 28, 32, 4)
    n_ovf = 0
    for b in range(nblk):
        e = nbr[:, b * bm:(b + 1) * bm]
        want = np.unique(e[e >= 0])
        if len(want) > hcap:
            assert hcnt[b] == -1
            n_ovf += 1
            continue
        c = len(want)
        assert hcnt[b] == c
        assert np.array_equal(hid[b, :c], want)                      # ascending, distinct, complete
        cpad = min((c + 15) // 16 * 16, hcap)
        assert (hid[b, c:cpad] == want[-1]).all()                    # padding: whole DMA instructions fetch valid rows
        rows = e.shape[1]
        swz = (lambda sl: (((sl >> 1) & 1) << 2) | ((sl >> 2) & 3), lambda sl: (sl >> 2) & 3)       # PTC_SWZ64 (csrc/ptc_common.h) / 32-channel rows
        for v, rowb in enumerate((128, 64)):      # 64-channel rows, 32-channel rows
            none = hcap * rowb
            full = tab[v, b, :27].transpose(0, 2, 1).reshape(27, 128).astype(np.int64)  # [k][32 t + r]
            le = full[:, :rows]
            assert np.array_equal(le == none, e < 0)
            slot = le // rowb
            assert np.array_equal((le % rowb)[e >= 0], (swz[v](slot) * 16)[e >= 0])     # piece 0 at its swizzled position
            assert np.array_equal(hid[b][np.where(le != none, slot, 0)][e >= 0], e[e >= 0])
            assert (full[:, rows:] == none).all()
            # table row 27: the tap masks (bit k of word t: tile t has a neighbour at tap k; word 4: their OR; words 5..12: bit k of
            # word 5 + s set when one of the rows {32 t + 4 s + q} -- an MFMA step of wgrad7 -- has one), then "none" padding
            pad = tab[v, b, 27].reshape(-1)
            words = pad[:26].astype(np.uint32)
            masks = words[0::2] | (words[1::2] << 16)
            occ = np.zeros((27, 128), bool)
            occ[:, :rows] = e >= 0
            want_masks = [sum(1 << k for k in range(27) if occ[k, 32 * t:32 * t + 32].any()) for t in range(4)]
            assert masks[:4].tolist() == want_masks and int(masks[4]) == (want_masks[0] | want_masks[1] | want_masks[2] | want_masks[3])
            step_rows = [[32 * t + 4 * s_ + q for t in range(4) for q in range(4)] for s_ in range(8)]
            assert masks[5:13].tolist() == [sum(1 << k for k in range(27) if occ[k, step_rows[s_]].any()) for s_ in range(8)]
            assert (pad[26:] == none).all()
    assert int(bt.n_overflow.item()) == n_ovf
    if ordered:
        assert n_ovf == 0, "curve-ordered rows must fit their halo budget"
    else:
        assert n_ovf > 0, "this case is meant to exercise the overflow flag"


@pytest.mark.parametrize("c", [32, 64])
@pytest.mark.parametrize("ordered", [True, False])
def test_spconv_fwd_block_staged(cuda, c, ordered, n_rows=35000):
    """conv7 (weights in registers, the input rows of a 128-row block staged once in LDS by the DMA path, csrc/conv7.h): within the
    16-bit bar of the fp32 oracle and within fp32 summation-order noise of the global-gather kernel on the same table; the un-ordered
    case runs conv7 for the blocks that fit and the global-gather kernel for the overflowing ones.  Ragged row count (last block
    partial), several blocks per persistent workgroup (70000 rows = 547 blocks on <= 256 workgroups), bf16 and f16, with / without bias."""
    from pointcept_amd import ops

    ind = _curve_sorted_indices(n_rows)      # (the host-emulation tier runs this body with 4500 rows)
    if not ordered:   # second half of the rows in random order: the first blocks fit their halo budget, the others overflow
        rng = np.random.default_rng(c)
        h = ind.shape[0] // 2
        ind = np.concatenate([ind[:h], ind[h:][rng.permutation(ind.shape[0] - h)]])
    nbr = oops.subm_rulebook(ind, 3)
    n = nbr.shape[1]
    assert ops.block_plan(c, c, 27, torch.bfloat16, n) is not None
    nbr_d = _t(nbr, cuda)
    bt = ops.BlockTables(nbr_d)
    assert (int(bt.n_overflow.item()) == 0) == ordered
    g = torch.Generator().manual_seed(c * 131)
    for dtype in (torch.bfloat16, torch.float16):
        feat = (torch.randn(n, c, generator=g) * 0.5).to(dtype)
        w = (torch.randn(c, 27, c, generator=g) / (27 * c) ** 0.5 * 2).to(dtype)
        bias = torch.randn(c, generator=g)
        base = ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), nbr_d)
        got = ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), nbr_d, bt)
        assert torch.isfinite(got.float()).all()
        ref = oops.gather_conv(feat.float(), w.float(), bias, nbr)
        rtol, atol = _tols(dtype)
        _close(f"conv7_{dtype}", got, ref, rtol, atol)
        # against the global-gather kernel: one rounding step of the output dtype at most (different fp32 summation order)
        ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        assert float((got.float() - base.float()).abs().max()) <= 2 * ulp * float(ref.abs().max())
        assert torch.equal(got, ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), nbr_d, bt)), "conv7 must be bit-reproducible"
        nb = ops.spconv_fwd(feat.to(cuda), w.to(cuda), None, nbr_d, bt)
        _close(f"conv7_nobias_{dtype}", nb, ref - bias, rtol, atol)


@pytest.mark.parametrize("c", [(96, 96), (128, 96), (128, 128), (256, 256), (224, 96), (192, 64), (512, 512), (96, 32)])
@pytest.mark.parametrize("ordered", [True, False])
def test_spconv_fwd_block_staged_wide(cuda, c, ordered, monkeypatch, n_rows=9000):
    """round 6, conv8 (csrc/conv8.h): the block-staged convolution for rows of 96 channels and more -- halo rows of a 128-row block staged
    once per 64-channel chunk in LDS, weights fetched in MFMA fragment order straight into registers -- at SpUNet's decoder widths (96 /
    128 / 224 / 192), PT-v3's deep stages (128 / 256 / 512) and mixed in / out widths (the 128-row form at c_out = 96, the 64-row form at
    c_out = 64, the 32-row form elsewhere at this row count; the full-size model tests run the 128-row form at c_out = 128): within the
    16-bit bar of the fp32 oracle and within fp32 summation-order noise of the global-gather kernel (conv3) on the same table.  The
    un-ordered case mixes staged blocks with blocks whose halo does not fit (blocks.hip's overflow mark, or more rows than the LDS
    image holds): those are served by the conv3 follow-up launch, which skips the staged blocks.  Ragged row count, bf16 and f16, with /
    without bias, bit-reproducible."""
    from pointcept_amd import ops

    monkeypatch.setenv("PTC_CONV8", "1")       # (the default since profiles/r06_y_conv8_no_copies.txt)
    c_in, c_out = c
    if c_in >= 512:
        n_rows = min(n_rows, 3000)
    ind = _curve_sorted_indices(n_rows)
    if not ordered:
        rng = np.random.default_rng(c_in + c_out)
        h = ind.shape[0] // 2
        ind = np.concatenate([ind[:h], ind[h:][rng.permutation(ind.shape[0] - h)]])
    nbr = oops.subm_rulebook(ind, 3)
    n = nbr.shape[1]
    assert ops.block_plan(c_in, c_out, 27, torch.bfloat16, n) is not None
    nbr_d = _t(nbr, cuda)
    bt = ops.BlockTables(nbr_d)
    assert (int(bt.n_overflow.item()) == 0) == ordered
    g = torch.Generator().manual_seed(c_in * 131 + c_out)
    for dtype in (torch.bfloat16, torch.float16):
        feat = (torch.randn(n, c_in, generator=g) * 0.5).to(dtype)
        w = (torch.randn(c_out, 27, c_in, generator=g) / (27 * c_in) ** 0.5 * 2).to(dtype)
        bias = torch.randn(c_out, generator=g)
        base = ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), nbr_d)
        got = ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), nbr_d, bt)
        assert torch.isfinite(got.float()).all()
        ref = oops.gather_conv(feat.float(), w.float(), bias, nbr)
        rtol, atol = _tols(dtype)
        _close(f"conv8_{dtype}", got, ref, rtol, atol)
        ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        assert float((got.float() - base.float()).abs().max()) <= 2 * ulp * float(ref.abs().max())
        assert torch.equal(got, ops.spconv_fwd(feat.to(cuda), w.to(cuda), bias.to(cuda), nbr_d, bt)), "conv8 must be bit-reproducible"
        if dtype == torch.bfloat16:
            nb = ops.spconv_fwd(feat.to(cuda), w.to(cuda), None, nbr_d, bt)
            _close(f"conv8_nobias_{dtype}", nb, ref - bias, rtol, atol)


@pytest.mark.parametrize("c", [32, 64, 128, (128, 96), 256, (192, 64), 96, (32, 64), (96, 32)])
@pytest.mark.parametrize("ordered", [True, False])
def test_spconv_wgrad_block_staged(cuda, c, ordered, n_rows=35000):
    """wgrad7 (the whole weight gradient held in MFMA accumulators of persistent workgroups, both operands built from a block's LDS
    images by transposing reads, empty (step, tap) pairs skipped, csrc/wgrad7.h): against the fp32 oracle (autograd of the gather
    convolution) and within fp32 summation-order noise of the global-gather kernel wgrad2 on the same table.  The un-ordered case has
    overflowing blocks: the device-side gate hands the call to wgrad2 (same result as the plain entry point, bit for bit).  Ragged row
    count, several blocks per persistent workgroup, bf16 and f16, bit-reproducible."""
    from pointcept_amd import ops

    # (c_in, c_out) pairs and c >= 128: the channel-sliced form (round 4) -- (c_out / 32)(c_in / 64) workgroups per block sequence, each
    # with one (32 x 64)-channel slice of dw; fewer rows there (the oracle's autograd convolution is the slow part)
    c_in, c_out = c if isinstance(c, tuple) else (c, c)
    if c_in > 64 or c_in != c_out:
        n_rows = min(n_rows, 9000)
    c = c_in
    ind = _curve_sorted_indices(n_rows)      # (the host-emulation tier runs this body with 4500 rows)
    if not ordered:
        rng = np.random.default_rng(c)
        h = ind.shape[0] // 2
        ind = np.concatenate([ind[:h], ind[h:][rng.permutation(ind.shape[0] - h)]])
    nbr = oops.subm_rulebook(ind, 3)
    n = nbr.shape[1]
    nbr_d = _t(nbr, cuda)
    bt = ops.BlockTables(nbr_d)
    assert (int(bt.n_overflow.item()) == 0) == ordered
    g = torch.Generator().manual_seed(c * 77)
    for dtype in (torch.bfloat16, torch.float16):
        feat = (torch.randn(n, c_in, generator=g) * 0.5).to(dtype)
        dout = (torch.randn(n, c_out, generator=g) * 0.5).to(dtype)
        wr = torch.zeros(c_out, 27, c_in, requires_grad=True)
        oops.gather_conv(feat.float(), wr, None, nbr).backward(dout.float())
        base = ops.spconv_wgrad(feat.to(cuda), dout.to(cuda), nbr_d)
        got = ops.spconv_wgrad(feat.to(cuda), dout.to(cuda), nbr_d, blk=bt)
        assert got.shape == (c_out, 27, c_in) and torch.isfinite(got).all()
        scale = float(wr.grad.abs().max())
        _close(f"wgrad7_{dtype}", got, wr.grad, 1e-4, 1e-3 * scale)
        if ordered:   # fp32 accumulation in another order: a few ulps of the partial sums
            assert float((got - base).abs().max()) <= 2e-5 * scale * max(1.0, (n / 4096) ** 0.5)
        else:         # the gate picked wgrad2: the same partials, the same reduction
            assert torch.equal(got, base)
        assert torch.equal(got, ops.spconv_wgrad(feat.to(cuda), dout.to(cuda), nbr_d, blk=bt)), "wgrad7 must be bit-reproducible"


def test_spconv_block_staged_full_size(cuda):
    """BASELINE size: 8 x 102400 voxels in curve order, 64 -> 64 and 32 -> 32: conv7 within summation-order noise of the global-gather
    kernel, no block overflows; and -- size-independent property -- linearity: conv(x1 + x2) == conv(x1) + conv(x2) to the output rounding."""
    from pointcept_amd import ops, synthetic

    b = synthetic.to_torch(synthetic.indoor_batch(8, 102400), cuda)
    off = b["offset"]
    bt_ = torch.repeat_interleave(torch.arange(off.numel(), device=cuda), torch.diff(off, prepend=off.new_zeros(1)))
    code = ops.serialize_encode(b["grid_coord"], bt_, 8, ("hilbert",))
    order, _ = ops.sort_keys(code, 0, 3 * 8 + 3)
    ind = torch.cat([bt_[:, None].int(), b["grid_coord"].int()], 1)[order[0]].contiguous()
    nbr = ops.rulebook_subm(ind, 3, ops.HashTable(ind))
    n = ind.shape[0]
    blk = ops.BlockTables(nbr)
    assert int(blk.n_overflow.item()) == 0 and int(blk.hcnt.min().item()) >= 1 and int(blk.hcnt.max().item()) <= blk.hcap
    for c in (64, 32):
        g = torch.Generator().manual_seed(c)
        x = torch.randn(n, c, generator=g).to(torch.bfloat16).to(cuda)
        w = (torch.randn(c, 27, c, generator=g) * 0.05).to(torch.bfloat16).to(cuda)
        bias = torch.randn(c, generator=g).to(cuda)
        got, base = ops.spconv_fwd(x, w, bias, nbr, blk), ops.spconv_fwd(x, w, bias, nbr)
        scale = float(base.float().abs().max())
        assert float((got.float() - base.float()).abs().max()) <= 2.0 ** -6 * scale
        assert float((got.float() - base.float()).abs().mean()) <= 2.0 ** -11 * scale
        x2 = torch.randn(n, c, generator=g).to(torch.bfloat16).to(cuda)
        xs = (x.float() + x2.float()).to(torch.bfloat16)
        lhs = ops.spconv_fwd(xs, w, None, nbr, blk).float()
        rhs = ops.spconv_fwd(x, w, None, nbr, blk).float() + ops.spconv_fwd(x2, w, None, nbr, blk).float()
        assert float((lhs - rhs).abs().max()) <= 2.0 ** -5 * float(rhs.abs().max())
        # the block-staged weight gradient (wgrad7) against the global-gather one (wgrad2) on the same operands: fp32 sums of ~7.6 M
        # products per tap in two different orders; and a size-independent property: for dout = 1 and x = 1 every entry of tap k
        # equals the number of (row, neighbour) pairs of tap k
        dw7, dw2 = ops.spconv_wgrad(x, x2, nbr, blk=blk), ops.spconv_wgrad(x, x2, nbr)
        assert float((dw7 - dw2).abs().max()) <= 1e-3 * float(dw2.abs().max())
        ones = torch.ones(n, c, dtype=torch.bfloat16, device=cuda)
        cnt = ops.spconv_wgrad(ones, ones, nbr, blk=blk)
        pairs = (nbr >= 0).sum(1).float()
        assert torch.equal(cnt, pairs[None, :, None].expand(c, 27, c))


@pytest.mark.parametrize("c,width", [(36, 48), (72, 80), (64, 64)])
def test_batch_norm_act_on_a_column_slice_of_a_padded_gemm_output(cuda, c, width):
    """functional.batch_norm_act on x = wide[:, :c] (what a Linear / conv with a channel count that is not a multiple of 16 hands to
    the norm that follows it: LitePT's 36 / 72 channels): forward, input gradient, affine gradients against ATen in fp32.  The backward
    used to read the saved NON-contiguous view as dense rows (wrong gradients on hardware only: the CPU stand-ins honour strides)."""
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(c)
    n = 3000
    wide = (torch.randn(n, width, generator=g) * 2 + 0.3).to(cuda)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(cuda), torch.randn(c, generator=g).to(cuda)
    dy = torch.randn(n, c, generator=g).to(cuda)
    for act in ("none", "gelu"):
        xr = wide[:, :c].clone().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        y = torch.nn.functional.batch_norm(xr, None, None, gr, br, True, 0.01, 1e-3)
        y = torch.nn.functional.gelu(y) if act == "gelu" else y
        y.backward(dy)
        we = wide.clone().requires_grad_(True)
        ge, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        ye = PF.batch_norm_act(we[:, :c], ge, be, None, None, True, 0.01, 1e-3, act)
        ye.backward(dy)
        _close(f"bn_{act}_fwd", ye, y, 1e-4, 1e-4)
        _close(f"bn_{act}_dx", we.grad[:, :c], xr.grad, 1e-3, 1e-4)
        assert float(we.grad[:, c:].abs().max()) == 0.0 if width > c else True
        _close(f"bn_{act}_dgamma", ge.grad, gr.grad, 1e-3, 1e-3)
        _close(f"bn_{act}_dbeta", be.grad, br.grad, 1e-3, 1e-3)


@pytest.mark.parametrize("cin,cout,ksize,bias", [(36, 36, 3, True), (72, 72, 3, True), (6, 36, 5, False), (36, 72, 3, False), (144, 144, 3, True)])
def test_sparse_conv_autograd_channels_not_multiple_of_16(cuda, cin, cout, ksize, bias):
    """functional.sparse_conv (the autograd wrapper: channel padding to 16 / 8, weight shadows, mirrored-weight input gradient, weight
    and bias gradients) at LitePT's / PT-v3m3's channel counts (36, 72, 144; the 6 -> 36 stem) under bf16 autocast against the
    oracle's autograd in fp32: output, d feat, d weight, d bias."""
    from pointcept_amd import functional as PF
    from pointcept_amd import ops

    ind = _scene_indices(1800)
    nbr = oops.subm_rulebook(ind, ksize)
    kv, n = nbr.shape
    g = torch.Generator().manual_seed(cin * 17 + cout)
    feat = (torch.randn(n, cin, generator=g) * 0.5)
    w = (torch.randn(cout, kv, cin, generator=g) / (kv * cin) ** 0.5 * 2)
    b = torch.randn(cout, generator=g) if bias else None
    dout = torch.randn(n, cout, generator=g)
    fr, wr = feat.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = None if b is None else b.clone().requires_grad_(True)
    ref = oops.gather_conv(fr, wr, br, nbr)
    ref.backward(dout)
    stem = cin <= 8                       # a stem's input is data: no input gradient (and no 8-channel input-gradient kernel)
    fe, we = feat.to(cuda).requires_grad_(not stem), w.to(cuda).requires_grad_(True)
    be = None if b is None else b.to(cuda).requires_grad_(True)
    nbr_d = _t(nbr, cuda)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = PF.sparse_conv(fe, we, be, nbr_d, nbr_d, True)
    out.float().backward(dout.to(cuda))
    _close("sparse_conv_out", out.float(), ref.detach(), 1.0 / 64, 1e-2 * float(ref.abs().max()))
    if not stem:
        _close("sparse_conv_dfeat", fe.grad, fr.grad, 1.0 / 32, 2e-2 * float(fr.grad.abs().max()))
    _close("sparse_conv_dw", we.grad, wr.grad, 1.0 / 32, 2e-2 * float(wr.grad.abs().max()))
    if bias:
        _close("sparse_conv_db", be.grad, br.grad, 1.0 / 64, 1e-2 * float(br.grad.abs().max()))


def test_spconv_dgrad_via_mirrored_table(cuda):
    """dgrad = the same kernel with W' = W.permute(ci,k,co).flip(k) on the SAME table (Appendix A.6)."""
    from pointcept_amd import ops

    ind = _scene_indices(3000)
    n = ind.shape[0]
    nbr = oops.subm_rulebook(ind, 3)
    g = torch.Generator().manual_seed(5)
    cin, cout = 32, 64
    feat = torch.randn(n, cin, generator=g).requires_grad_(True)
    w = torch.randn(cout, 27, cin, generator=g) * 0.1
    dout = torch.randn(n, cout, generator=g)
    oops.gather_conv(feat, w, None, nbr).backward(dout)
    wt = w.permute(2, 1, 0).flip(1).contiguous()
    got = ops.spconv_fwd(dout.to(cuda), wt.to(cuda), None, _t(nbr, cuda))
    _close("spconv_dgrad", got, feat.grad, 2e-5, 2e-5)


def test_spconv_down_up_tables(cuda):
    from pointcept_amd import ops

    ind = _scene_indices(5000)
    oi, ooi, nd, nu = oops.down_rulebook(ind)
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(ind.shape[0], 32, generator=g)
    w = torch.randn(64, 8, 32, generator=g) * 0.2
    ref = oops.gather_conv(feat, w, None, nd)
    got = ops.spconv_fwd(feat.to(cuda), w.to(cuda), None, _t(nd, cuda))
    _close("down_conv", got, ref, 2e-5, 2e-5)
    wi = torch.randn(32, 8, 64, generator=g) * 0.2
    ref_up = oops.gather_conv(ref, wi, None, nu)
    got_up = ops.spconv_fwd(got, wi.to(cuda), None, _t(nu, cuda))
    _close("inverse_conv", got_up, ref_up, 1e-4, 1e-4)


# ------------------------------------------------------------------------------------------------
# G'. identity / gather-table GEMM (nn.Linear on point features) and LayerNorm
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,cin,cout", [(1, 32, 96), (5000, 32, 96), (3001, 64, 192), (777, 128, 512), (300, 512, 2048),
                                        (900, 2048, 512), (2500, 64, 20), (4100, 6, 32), (2821, 512, 1536), (12115, 1024, 256),
                                        (50360, 512, 128), (2821, 512, 512), (70000, 256, 1024),
                                        (16500, 512, 512),     # 65 x 4 wide (128-column) workgroups: conv3's NTILES = 8 instance, forward and dgrad
                                        # round 6: gemm3.h / wgrad3.h at their edges -- one row, less than a chunk, one row beyond a tile, a ragged last
                                        # 64-row chunk with several splits, the tall-tile form (>= 512 tiles of 128 rows)
                                        (1, 256, 128), (63, 128, 128), (129, 512, 256), (4133, 128, 384), (33000, 256, 2048),
                                        # round 5: contractions that are no multiple of 128 (general chunking of the identity-table kernel, operands
                                        # padded to 32): the MLP / qkv shapes of LitePT (36 .. 504), PT-v3m3 (54 .. 576) and PT-v3m2 (48 .. 384)
                                        (3000, 72, 288), (3000, 288, 72), (2000, 144, 576), (2000, 576, 144), (1500, 252, 1008), (1500, 1008, 252),
                                        (900, 504, 2016), (900, 2016, 504), (1200, 432, 1728), (1200, 1728, 432), (700, 576, 2304), (700, 2304, 576),
                                        (40000, 108, 432), (40000, 432, 108), (2500, 216, 648), (5000, 96, 288), (70000, 288, 96), (3000, 864, 216)])
def test_linear_identity_table(cuda, dtype, n, cin, cout):
    """PF.linear == F.linear (forward, input / weight / bias gradients) incl. channel padding; the contractions wider than 256 (the
    last cases = the qkv / fc2 / proj / fc1-dgrad shapes of PT-v3m1's 128 .. 512-channel stages) run on the identity-table instances of
    the chunked implicit-GEMM kernel, their small-row weight gradients on the split-K kernel: no library GEMM on the hot path."""
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(n + cin + cout)
    x = torch.randn(n, cin, generator=g).to(dtype)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dtype)
    b = torch.randn(cout, generator=g)
    dout = torch.randn(n, cout, generator=g).to(dtype)
    xr, wr, br = x.float().clone().requires_grad_(True), w.float().clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.linear(xr, wr, br)
    ref.backward(dout.float())
    xe, we, be = x.to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    got = PF.linear(xe, we, be)
    got.backward(dout.to(cuda))
    rtol, atol = _tols(dtype)
    _close("linear_fwd", got, ref, rtol, atol * 4)
    _close("linear_dx", xe.grad, xr.grad, rtol, atol * 4)
    _close("linear_dw", we.grad, wr.grad, 2 * rtol, 2e-3 * float(wr.grad.abs().max()))
    _close("linear_db", be.grad, br.grad, 1e-4, 2e-3 * float(br.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(32, 32), (128, 32), (64, 64), (256, 64), (128, 128)])
def test_linear_with_the_residual_joint_in_its_epilogue(cuda, dtype, cin, cout):
    """ptc_linear_joint_fwd (fwd2_joint.h): `proj` / `fc2` of a Block with the joint behind them -- z = a + droppath(x W^T + b), y = norm2(z) or
    the cast of z -- in ONE launch, against the two launches it replaces (the Linear on the same GEMM loop, then ptc_add_norm_fwd): the
    same arithmetic statement for statement, so z, y and the LayerNorm statistics must be bit-identical; with and without the inverse
    serialization table, DropPath row factors, LayerNorm; ragged row count."""
    from pointcept_amd import ops

    assert ops.linear_joint_supported(cin, cout, dtype) and not ops.linear_joint_supported(96, cout, dtype)
    g = torch.Generator().manual_seed(cin * 3 + cout)
    n, n_in = 3001, 3072
    x = torch.randn(n_in, cin, generator=g).to(dtype).to(cuda)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dtype).to(cuda)
    b = torch.randn(cout, generator=g).to(cuda)
    a = torch.randn(n, cout, generator=g).to(cuda)
    rs = (torch.rand(n, generator=g) > 0.3).float().to(cuda) / 0.7
    tab = torch.randint(0, n_in, (n,), generator=g).int().to(cuda)
    gam, bet = (torch.rand(cout, generator=g) + 0.5).to(cuda), torch.randn(cout, generator=g).to(cuda)
    for table, scale, norm in ((tab, rs, (gam, bet, 1e-5)), (None, None, None), (None, rs, None), (tab, None, (gam, bet, 1e-5))):
        xin = x if table is not None else x[:n].contiguous()
        z, y, st = ops.linear_joint_fwd(xin, w, b, table, a, scale, norm, dtype)
        u = ops.spconv_fwd(xin, w[:, None, :].contiguous(), b, None if table is None else table[None, :].contiguous())
        z2, y2, _, st2 = ops.add_norm_fwd(u, a, scale, None, norm, dtype)
        assert torch.equal(z, z2) and torch.equal(y, y2), (table is not None, scale is not None, norm is not None)
        assert (st is None) == (st2 is None) and (st is None or torch.equal(st, st2))
        ref = a.double() + (1.0 if scale is None else scale.double()[:, None]) * u.double()
        assert float((z.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    if cin == cout:
        # the positional-encoding joint (ptc_linear_norm_joint_fwd): the branch operand normalised first, the Linear's output written too;
        # the residual operand fp32 (the stream) or 16-bit (first Block of a stage)
        ga2, be2 = (torch.rand(cout, generator=g) + 0.5).to(cuda), torch.randn(cout, generator=g).to(cuda)
        xin = x[:n].contiguous()
        for res, norm in ((a, (gam, bet, 1e-5)), (a.to(dtype), (gam, bet, 1e-5)), (a, None)):
            u, z, y, sa, sb = ops.linear_norm_joint_fwd(xin, w, b, (ga2, be2, 1e-6), res, norm, dtype)
            u2 = ops.spconv_fwd(xin, w[:, None, :].contiguous(), b, None)
            z2, y2, sa2, sb2 = ops.add_norm_fwd(u2, res, None, (ga2, be2, 1e-6), norm, dtype)
            assert torch.equal(u, u2) and torch.equal(z, z2) and torch.equal(y, y2) and torch.equal(sa, sa2), (res.dtype, norm is not None)
            assert (sb is None) == (sb2 is None) and (sb is None or torch.equal(sb, sb2))


@pytest.mark.parametrize("dtype,c", [(torch.bfloat16, 64), (torch.float16, 32), (torch.float32, 20), (torch.bfloat16, 36)])
def test_unpooling_gather_with_addend(cuda, dtype, c):
    """round 6: SerializedUnpooling's `parent.feat + point.feat[inverse]` (ptv3m1:478) as ONE pass (ptc_gather_rows_add) -- bit-identical to
    the gather kernel followed by torch's add (one rounding of the fp32 sum either way); gradients: the addend's is the incoming gradient,
    the source's the segmented sum over the cluster CSR.  Widths that are no multiple of a 16-byte lane keep the two-pass form."""
    from pointcept_amd import functional as PF
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(c)
    n_par, n_child = 5003, 1201
    cluster = torch.randint(0, n_child, (n_par,), generator=g)
    perm = torch.argsort(cluster, stable=True)
    indptr = torch.zeros(n_child + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(torch.bincount(cluster, minlength=n_child), 0)
    src = torch.randn(n_child, c, generator=g).to(dtype)
    add = torch.randn(n_par, c, generator=g).to(dtype)
    dy = torch.randn(n_par, c, generator=g).to(dtype)
    s1, a1 = src.to(cuda).requires_grad_(True), add.to(cuda).requires_grad_(True)
    y = PF.gather_by_cluster_add(a1, s1, cluster.to(cuda), perm.to(cuda), indptr.to(cuda))
    y.backward(dy.to(cuda))
    s2, a2 = src.to(cuda).requires_grad_(True), add.to(cuda).requires_grad_(True)
    y2 = a2 + PF.gather_by_cluster(s2, cluster.to(cuda), perm.to(cuda), indptr.to(cuda))
    y2.backward(dy.to(cuda))
    assert torch.equal(y, y2) and torch.equal(a1.grad, a2.grad) and torch.equal(s1.grad, s2.grad)
    ref = add.float() + src.float()[cluster]
    assert torch.equal(y.detach().cpu(), ref.to(dtype))
    if torch.device(cuda).type == "cuda" or dtype != torch.float32 or c % 4 == 0:
        assert ops.gather_rows_add_supported(s1, a1) == (c % (4 if dtype == torch.float32 else 8) == 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c,n", [(64, 3001), (32, 3001), (64, 100), (32, 129)])
def test_mlp_one_kernel_per_direction(cuda, dtype, c, n, monkeypatch):
    """round 6 (csrc/mlp.hip): the MLP of a PT-v3m1 Block, fc1 -> GELU -> fc2 (+ the residual joint), as ONE kernel per direction with the
    hidden tensor kept on the CU (forward: fc1 accumulators -> GELU -> fc2 operand registers; backward: h recomputed, GELU(h) / dh through LDS
    into the weight gradients).  Against the split kernels it replaces:
      forward   z, y (with the joint) and m (without) BIT-IDENTICAL to ptc_linear_fwd_ex(epilogue 1) + ptc_linear_joint_fwd / ptc_spconv_fwd;
      backward  dx BIT-IDENTICAL to ptc_linear_fwd_ex(epilogue 2) + ptc_spconv_fwd(W1^T); weight / bias gradients equal to the split-K kernels'
                up to summation order (1e-5 of the largest entry), and reproducible run to run;
    and against the fp32 formulation of ptv3m1:225-248 at 16-bit operand accuracy.  Ragged row counts, several tiles per workgroup."""
    from pointcept_amd import ops

    on_gpu = torch.device(cuda).type == "cuda"
    if not on_gpu:
        monkeypatch.setenv("PTC_MLP_BWD_WGS", "3")          # emulation: three persistent workgroups = several tiles each at 3001 rows
        monkeypatch.setenv("PTC_MLP_FWD_WGS", "5")
    assert ops.mlp_supported(c, dtype) and not ops.mlp_supported(128, dtype) and not ops.mlp_supported(c, torch.float32)
    g = torch.Generator().manual_seed(c + n)
    hid = 4 * c
    x = torch.randn(n, c, generator=g).to(dtype).to(cuda)
    w1 = (torch.randn(hid, c, generator=g) / c ** 0.5).to(dtype).to(cuda)
    b1 = (torch.randn(hid, generator=g) * 0.5).to(cuda)
    w2 = (torch.randn(c, hid, generator=g) / hid ** 0.5).to(dtype).to(cuda)
    b2 = torch.randn(c, generator=g).to(cuda)
    a = torch.randn(n, c, generator=g).to(cuda)
    rs = (torch.rand(n, generator=g) > 0.3).float().to(cuda) / 0.7
    dm = (torch.randn(n, c, generator=g) * (1.0 if dtype == torch.bfloat16 else 0.25)).to(dtype).to(cuda)
    # ---- forward
    h, act = ops.linear_gelu_fwd(x, w1, b1)
    m_split = ops.spconv_fwd(act, w2[:, None, :].contiguous(), b2, None)
    m = ops.mlp_fwd(x, w1, b1, w2, b2)
    assert m.dtype == dtype and torch.equal(m, m_split)
    for scale in (rs, None):
        z_split, y_split, _ = ops.linear_joint_fwd(act, w2, b2, None, a, scale, None, dtype)
        z, y = ops.mlp_fwd(x, w1, b1, w2, b2, a, scale)
        assert torch.equal(z, z_split) and torch.equal(y, y_split), scale is not None
    z_only, none = ops.mlp_fwd(x, w1, b1, w2, b2, a, rs, want_y=False)
    assert none is None and torch.equal(z_only, ops.mlp_fwd(x, w1, b1, w2, b2, a, rs)[0])
    assert torch.equal(ops.mlp_fwd(x, w1, None, w2, None), ops.spconv_fwd(ops.linear_gelu_fwd(x, w1, None)[1], w2[:, None, :].contiguous(), None, None))
    xr, w1r, b1r = x.double().cpu().requires_grad_(True), w1.double().cpu().requires_grad_(True), b1.double().cpu().requires_grad_(True)
    w2r, b2r = w2.double().cpu().requires_grad_(True), b2.double().cpu().requires_grad_(True)
    ref = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xr, w1r, b1r)), w2r, b2r)
    lo = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -9
    assert float((m.double().cpu() - ref.detach()).norm() / ref.detach().norm()) < lo
    # ---- backward
    w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
    dh = ops.linear_gelu_bwd_input(dm, w2t, h)
    dx_split = ops.spconv_fwd(dh, w1t[:, None, :].contiguous(), None, None)
    dw2_s, db2_s = ops.spconv_wgrad(act, dm, None, want_bias=True)
    dw1_s, db1_s = ops.spconv_wgrad(x, dh, None, want_bias=True)
    dx, dw1, db1, dw2, db2 = ops.mlp_bwd(dm, x, w1, b1, w2t)
    assert torch.equal(dx, dx_split)
    for name, got, want in (("dw1", dw1, dw1_s[:, 0, :]), ("db1", db1, db1_s), ("dw2", dw2, dw2_s[:, 0, :]), ("db2", db2, db2_s)):
        assert got.dtype == torch.float32 and got.shape == want.shape, name
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-6, (name, float((got - want).abs().max()), float(want.abs().max()))
    again = ops.mlp_bwd(dm, x, w1, b1, w2t)
    assert all(torch.equal(u, v) for u, v in zip(again, (dx, dw1, db1, dw2, db2))), "not reproducible"
    dx2, dw1_2, none1, dw2_2, none2 = ops.mlp_bwd(dm, x, w1, None, w2t, want_b1=False, want_b2=False)
    assert none1 is None and none2 is None and dx2.shape == dx.shape
    ref.backward(dm.double().cpu())
    for name, got, want in (("dx", dx, xr.grad), ("dw1", dw1, w1r.grad), ("db1", db1, b1r.grad), ("dw2", dw2, w2r.grad), ("db2", db2, b2r.grad)):
        err = float((got.double().cpu() - want).norm() / want.norm())
        assert err < 4 * lo, (name, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [(32, 96), (128, 384), (256, 768)])   # the wide pairs: gemm3.h (gathered rows, kv = 2 input gradient)
def test_linear_gather_tables(cuda, dtype, cin, cout):
    """out = F.linear(x)[gidx] with a padded permutation (duplicated tail rows), gather-form backward."""
    from pointcept_amd import functional as PF

    n, n_pad = 3000, 3072
    g = torch.Generator().manual_seed(17)
    perm = torch.randperm(n, generator=g)
    gidx = torch.cat([perm, perm[n - 72 - 100:n - 100]])            # 72 padded slots repeat earlier points
    inv = torch.empty(n, dtype=torch.int64)
    inv[perm] = torch.arange(n)
    dup = torch.full((n,), -1, dtype=torch.int64)
    dup[gidx[n:]] = torch.arange(n, n_pad)
    x = torch.randn(n, cin, generator=g).to(dtype)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dtype)
    b = torch.randn(cout, generator=g)
    dout = torch.randn(n_pad, cout, generator=g).to(dtype)
    xr, wr, br = x.float().clone().requires_grad_(True), w.float().clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.linear(xr, wr, br)[gidx]
    ref.backward(dout.float())
    xe, we, be = x.to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    tf = gidx.to(torch.int32)[None].contiguous().to(cuda)
    tb = torch.stack([inv, dup]).to(torch.int32).contiguous().to(cuda)
    got = PF.linear(xe, we, be, tf, tb)
    got.backward(dout.to(cuda))
    rtol, atol = _tols(dtype)
    _close("glinear_fwd", got, ref, rtol, atol * 4)
    _close("glinear_dx", xe.grad, xr.grad, rtol, atol * 8)
    _close("glinear_dw", we.grad, wr.grad, 2 * rtol, 2e-3 * float(wr.grad.abs().max()))
    _close("glinear_db", be.grad, br.grad, 1e-4, 2e-3 * float(br.grad.abs().max()))
    # un-gather with dropped (non-primary) slots: out[p] = W a[inv[p]], backward table has -1 rows
    a = torch.randn(n_pad, cin, generator=g).to(dtype)
    prim = torch.where(inv[gidx] == torch.arange(n_pad), gidx, torch.full_like(gidx, -1))
    ar = a.float().clone().requires_grad_(True)
    ref2 = torch.nn.functional.linear(ar, w.float(), b)[inv]
    d2 = torch.randn(n, cout, generator=g).to(dtype)
    ref2.backward(d2.float())
    ae = a.to(cuda).requires_grad_(True)
    got2 = PF.linear(ae, w.to(cuda), b.to(cuda), inv.to(torch.int32)[None].contiguous().to(cuda),
                     prim.to(torch.int32)[None].contiguous().to(cuda))
    got2.backward(d2.to(cuda))
    _close("glinear2_fwd", got2, ref2, rtol, atol * 4)
    _close("glinear2_dx", ae.grad, ar.grad, rtol, atol * 8)


@pytest.mark.parametrize("c", [32, 64, 128, 256, 512,
                               36, 48, 54, 72, 96, 108, 144, 192, 216, 252, 384, 432, 504, 576, 1024, 10])   # round 5: the wave-per-row form
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
def test_layer_norm_fwd_bwd(cuda, c, xdt, ydt):
    """c in {32 .. 512}: the C / 8-lanes-per-row instances; every other width: the generic wave-per-row kernels -- the LayerNorm widths of
    PT-v3m2 (configs/sonata/*:45), PT-v3m3 (configs/utonia/*:21) and LitePT (litept_v1.py:601), plus the edges 10 and 1024"""
    from pointcept_amd import ops

    n = 70001 if c <= 64 else 5003
    g = torch.Generator().manual_seed(c)
    x = (torch.randn(n, c, generator=g) * 2 + 0.5).to(xdt)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    dy = torch.randn(n, c, generator=g).to(ydt)
    xr = x.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (c,), gr, br, 1e-5)
    ref.backward(dy.float())
    y, mean, rstd = ops.layer_norm_fwd(x.to(cuda), gamma.to(cuda), beta.to(cuda), 1e-5, ydt)
    rtol, atol = _tols(ydt)
    _close("ln_fwd", y, ref, rtol, atol * 4)
    _close("ln_mean", mean, x.float().mean(1), 1e-5, 1e-5)
    dx, dg, db = ops.layer_norm_bwd(dy.to(cuda), x.to(cuda), mean, rstd, gamma.to(cuda))
    rt2, at2 = _tols(xdt)
    _close("ln_dx", dx, xr.grad, rt2, at2 * 4)
    _close("ln_dgamma", dg, gr.grad, 1e-4, 1e-4 * float(gr.grad.abs().max()) + 1e-3)
    _close("ln_dbeta", db, br.grad, 1e-4, 1e-4 * float(br.grad.abs().max()) + 1e-3)


@pytest.mark.parametrize("c", [32, 128, 512, 48, 96, 192, 384, 432, 1024])       # round 5: the generic (wave-per-row) joint for the m2 / m3 / LitePT widths
@pytest.mark.parametrize("mode", ["ln_add_ln", "add_ln_scaled", "add_cast", "fp32", "f16_ln_add_ln", "f16_add_ln_scaled", "f16_add_cast"])
def test_add_norm_fused_joint(cuda, c, mode):
    """PF.add_norm == a + s * LN_A(u) followed by LN_B / cast, forward and every gradient; bf16, f16 (the reference's fp16 + GradScaler
    recipe runs the same fused joints) and fp32 operands."""
    from pointcept_amd import functional as PF

    n = 3001
    g = torch.Generator().manual_seed(c + len(mode))
    udt = torch.float32 if mode == "fp32" else (torch.float16 if mode.startswith("f16_") else torch.bfloat16)
    mode = mode[4:] if mode.startswith("f16_") else mode
    u = (torch.randn(n, c, generator=g) * 1.5).to(udt)
    a = torch.randn(n, c, generator=g)
    scale = (torch.rand(n, generator=g) > 0.3).float() / 0.7 if mode == "add_ln_scaled" else None
    na = torch.nn.LayerNorm(c) if mode in ("ln_add_ln", "fp32") else None
    nb = torch.nn.LayerNorm(c) if mode != "add_cast" else None
    for m in (na, nb):
        if m is not None:
            with torch.no_grad():
                m.weight.copy_(torch.rand(c, generator=g) + 0.5)
                m.bias.copy_(torch.randn(c, generator=g))
    dz = torch.randn(n, c, generator=g)
    dy = torch.randn(n, c, generator=g).to(udt)
    # reference (fp32 math on the same rounded inputs)
    ur, ar = u.float().clone().requires_grad_(True), a.clone().requires_grad_(True)
    fu = na(ur) if na is not None else ur
    zr = ar + (fu if scale is None else fu * scale[:, None])
    yr = nb(zr) if nb is not None else zr
    (zr * dz).sum().backward(retain_graph=True)
    (yr * dy.float()).sum().backward()
    ref_grads = {"u": ur.grad, "a": ar.grad}
    for name, m in (("A", na), ("B", nb)):
        if m is not None:
            ref_grads["g" + name], ref_grads["b" + name] = m.weight.grad.clone(), m.bias.grad.clone()
            m.weight.grad = None
            m.bias.grad = None
    # engine
    import copy
    na_e = copy.deepcopy(na).to(cuda) if na is not None else None
    nb_e = copy.deepcopy(nb).to(cuda) if nb is not None else None
    ue, ae = u.to(cuda).requires_grad_(True), a.to(cuda).requires_grad_(True)
    z, y = PF.add_norm(ue, ae, None if scale is None else scale.to(cuda), na_e, nb_e, udt)
    ((z * dz.to(cuda)).sum() + (y.float() * dy.to(cuda).float()).sum()).backward()
    rtol, atol = _tols(udt)
    _close("an_z", z, zr, 2e-5 if udt == torch.float32 else 1e-3, 2e-3 if udt != torch.float32 else 2e-5)
    _close("an_y", y, yr, rtol, atol * 4)
    _close("an_da", ae.grad, ref_grads["a"], 1e-3, 2e-3 * float(ref_grads["a"].abs().max()))
    _close("an_du", ue.grad, ref_grads["u"], rtol, atol * 4 + 4e-3 * float(ref_grads["u"].abs().max()) * (udt != torch.float32))
    for name, m in (("A", na_e), ("B", nb_e)):
        if m is not None:
            _close("an_dg" + name, m.weight.grad, ref_grads["g" + name], 1e-3, 2e-3 * float(ref_grads["g" + name].abs().max()))
            _close("an_db" + name, m.bias.grad, ref_grads["b" + name], 1e-3, 2e-3 * float(ref_grads["b" + name].abs().max()))


@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c", [64, 256])
def test_add_norm_bf16_residual_operand_and_unused_outputs(cuda, c, dt16):
    """First block of a stage: the residual operand `a` is the bf16 output of the pooling / unpooling.  The kernel reads it
    as bf16 (same values as a.float()) and writes da as bf16 (the cast autograd would apply); an output nobody uses gets no
    materialised zero gradient and the result equals the fp32-operand call on the same values."""
    from pointcept_amd import functional as PF

    n = 2049
    g = torch.Generator().manual_seed(c)
    u = (torch.randn(n, c, generator=g)).to(dt16).to(cuda)
    a16 = torch.randn(n, c, generator=g).to(dt16).to(cuda)
    nb = torch.nn.LayerNorm(c).to(cuda)
    dz, dy = torch.randn(n, c, generator=g).to(cuda), torch.randn(n, c, generator=g).to(dt16).to(cuda)
    res = {}
    for tag, a in (("bf16", a16.clone().requires_grad_(True)), ("fp32", a16.float().requires_grad_(True))):
        ue = u.clone().requires_grad_(True)
        nb.zero_grad(set_to_none=True)
        z, y = PF.add_norm(ue, a, None, None, nb, dt16)
        ((z * dz).sum() + (y.float() * dy.float()).sum()).backward()
        res[tag] = (z, y, a.grad, ue.grad, nb.weight.grad.clone())
    assert res["bf16"][2].dtype == dt16 and res["fp32"][2].dtype == torch.float32
    assert torch.equal(res["bf16"][0], res["fp32"][0]) and torch.equal(res["bf16"][1], res["fp32"][1])
    assert torch.equal(res["bf16"][2], res["fp32"][2].to(dt16))
    assert torch.equal(res["bf16"][3], res["fp32"][3]) and torch.equal(res["bf16"][4], res["fp32"][4])
    # only z used / only y used: the other gradient is absent, not zeros
    for use_z in (True, False):
        ue, ae = u.clone().requires_grad_(True), a16.float().requires_grad_(True)
        z, y = PF.add_norm(ue, ae, None, None, nb, dt16)
        ((z * dz).sum() if use_z else (y.float() * dy.float()).sum()).backward()
        ur, ar = u.float().requires_grad_(True), a16.float().requires_grad_(True)
        zr = ar + ur
        ((zr * dz).sum() if use_z else (torch.nn.functional.layer_norm(zr, (c,), nb.weight.detach(), nb.bias.detach(), nb.eps) * dy.float()).sum()).backward()
        _close("an_unused_da", ae.grad, ar.grad, 1e-3, 2e-3 * float(ar.grad.abs().max()))
        _close("an_unused_du", ue.grad, ur.grad, 1.0 / 64, 2e-2 * float(ur.grad.abs().max()))


def test_layer_norm_empty_and_unsupported(cuda):
    from pointcept_amd import ops
    from pointcept_amd._lib import PtcoreError

    assert not ops.layer_norm_supported(48) and ops.layer_norm_available(48) and ops.layer_norm_available(1024)
    assert not ops.layer_norm_available(49) and not ops.layer_norm_available(1026)
    for c in (49, 1026):
        with pytest.raises(PtcoreError):
            ops.layer_norm_fwd(torch.zeros(4, c, device=cuda), None, None, 1e-5, torch.float32)
    y, m, r = ops.layer_norm_fwd(torch.zeros(0, 64, device=cuda), None, None, 1e-5, torch.float32)
    assert y.shape == (0, 64)


@pytest.mark.parametrize("kw", [dict(bias=False), dict(elementwise_affine=False), dict()], ids=["no_bias", "no_affine", "affine"])
@pytest.mark.parametrize("c", [64, 48, 18])
def test_layer_norm_affine_variants(cuda, c, kw):
    """nn.LayerNorm's three parameter layouts (weight + bias, weight only, none) through the autograd function the module mirror calls:
    the backward returns no gradient for an absent input (autograd raises on one), and the present ones match torch's."""
    from pointcept_amd import functional as PF

    torch.manual_seed(c)
    ref = torch.nn.LayerNorm(c, **kw)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(torch.randn_like(p) * 0.3)
    par = {n: p.detach().to(cuda).requires_grad_(True) for n, p in ref.named_parameters()}
    x, dy = torch.randn(777, c), torch.randn(777, c)
    xe, xr = x.clone().to(cuda).requires_grad_(True), x.clone().requires_grad_(True)
    (PF.layer_norm(xe, par.get("weight"), par.get("bias"), ref.eps) * dy.to(cuda)).sum().backward()
    (ref(xr) * dy).sum().backward()
    _close("ln_var_dx", xe.grad, xr.grad, 1e-4, 1e-4 * float(xr.grad.abs().max()))
    for n, pr in ref.named_parameters():
        _close("ln_var_" + n, par[n].grad, pr.grad, 1e-4, 1e-4 * float(pr.grad.abs().max()))


# ------------------------------------------------------------------------------------------------
# H. attention
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lens,H", [([1024], 2), ([1024, 1024, 330], 4), ([48, 48, 17], 2), ([1, 2, 31, 32, 33, 65], 3),
                                    ([128] * 5, 8), ([1000, 24], 32)])
def test_attention_fwd_bwd(cuda, lens, H, monkeypatch):
    """forward and both forms of the backward (the two split kernels; the one-pass kernel of attention_bwd1.h, which launches of >= 176
    (sequence, head) units take by default) against the fp32 oracle; dK / dV of the two forms are the same sums in the same order
    (bit-identical), dQ differs in the order of its partial sums only."""
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(sum(lens) + H)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    qkv = (torch.randn(T, 3, H, 16, generator=g) * 1.5).to(torch.bfloat16)
    scale = 16 ** -0.5
    fro = lambda a, b: float((a.float().cpu() - b.detach()).norm() / b.detach().norm())  # noqa: E731
    out, lse = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), max(lens), scale)
    q32 = qkv.float().requires_grad_(True)
    ref, ref_lse = oops.attention_varlen(q32, cu, scale, return_lse=True)
    # bf16 output rounding (rtol) + bf16 rounding of P: the error of sum_k p_k v_k is ~ 2^-9 |v|_max for a
    # peaked row however small the result itself is (cancellation), hence atol scales with max |v|
    vmax = float(qkv[:, 2].float().abs().max())
    _close("attn_fwd", out, ref, 1.0 / 64, 2.0 ** -9 * vmax)
    _close("attn_lse", lse, ref_lse, 1e-3, 2e-2)         # denominator summed from bf16-rounded P
    dout = torch.randn(T, H, 16, generator=g).to(torch.bfloat16)
    ref.backward(dout.float())
    monkeypatch.setenv("PTC_AT_BWD1", "0")
    dqkv = ops.attn_varlen_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale)
    gmax = float(q32.grad.abs().max())
    _close("attn_bwd", dqkv, q32.grad, 1.0 / 32, 1e-2 * gmax)
    monkeypatch.setenv("PTC_AT_BWD1", "1")
    dqkv1 = ops.attn_varlen_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale)
    _close("attn_bwd one-pass", dqkv1, q32.grad, 1.0 / 32, 1e-2 * gmax)
    assert torch.equal(dqkv1[:, 1:], dqkv[:, 1:]), "dK / dV of the one-pass kernel differ from the split kernels'"
    assert torch.equal(dqkv1, ops.attn_varlen_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale)), "not reproducible"
    assert fro(dqkv1, q32.grad) < 2.0 ** -7, fro(dqkv1, q32.grad)
    # the elementwise bars above are set by the worst element of a cancelling sum; over the whole tensor the kernels sit at the
    # rounding floor of their bf16 operands (P, dS and the outputs are rounded to bf16 as flash-attn rounds them): measured on the
    # MI355X 2.1e-3 (forward) and 3.0-3.3e-3 (backward) relative Frobenius error at every shape (tools/attn_err_probe.py); bars = 2^-8 / 2^-7
    assert fro(out, ref) < 2.0 ** -8, fro(out, ref)
    assert fro(dqkv, q32.grad) < 2.0 ** -7, fro(dqkv, q32.grad)


@pytest.mark.parametrize("lens,H", [([1024] * 5 + [700, 33], 3), ([256, 1, 300], 2)])
def test_attention_forward_launch_plans_agree_bit_for_bit(cuda, lens, H, monkeypatch):
    """round 5: the forward keeps the first units of every XCD's chunk whole and cuts only the last partial round into 2 / 4 parts
    (at_plan_host in attention.hip; PTC_AT_PLAN = "whole,qs" forces a plan, "0" the round-4 uniform split).  A query row is computed
    by exactly one wave over all keys whichever part owns its tile: every plan must give the same bits, with ragged sequences, chunks
    shorter than `whole`, and more parts than a short sequence has query tiles."""
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(len(lens) + H)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).to(cuda)
    qkv = (torch.randn(T, 3, H, 16, generator=g) * 1.5).to(torch.bfloat16).to(cuda)
    monkeypatch.delenv("PTC_AT_PLAN", raising=False)
    out0, lse0 = ops.attn_varlen_fwd(qkv, cu, max(lens), 0.25)
    ref, ref_lse = oops.attention_varlen(qkv.float().cpu(), cu.cpu(), 0.25, return_lse=True)
    assert float((out0.float().cpu() - ref).norm() / ref.norm()) < 2.0 ** -8
    for plan in ("0", "0,1", "0,2", "0,4", "1,2", "1,4", "2,4", "3,2", "1000,4"):
        monkeypatch.setenv("PTC_AT_PLAN", plan)
        out, lse = ops.attn_varlen_fwd(qkv, cu, max(lens), 0.25)
        assert torch.equal(out, out0) and torch.equal(lse, lse0), plan


@pytest.mark.parametrize("n_shapes", [80])
def test_attention_launch_plan_cache_survives_more_shapes_than_it_holds(cuda, n_shapes, monkeypatch):
    """the per-thread plan cache (64 entries, keyed on workgroups per XCD chunk and query tiles, replaced round-robin) is
    swept with more distinct shapes than it holds, then the first shapes come back: every launch must still equal the uniform
    round-4 split bit for bit (every plan gives the same bits by construction, so what this guards is the replacement path itself:
    no overrun of the table, no launch refused, lookups after eviction still well-formed)."""
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(7)
    shapes = [(1 + 8 * i, 512 if i < 3 else 32) for i in range(n_shapes)] + [(1, 512), (9, 512), (17, 32)]
    for n_seq, L in shapes:
        lens = [L] * n_seq
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).to(cuda)
        qkv = torch.randn(sum(lens), 3, 1, 16, generator=g).to(torch.bfloat16).to(cuda)
        monkeypatch.delenv("PTC_AT_PLAN", raising=False)
        out, lse = ops.attn_varlen_fwd(qkv, cu, L, 0.25)
        monkeypatch.setenv("PTC_AT_PLAN", "0")
        out0, lse0 = ops.attn_varlen_fwd(qkv, cu, L, 0.25)
        assert torch.equal(out, out0) and torch.equal(lse, lse0), (n_seq, L)


@pytest.mark.parametrize("lens,H", [([1024, 330], 4), ([1, 2, 31, 32, 33, 65], 3)])
@pytest.mark.parametrize("one_pass", ["0", "1"])
def test_attention_f16_io_equals_the_reference_cast_passes(cuda, lens, H, one_pass, monkeypatch):
    """fp16 autocast call site (ptv3m1:209,215): flash_attn(qkv.to(bfloat16)).to(qkv.dtype) and its autograd.  With f16 tensors the kernels do
    the four casts in their load / store paths -- bit for bit the tensors the separate cast passes produce around the bf16 kernels
    (forward output, and dqkv for an f16 dout), including values that only f16 can hold (rounded to bf16 on the way in)."""
    from pointcept_amd import ops

    monkeypatch.setenv("PTC_AT_BWD1", one_pass)          # both forms of the backward
    g = torch.Generator().manual_seed(sum(lens) * 3 + H)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).to(cuda)
    qkv = (torch.randn(T, 3, H, 16, generator=g) * 1.5).to(torch.float16).to(cuda)        # 11-bit mantissas: .to(bfloat16) rounds
    scale = 16 ** -0.5
    out16, lse16 = ops.attn_varlen_fwd(qkv, cu, max(lens), scale)
    out_b, lse_b = ops.attn_varlen_fwd(qkv.to(torch.bfloat16), cu, max(lens), scale)
    assert out16.dtype == torch.float16 and torch.equal(out16, out_b.to(torch.float16)) and torch.equal(lse16, lse_b)
    dout = torch.randn(T, H, 16, generator=g).to(torch.float16).to(cuda)
    d16 = ops.attn_varlen_bwd(qkv, out16, dout, lse16, cu, max(lens), scale)
    d_b = ops.attn_varlen_bwd(qkv.to(torch.bfloat16), out_b, dout.to(torch.bfloat16), lse_b, cu, max(lens), scale)
    assert d16.dtype == torch.float16 and torch.equal(d16, d_b.to(torch.float16))


@pytest.mark.parametrize("lens,H,p", [([1024, 330], 4, 0.1), ([1, 2, 31, 32, 33, 65], 3, 0.25), ([200], 2, 0.5)])
def test_attention_dropout_fwd_bwd(cuda, lens, H, p):
    """Attention dropout on the flash path (ptv3m1:212, dropout_p = attn_drop in training): flash-attn's semantics -- softmax over all
    keys, then drop with probability p and rescale by 1 / (1 - p), lse of the undropped scores -- against the oracle applying the SAME
    keep mask (oracle/ops.py::attn_dropout_keep restates the kernels' integer hash), forward and backward; the mask is a function of
    the seed only (same seed: bit-identical results, another seed: another mask), its keep rate is 1 - p, and p = 0 through the same
    entry point reproduces the plain kernels' tolerance."""
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(sum(lens) + H)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    qkv = (torch.randn(T, 3, H, 16, generator=g) * 1.5).to(torch.bfloat16)
    scale, seed = 16 ** -0.5, 0x1234_5678_9ABC_DEF
    out, lse = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), max(lens), scale, p, seed)
    q32 = qkv.float().requires_grad_(True)
    ref, ref_lse = oops.attention_varlen(q32, cu, scale, return_lse=True, dropout_p=p, seed=seed)
    vmax = float(qkv[:, 2].float().abs().max())
    _close("attn_drop_fwd", out, ref, 1.0 / 64, 2.0 ** -9 * vmax / (1 - p))
    _close("attn_drop_lse", lse, ref_lse, 1e-3, 2e-2)
    dout = torch.randn(T, H, 16, generator=g).to(torch.bfloat16)
    ref.backward(dout.float())
    dqkv = ops.attn_varlen_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale, p, seed)
    gmax = float(q32.grad.abs().max())
    _close("attn_drop_bwd", dqkv, q32.grad, 1.0 / 32, 1e-2 * gmax)
    fro = lambda a, b: float((a.float().cpu() - b.detach()).norm() / b.detach().norm())
    assert fro(out, ref) < 2.0 ** -8 and fro(dqkv, q32.grad) < 2.0 ** -7, (fro(out, ref), fro(dqkv, q32.grad))
    again, _ = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), max(lens), scale, p, seed)
    other, _ = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), max(lens), scale, p, seed + 1)
    assert torch.equal(out, again) and not torch.equal(out, other)
    keep = oops.attn_dropout_keep(seed, 0, lens[0], lens[0], p).float().mean()
    if lens[0] >= 200:
        assert abs(float(keep) - (1 - p)) < 0.01, float(keep)


def test_attention_large_logits(cuda):
    """Peaked softmax (online-max path): one key dominates each query."""
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(3)
    L, H = 512, 2
    qkv = torch.randn(L, 3, H, 16, generator=g)
    qkv[:, 0] *= 6.0
    qkv[:, 1] *= 6.0
    qkv = qkv.to(torch.bfloat16)
    cu = torch.tensor([0, L], dtype=torch.int32)
    out, lse = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), L, 0.25)
    ref, ref_lse = oops.attention_varlen(qkv.float(), cu, 0.25, return_lse=True)
    _close("attn_fwd_peaked", out, ref, 1.0 / 64, 1e-2)
    _close("attn_lse_peaked", lse, ref_lse, 1e-3, 3e-2)


@pytest.mark.parametrize("D,lens,H", [(18, [1024, 700, 33], 3), (18, [1, 2, 31, 32, 33, 65], 6), (24, [1024, 330], 2),
                                      (32, [1024, 48, 17], 2), (17, [257, 64], 3), (40, [672, 100], 2), (48, [512, 512, 9], 4),
                                      (64, [512, 300], 2), (33, [96], 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_other_head_dims_fwd_bwd(cuda, D, lens, H, dtype):
    """head_dim 17..64 (PT-v3m3 / LitePT use 18, point_transformer_v3m3_utonia.py:354, litept_v1.py:244-256): the
    multi-slab kernels of attention_hd.h against the oracle, same bars as the head_dim-16 kernels.  dtype = float16 (round 4): f16 OPERANDS
    -- f16 MFMAs, P / dS rounded to f16, fp32 accumulation -- what flash-attn does with the fp16 tensors LitePT hands it
    (litept_v1.py:259-265); the whole-tensor error must then sit at the f16 rounding floor, eight times below the bf16 one."""
    from pointcept_amd import ops

    assert ops.attn_hd_supported(D, max(lens))
    g = torch.Generator().manual_seed(sum(lens) + H + D)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    qkv = (torch.randn(T, 3, H, D, generator=g) * 1.5).to(dtype)
    scale = D ** -0.5
    out, lse = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), max(lens), scale)
    assert out.shape == (T, H, D) and lse.shape == (H, T)
    q32 = qkv.float().requires_grad_(True)
    ref, ref_lse = oops.attention_varlen(q32, cu, scale, return_lse=True)
    vmax = float(qkv[:, 2].float().abs().max())
    _close(f"attn_hd{D}_fwd", out, ref, 1.0 / 64, 2.0 ** -9 * vmax)
    _close(f"attn_hd{D}_lse", lse, ref_lse, 1e-3, 2e-2)
    dout = torch.randn(T, H, D, generator=g).to(dtype)
    ref.backward(dout.float())
    dqkv = ops.attn_varlen_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale)
    assert out.dtype == dtype and dqkv.dtype == dtype
    gmax = float(q32.grad.abs().max())
    _close(f"attn_hd{D}_bwd", dqkv, q32.grad, 1.0 / 32, 1e-2 * gmax)
    d2 = ops.attn_varlen_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale)
    assert torch.equal(dqkv, d2), "backward is not bit-reproducible"
    fro = lambda a, b: float((a.float().cpu() - b.detach()).norm() / b.detach().norm())
    bar = (2.0 ** -8, 2.0 ** -7) if dtype == torch.bfloat16 else (2.0 ** -11, 2.0 ** -10)
    assert fro(out, ref) < bar[0] and fro(dqkv, q32.grad) < bar[1], (fro(out, ref), fro(dqkv, q32.grad), bar)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("lens,H", [([1024, 700, 33], 3), ([1, 2, 31, 32, 33, 65], 6)])
def test_attention_with_fused_rope_equals_the_two_pass_form(cuda, lens, H, dtype):
    """The 3-D rotary embedding in the attention prologue / epilogue (ptc_attn_varlen_hd_rope_*, head_dim 18) against the two-pass form it
    replaces -- ptc_rope3d_xyz on the packed rows, then the window-attention kernels, then the inverse rotation of the gradient: the same
    fp32 rotation rounded to the same operand dtype, so output, lse and the gradient of the UN-rotated qkv are bit-identical; and against the
    fp32 oracle through the reference's own formulation (rope_xyz_torch = utonia.py:58-101,303-323)."""
    from pointcept_amd import functional as PF
    from pointcept_amd import ops

    D = 18
    assert ops.attn_rope_supported(D, max(lens)) and not ops.attn_rope_supported(24, max(lens))
    g = torch.Generator().manual_seed(sum(lens) + H)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).to(cuda)
    qkv = (torch.randn(T, 3, H, D, generator=g) * 1.5).to(dtype).to(cuda)
    xyz = (torch.rand(T, 3, generator=g) * 40.0 - 20.0).to(cuda)
    inv_freq = (1.0 / (100.0 ** (torch.arange(3, dtype=torch.float32) / 3))).to(cuda)
    dout = torch.randn(T, H, D, generator=g).to(dtype).to(cuda)
    scale = D ** -0.5
    out, lse = ops.attn_rope_fwd(qkv, xyz, inv_freq, cu, max(lens), scale)
    dqkv = ops.attn_rope_bwd(qkv, out, dout, lse, xyz, inv_freq, cu, max(lens), scale)
    rot = ops.rope3d_xyz(qkv, xyz, inv_freq, 2, 1.0, dtype)
    out2, lse2 = ops.attn_varlen_fwd(rot, cu, max(lens), scale)
    d_rot = ops.attn_varlen_bwd(rot, out2, dout, lse2, cu, max(lens), scale)
    dqkv2 = ops.rope3d_xyz(d_rot, xyz, inv_freq, 2, -1.0, dtype)
    # the two forms run the same fp32 rotation and round to the same dtype; the compiler may still contract the inlined sincos /
    # products of the two translation units differently (measured on the MI355X: 1 of 17 712 f16 gradient elements one ulp apart, bf16
    # and every forward tensor identical; the host emulation: everything identical): at most 0.1 % of the elements, at most two ulps (11 of 284 634 at the large shape)
    # A rotated operand that lands one ulp apart perturbs every product of its token's row, so the bar of an element is two ulps of
    # the LARGEST magnitude of its token (an element-relative bar fails on the small entries of such a row: 2^-11 absolute on an
    # entry of 0.2 beside entries of 0.4, MI355X, round 4).
    def same(name, x, y):
        d = (x.float() - y.float()).abs()
        row = torch.maximum(x.float().abs(), y.float().abs()).flatten(1).amax(1).clamp_min(1e-3)
        ulp = ((2.0 ** -6 if dtype == torch.bfloat16 else 2.0 ** -9) * row).reshape(-1, *([1] * (x.dim() - 1)))      # two ulps
        assert int((d > 0).sum()) <= max(1, x.numel() // 1000) and bool((d <= ulp).all()), (name, int((d > 0).sum()), float(d.max()))

    same("out", out, out2)
    assert float((lse - lse2).abs().max()) <= 1e-5
    same("dqkv", dqkv, dqkv2)
    # autograd wrapper: the fused operator is what PF.attn_rope_qkvpacked picks for this shape
    q1 = qkv.clone().requires_grad_(True)
    o1 = PF.attn_rope_qkvpacked(q1, xyz, inv_freq, cu, max(lens), scale, dtype)
    o1.backward(dout)
    assert torch.equal(o1.detach(), out) and torch.equal(q1.grad, dqkv)
    # against the reference's formulation in fp32
    q32 = qkv.float().cpu().requires_grad_(True)
    n = T
    emb = xyz.cpu()[:, :, None] * inv_freq.cpu()[None, None, :]
    cos, sin = emb.cos()[:, None, None, :, None, :], emb.sin()[:, None, None, :, None, :]
    t = q32[:, :2].reshape(n, 2, H, 3, 2, 3)
    u, v = t[..., 0:1, :], t[..., 1:2, :]
    rot32 = torch.cat((torch.cat((u * cos - v * sin, v * cos + u * sin), dim=-2).reshape(n, 2, H, D), q32[:, 2:]), dim=1)
    ref = oops.attention_varlen(rot32, cu.cpu(), scale)
    ref.backward(dout.float().cpu())
    fro = lambda a, b: float((a.float().cpu() - b.detach()).norm() / b.detach().norm())
    bar = (2.0 ** -7, 2.0 ** -6) if dtype == torch.bfloat16 else (2.0 ** -10, 2.0 ** -9)      # operand rounding of the rotated q / k on top of the kernels'
    assert fro(out, ref) < bar[0] and fro(dqkv, q32.grad) < bar[1], (fro(out, ref), fro(dqkv, q32.grad))


def test_attention_other_head_dims_large_logits_and_limits(cuda):
    """Peaked rows take the online-softmax loop; windows that do not fit LDS are refused (by the flash_attn mirror too:
    there is no library path behind it); a sequence longer than max_seqlen poisons its rows instead of overrunning LDS."""
    from pointcept_amd import ops
    from pointcept_amd._lib import PtcoreError
    from pointcept_amd.flash_attn_api import flash_attn_varlen_qkvpacked_func

    g = torch.Generator().manual_seed(5)
    L, H, D = 512, 2, 18
    qkv = torch.randn(L, 3, H, D, generator=g)
    qkv[:, 0] *= 6.0
    qkv[:, 1] *= 6.0
    qkv = qkv.to(torch.bfloat16)
    cu = torch.tensor([0, L], dtype=torch.int32)
    out, lse = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), L, 0.25)
    ref, ref_lse = oops.attention_varlen(qkv.float(), cu, 0.25, return_lse=True)
    _close("attn_hd_fwd_peaked", out, ref, 1.0 / 64, 1e-2)
    _close("attn_hd_lse_peaked", lse, ref_lse, 1e-3, 3e-2)

    assert ops.attn_hd_supported(32, 1024) and ops.attn_hd_supported(48, 672) and ops.attn_hd_supported(64, 512)
    assert not ops.attn_hd_supported(48, 1024) and not ops.attn_hd_supported(64, 1024) and not ops.attn_hd_supported(72, 64)
    big = torch.randn(1024, 3, 2, 64, generator=g).to(torch.bfloat16).to(cuda)
    cu1 = torch.tensor([0, 1024], dtype=torch.int32, device=cuda)
    with pytest.raises(PtcoreError):
        ops.attn_varlen_fwd(big, cu1, 1024, 0.125)
    with pytest.raises(PtcoreError):                                  # round 4: no library (SDPA) backend behind the mirror
        flash_attn_varlen_qkvpacked_func(big, cu1, 1024)

    short = (torch.randn(200, 3, 2, 18, generator=g)).to(torch.bfloat16).to(cuda)
    cu2 = torch.tensor([0, 200], dtype=torch.int32, device=cuda)
    o, l = ops.attn_varlen_fwd(short, cu2, 64, 0.25)                    # max_seqlen lies: 200 > 64
    assert torch.isnan(o.float()).all() and torch.isnan(l).all()


@pytest.mark.parametrize("one_pass", ["0", "1"])
def test_attention_backward_poisons_a_sequence_longer_than_max_seqlen(cuda, one_pass, monkeypatch):
    """head_dim 16, both forms of the backward: cu_seqlens built for longer windows than the caller's max_seqlen (the LDS images are
    sized from max_seqlen) must not overrun anything -- the overlong sequence's gradient rows come back NaN (loud in the loss), every
    other sequence's gradient is what it is without the bad neighbour."""
    from pointcept_amd import ops

    monkeypatch.setenv("PTC_AT_BWD1", one_pass)
    g = torch.Generator().manual_seed(5)
    lens, H = [40, 100, 64], 2                                        # max_seqlen = 64: the middle sequence is too long
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).to(cuda)
    qkv = torch.randn(T, 3, H, 16, generator=g).to(torch.bfloat16).to(cuda)
    dout = torch.randn(T, H, 16, generator=g).to(torch.bfloat16).to(cuda)
    out, lse = ops.attn_varlen_fwd(qkv, cu, 64, 0.25)
    d = ops.attn_varlen_bwd(qkv, out, dout, lse, cu, 64, 0.25)
    assert torch.isnan(d[40:140].float()).all()
    keep = torch.cat([torch.arange(0, 40), torch.arange(140, T)]).to(cuda)
    cu_ok = torch.tensor([0, 40, 104], dtype=torch.int32).to(cuda)
    q2 = qkv[keep].contiguous()
    o2, l2 = ops.attn_varlen_fwd(q2, cu_ok, 64, 0.25)
    d2 = ops.attn_varlen_bwd(q2, o2, dout[keep].contiguous(), l2, cu_ok, 64, 0.25)
    assert torch.equal(d[keep], d2)


def test_flash_attn_api_head_dim_18_autograd(cuda):
    """The call PT-v3m3 / LitePT make (head_dim 18) through the flash_attn mirror, with autograd."""
    from pointcept_amd.flash_attn_api import flash_attn_varlen_qkvpacked_func

    g = torch.Generator().manual_seed(8)
    lens, H, D = [1024, 1024, 513], 6, 18
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    x = (torch.randn(T, 3, H, D, generator=g)).to(torch.bfloat16)
    xq = x.to(cuda).requires_grad_(True)
    o = flash_attn_varlen_qkvpacked_func(xq, cu.to(cuda), max_seqlen=1024, dropout_p=0.0, softmax_scale=D ** -0.5)
    w = torch.randn(T, H, D, generator=g)
    (o.float() * w.to(cuda)).sum().backward()
    x32 = x.float().requires_grad_(True)
    r = oops.attention_varlen(x32, cu, D ** -0.5)
    (r * w).sum().backward()
    _close("fa18_fwd", o, r, 1.0 / 64, 2.0 ** -9 * float(x[:, 2].float().abs().max()))
    _close("fa18_bwd", xq.grad, x32.grad, 1.0 / 32, 1e-2 * float(x32.grad.abs().max()))


@pytest.mark.parametrize("D,dtype", [(8, torch.bfloat16), (12, torch.float16), (3, torch.bfloat16)])
def test_flash_attn_api_small_heads(cuda, D, dtype):
    """head_dim below one MFMA k-step through the flash_attn mirror: zero-padded to 16 on the way in, cut on the way out; the default
    softmax scale is that of the caller's head_dim."""
    from pointcept_amd.flash_attn_api import flash_attn_varlen_qkvpacked_func

    g = torch.Generator().manual_seed(80 + D)
    lens, H = [700, 1, 64, 333], 5
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    x = (torch.randn(T, 3, H, D, generator=g) * 1.5).to(dtype)
    xq = x.to(cuda).requires_grad_(True)
    o = flash_attn_varlen_qkvpacked_func(xq, cu.to(cuda), max_seqlen=700)
    assert o.shape == (T, H, D) and o.dtype == dtype
    w = torch.randn(T, H, D, generator=g)
    (o.float() * w.to(cuda)).sum().backward()
    x32 = x.float().requires_grad_(True)
    r = oops.attention_varlen(x32, cu, D ** -0.5)
    (r * w).sum().backward()
    _close("fa_small_fwd", o, r, 1.0 / 64, 2.0 ** -8 * float(x[:, 2].float().abs().max()))
    _close("fa_small_bwd", xq.grad, x32.grad, 1.0 / 32, 1.5e-2 * float(x32.grad.abs().max()))


def _rpe_reference(qkv, cu, scale, gc, table, bnd):
    """ptv3m1:29-48,190-206 on the CPU in fp32, window by window (the oracle of the RPE kernels)."""
    T, _, H, D = qkv.shape
    R = 2 * bnd + 1
    out = torch.zeros(T, H, D)
    lse = torch.zeros(H, T)
    for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):
        if b <= a:
            continue
        q, k, v = (qkv[a:b, j].permute(1, 0, 2) for j in range(3))                  # [H, L, D]
        rel = gc[a:b, None, :].long() - gc[None, a:b, :].long()                     # [L(query), L(key), 3]
        idx = rel.clamp(-bnd, bnd) + bnd + torch.arange(3) * R
        bias = table[idx.reshape(-1)].view(b - a, b - a, 3, H).sum(2).permute(2, 0, 1)
        logits = (q * scale) @ k.transpose(1, 2) + bias
        lse[:, a:b] = torch.logsumexp(logits, dim=-1)
        out[a:b] = (torch.softmax(logits, dim=-1) @ v).permute(1, 0, 2)
    return out, lse


@pytest.mark.parametrize("lens,H,bnd", [([256, 256], 2, 20), ([1024, 1024, 1024], 4, 32), ([200, 200, 200], 3, 18), ([33], 1, 4),
                                        ([1024, 330], 2, 32)])
def test_attention_rpe_fwd_bwd(cuda, lens, H, bnd):
    """relative-position-bias attention kernels against the reference formulation (fp32, CPU): output, lse,
    dqkv and the table gradient (2^-24 fixed-point integer atomics: bit-reproducible as well)."""
    from pointcept_amd import functional as PF
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(sum(lens) + H + bnd)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    qkv = (torch.randn(T, 3, H, 16, generator=g) * 1.2).to(torch.bfloat16)
    gc = torch.randint(0, 3 * bnd, (T, 3), generator=g).to(torch.int32)          # offsets beyond +-bnd get clamped
    gc[::7] += 40000                                                               # large coordinates: 16-bit packing
    table = torch.randn(3 * (2 * bnd + 1), H, generator=g) * 0.5
    scale = 0.25
    assert ops.attn_rpe_supported(16, max(lens), bnd)
    out, lse = ops.attn_rpe_fwd(qkv.to(cuda), cu.to(cuda), max(lens), scale, gc.to(cuda), table.to(cuda), bnd)
    q32, t32 = qkv.float().requires_grad_(True), table.clone().requires_grad_(True)
    ref, ref_lse = _rpe_reference(q32, cu, scale, gc, t32, bnd)
    vmax = float(qkv[:, 2].float().abs().max())
    _close("rpe_fwd", out, ref, 1.0 / 64, 2.0 ** -9 * vmax)
    _close("rpe_lse", lse, ref_lse, 1e-3, 2e-2)
    dout = torch.randn(T, H, 16, generator=g).to(torch.bfloat16)
    ref.backward(dout.float())
    dqkv, dtab = ops.attn_rpe_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale, gc.to(cuda), table.to(cuda), bnd)
    _close("rpe_dqkv", dqkv, q32.grad, 1.0 / 32, 1e-2 * float(q32.grad.abs().max()))
    _close("rpe_dtable", dtab, t32.grad, 2e-2, 1e-2 * float(t32.grad.abs().max()))
    d2, t2 = ops.attn_rpe_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu.to(cuda), max(lens), scale, gc.to(cuda), table.to(cuda), bnd)
    assert torch.equal(dqkv, d2), "dqkv is not bit-reproducible"
    assert torch.equal(dtab, t2), "the fixed-point table gradient is not bit-reproducible"
    # autograd wrapper
    xq, tq = qkv.to(cuda).requires_grad_(True), table.to(cuda).requires_grad_(True)
    o = PF.attn_rpe_qkvpacked(xq, cu.to(cuda), max(lens), scale, gc.to(cuda), tq, bnd)
    (o.float() * dout.to(cuda).float()).sum().backward()
    assert torch.equal(xq.grad, dqkv)
    _close("rpe_dtable_autograd", tq.grad, t32.grad, 2e-2, 1e-2 * float(t32.grad.abs().max()))


@pytest.mark.parametrize("lens,H,bnd", [([200, 200, 200], 3, 18), ([33], 1, 4), ([1024, 330], 2, 32)])
def test_attention_rpe_f16_io_equals_the_cast_passes(cuda, lens, H, bnd):
    """round 6: under the reference's fp16 AMP (configs/s3dis/semseg-pt-v3m1-1-rpe.py) qkv arrives in f16.  The RPE
    kernels take f16 tensors directly -- the cast to their bf16 operands in the load path, the cast back in the store path -- and must
    equal, bit for bit, the bf16 kernels between explicit cast passes; against the fp32 formulation they sit at the bf16 kernels' bars."""
    from pointcept_amd import functional as PF
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(sum(lens) + H + bnd)
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).to(cuda)
    qkv = (torch.randn(T, 3, H, 16, generator=g) * 1.2).to(torch.float16)
    gc = torch.randint(0, 3 * bnd, (T, 3), generator=g).to(torch.int32)
    table = torch.randn(3 * (2 * bnd + 1), H, generator=g) * 0.5
    dout = torch.randn(T, H, 16, generator=g).to(torch.float16)
    scale = 0.25
    out, lse = ops.attn_rpe_fwd(qkv.to(cuda), cu, max(lens), scale, gc.to(cuda), table.to(cuda), bnd)
    assert out.dtype == torch.float16
    outb, lseb = ops.attn_rpe_fwd(qkv.to(torch.bfloat16).to(cuda), cu, max(lens), scale, gc.to(cuda), table.to(cuda), bnd)
    assert torch.equal(out, outb.to(torch.float16)) and torch.equal(lse, lseb)
    dqkv, dtab = ops.attn_rpe_bwd(qkv.to(cuda), out, dout.to(cuda), lse, cu, max(lens), scale, gc.to(cuda), table.to(cuda), bnd)
    # the cast-pass form: every 16-bit tensor the kernels read is the bf16 rounding of the f16 one (`out` of the f16 run: bf16 -> f16 -> bf16 is exact)
    dqb, dtb = ops.attn_rpe_bwd(qkv.to(torch.bfloat16).to(cuda), outb, dout.to(torch.bfloat16).to(cuda), lseb, cu, max(lens), scale,
                                gc.to(cuda), table.to(cuda), bnd)
    assert dqkv.dtype == torch.float16 and torch.equal(dqkv, dqb.to(torch.float16)) and torch.equal(dtab, dtb)
    q32, t32 = qkv.float().requires_grad_(True), table.clone().requires_grad_(True)
    ref, _ = _rpe_reference(q32, torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32), scale, gc, t32, bnd)
    ref.backward(dout.float())
    _close("rpe_fwd f16", out, ref, 1.0 / 64, 2.0 ** -8 * float(qkv[:, 2].float().abs().max()))
    _close("rpe_dqkv f16", dqkv, q32.grad, 1.0 / 32, 2e-2 * float(q32.grad.abs().max()))
    _close("rpe_dtable f16", dtab, t32.grad, 2e-2, 2e-2 * float(t32.grad.abs().max()))
    xq, tq = qkv.to(cuda).requires_grad_(True), table.to(cuda).requires_grad_(True)
    o = PF.attn_rpe_qkvpacked(xq, cu, max(lens), scale, gc.to(cuda), tq, bnd)
    (o.float() * dout.to(cuda).float()).sum().backward()
    assert xq.grad.dtype == torch.float16 and torch.equal(xq.grad, dqkv)


# ------------------------------------------------------------------------------------------------
# I. ends of the step: coordinate maxima, cross entropy
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 63, 100000])
@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
def test_coord_max(cuda, n, dtype):
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(n + 1)
    gc = torch.randint(0, 60000, (n, 3), generator=g).to(dtype)
    got = ops.coord_max(gc.to(cuda)).cpu()
    ref = gc.max(0).values.to(torch.int64) if n else torch.zeros(3, dtype=torch.int64)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c,strided", [(1, 20, False), (1000, 20, True), (70001, 16, False), (513, 13, True), (700, 13, False),
                                         (300, 40, True)])
def test_cross_entropy_fwd_bwd(cuda, dtype, n, c, strided):
    """CrossEntropyLoss(ignore_index=-1, mean) of pointcept/models/losses/misc.py: loss and gradient vs
    torch on the same (rounded) logits in fp32; strided = a [:, :c] view of a wider head output.  The cases cover the row loads of
    csrc/loss_rows.h: 16-byte (64- / 128-byte rows), 8-byte (dense 20 x bf16), element-wise (dense 13 columns), and the kernels for
    more than 32 classes."""
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(n * 31 + c)
    wide = (torch.randn(n, 32 if c <= 32 else 64, generator=g) * 3).to(dtype)
    tgt = torch.randint(0, c, (n,), generator=g)
    tgt[torch.rand(n, generator=g) < 0.1] = -1
    if n == 1:
        tgt[0] = 3
    base = wide.clone().to(cuda).requires_grad_(True)      # (clone: on the CPU test tiers .to() would alias `wide`)
    logits = base[:, :c] if strided else base[:, :c].contiguous()
    loss = PF.cross_entropy(logits, tgt.to(cuda), -1)
    loss.backward()
    ref_in = wide[:, :c].float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, tgt, ignore_index=-1)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref)))
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8 * float(ref_in.grad.abs().max())
    _close("ce_grad", base.grad[:, :c], ref_in.grad, 1e-5 if dtype == torch.float32 else 2.0 ** -7, tol)
    assert float(base.grad[:, c:].abs().max()) == 0.0
    again = PF.cross_entropy(logits.detach(), tgt.to(cuda), -1)
    assert float(again) == float(loss), "deterministic reduction"


# ------------------------------------------------------------------------------------------------
# J. BatchNorm1d + activation
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c", [(2, 32), (1000, 32), (70001, 64), (513, 96), (4097, 512),
                                 (3001, 36), (2500, 54), (1777, 108), (1200, 252), (900, 6)])   # round 6: 8- / 4-byte lanes (LitePT 36 / 252, PT-v3m3 54 / 108)
@pytest.mark.parametrize("act", ["none", "gelu", "relu"])
def test_batch_norm_act_train(cuda, dtype, n, c, act):
    """training mode: output, running statistics, dx / dgamma / dbeta vs torch (fp32 on the same rounded input).  Channel counts that
    are no multiple of a 16-byte lane run on the narrow-lane instances of the same kernels (ops.batch_norm_supported says so)."""
    from pointcept_amd import ops as _ops
    assert _ops.batch_norm_supported(c, dtype) or torch.device(cuda).type != "cuda"
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(n + c)
    x = (torch.randn(n, c, generator=g) * 2 + 0.7).to(dtype)
    w = torch.rand(c, generator=g) + 0.5
    b = torch.randn(c, generator=g) * 0.3
    dy = torch.randn(n, c, generator=g).to(dtype)
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    fn = {"none": lambda t: t, "gelu": torch.nn.functional.gelu, "relu": torch.relu}[act]
    xr, wr, br = x.float().clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    ref = fn(torch.nn.functional.batch_norm(xr, rm, rv, wr, br, True, 0.01, 1e-3))
    ref.backward(dy.float())
    xg, wg, bg = x.to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    rmg, rvg = rm0.to(cuda), rv0.to(cuda)
    got = PF.batch_norm_act(xg, wg, bg, rmg, rvg, True, 0.01, 1e-3, act)
    got.backward(dy.to(cuda))
    assert got.dtype == dtype
    lo = dtype == torch.bfloat16
    _close("bn_y", got, ref, 2.0 ** -7 if lo else 2e-5, 2e-2 if lo else 2e-5)
    _close("bn_running_mean", rmg, rm, 1e-5, 1e-5)
    _close("bn_running_var", rvg, rv, 1e-5, 1e-5)
    gmax = float(xr.grad.abs().max())
    _close("bn_dx", xg.grad, xr.grad, 2.0 ** -6 if lo else 1e-4, (2e-2 if lo else 1e-4) * max(gmax, 1e-3))
    _close("bn_dgamma", wg.grad, wr.grad, 1e-3, 1e-3 * float(wr.grad.abs().max()) + 1e-4)
    _close("bn_dbeta", bg.grad, br.grad, 1e-3, 1e-3 * float(br.grad.abs().max()) + 1e-4)
    again = PF.batch_norm_act(x.to(cuda), w.to(cuda), b.to(cuda), rm0.to(cuda), rv0.to(cuda), True, 0.01, 1e-3, act)
    assert torch.equal(again, got.detach()), "bit-reproducible"


@pytest.mark.parametrize("dtype,n,c", [(torch.float32, 3000, 64), (torch.bfloat16, 5003, 96), (torch.float16, 2048, 32)])
def test_batch_norm_add_act_is_the_residual_block_tail(cuda, dtype, n, c):
    """relu(BN(x) + residual) in the BatchNorm's apply pass (ptc_batch_norm_add_act_{fwd,bwd}; spconv_unet_v1m1_base.py:79-83 runs it as
    bn2, an add and a ReLU): output, dx, d residual, dgamma, dbeta and the running statistics against torch in fp32 on the same rounded
    inputs; the residual does not enter the statistics; bit-reproducible."""
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(n * 3 + c)
    x = (torch.randn(n, c, generator=g) * 2 + 0.7).to(dtype)
    r = (torch.randn(n, c, generator=g) * 1.5).to(dtype)
    w, b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    dy = torch.randn(n, c, generator=g).to(dtype)
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    xr, rr = x.float().clone().requires_grad_(True), r.float().clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    ref = torch.relu(torch.nn.functional.batch_norm(xr, rm, rv, wr, br, True, 0.01, 1e-3) + rr)
    ref.backward(dy.float())
    xg, rg = x.to(cuda).requires_grad_(True), r.to(cuda).requires_grad_(True)
    wg, bg = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    rmg, rvg = rm0.to(cuda), rv0.to(cuda)
    got = PF.batch_norm_act(xg, wg, bg, rmg, rvg, True, 0.01, 1e-3, "relu", rg)
    got.backward(dy.to(cuda))
    assert got.dtype == dtype and rg.grad.dtype == dtype
    lo = dtype != torch.float32
    _close("bnr_y", got, ref, 2.0 ** -7 if lo else 2e-5, 2e-2 if lo else 2e-5)
    _close("bnr_running_mean", rmg, rm, 1e-5, 1e-5)
    _close("bnr_running_var", rvg, rv, 1e-5, 1e-5)
    # the ReLU mask is decided on the fp32 pre-activation here and in the reference: elements within rounding of zero may differ
    mism = ((got.float().cpu() > 0) != (ref > 0)).float().mean()
    assert float(mism) < 1e-3, float(mism)
    gmax = float(xr.grad.abs().max())
    _close("bnr_dx", xg.grad, xr.grad, 2.0 ** -6 if lo else 1e-4, (2e-2 if lo else 1e-4) * max(gmax, 1e-3))
    _close("bnr_dres", rg.grad, rr.grad, 2.0 ** -7 if lo else 1e-6, 1e-6)
    _close("bnr_dgamma", wg.grad, wr.grad, 1e-3, 1e-3 * float(wr.grad.abs().max()) + 1e-4)
    _close("bnr_dbeta", bg.grad, br.grad, 1e-3, 1e-3 * float(br.grad.abs().max()) + 1e-4)
    again = PF.batch_norm_act(x.to(cuda), w.to(cuda), b.to(cuda), rm0.to(cuda), rv0.to(cuda), True, 0.01, 1e-3, "relu", r.to(cuda))
    assert torch.equal(again, got.detach()), "bit-reproducible"


def test_batch_norm_act_eval_and_module(cuda):
    """eval mode uses the running statistics; the nn.Module wrapper keeps nn.BatchNorm1d's state dict."""
    from pointcept_amd import nn as PNN

    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.BatchNorm1d(64, eps=1e-3, momentum=0.01), torch.nn.GELU())
    from pointcept_amd.point_transformer_v3 import PointSequential
    eng = PointSequential(PNN.BatchNorm1d(64, eps=1e-3, momentum=0.01), PNN.GELU())     # runs as ONE pass (PNN.fused_act)
    assert PNN.fused_act(eng[0], eng[1]) == "gelu"
    with torch.no_grad():
        ref[0].weight.uniform_(0.5, 1.5); ref[0].bias.normal_()
    assert set(eng.state_dict()) == set(ref.state_dict())
    eng.load_state_dict(ref.state_dict())
    eng = eng.to(cuda)
    x = torch.randn(3000, 64) * 1.5 + 0.2
    for _ in range(3):   # training steps move the running statistics identically
        yr, ye = ref(x), eng(x.to(cuda))
        _close("bn_module_train", ye, yr, 2e-5, 2e-5)
    assert int(eng[0].num_batches_tracked) == int(ref[0].num_batches_tracked) == 3
    _close("bn_module_rm", eng[0].running_mean, ref[0].running_mean, 1e-5, 1e-6)
    _close("bn_module_rv", eng[0].running_var, ref[0].running_var, 1e-5, 1e-6)
    ref.eval(); eng.eval()
    _close("bn_module_eval", eng(x.to(cuda)), ref(x), 2e-5, 2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c", [(1, 32), (4099, 64), (70001, 96), (333, 20)])
def test_column_sum(cuda, dtype, n, c):
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, generator=g).to(dtype)
    got = ops.column_sum(x.to(cuda))
    ref = x.double().sum(0)
    _close("column_sum", got.double(), ref, 1e-5, 1e-4 * max(1.0, float(ref.abs().max())))
    assert got.dtype == torch.float32


@pytest.mark.parametrize("n,c", [(5000, 32), (4097, 64), (1000, 128), (333, 256), (2111, 512)])    # (512: gemm3.h's GELU epilogues, round 6)
def test_mlp_gelu_fused(cuda, n, c):
    """fc1 -> GELU -> fc2 with GELU / GELU' fused into the GEMM epilogues vs torch fp32 on the same rounded
    operands (bf16 autocast): output, input gradient, all four parameter gradients."""
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(n + c)
    x = (torch.randn(n, c, generator=g)).to(torch.bfloat16)
    w1 = torch.randn(4 * c, c, generator=g) / c ** 0.5
    b1 = torch.randn(4 * c, generator=g) * 0.1
    w2 = torch.randn(c, 4 * c, generator=g) / (4 * c) ** 0.5
    b2 = torch.randn(c, generator=g) * 0.1
    dy = torch.randn(n, c, generator=g).to(torch.bfloat16)
    ps = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    xr = x.float().clone().requires_grad_(True)
    rnd = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    h = rnd(xr @ rnd(ps[0]).t() + ps[1])
    a = rnd(torch.nn.functional.gelu(h))
    ref = a @ rnd(ps[2]).t() + ps[3]
    ref.backward(dy.float())
    pg = [t.to(cuda).requires_grad_(True) for t in (w1, b1, w2, b2)]
    xg = x.to(cuda).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert PF.mlp_gelu_supported(xg, pg[0], pg[2])
        out = PF.mlp_gelu(xg, *pg)
    out.backward(dy.to(cuda))
    _close("mlp_out", out, ref, 2.0 ** -7, 2e-2)
    _close("mlp_dx", xg.grad, xr.grad, 2.0 ** -6, 2e-2 * float(xr.grad.abs().max()))
    for name, a_, b_ in zip(("dw1", "db1", "dw2", "db2"), pg, ps):
        _close("mlp_" + name, a_.grad, b_.grad, 2.0 ** -6, 2e-2 * float(b_.grad.abs().max()))


# ------------------------------------------------------------------------------------------------
# K. BASELINE full size (8 x 102400 voxels): size-independent properties (the oracle is too slow here)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_scene(cuda):
    from pointcept_amd import ops, synthetic

    b = synthetic.indoor_batch(8, 102400)
    ind = np.concatenate([omaps.offset2batch(b["offset"])[:, None], b["grid_coord"]], axis=1).astype(np.int32)
    ind_t = _t(ind, cuda)
    table = ops.HashTable(ind_t)
    return ind_t, table, ops.rulebook_subm(ind_t, 3, table)


def test_full_size_rulebook_properties(cuda, full_scene):
    """k=3 and k=5 tables of the bench batch: centre column is the identity, entries are in range, the map is
    symmetric (nbr[k][i] = j <=> nbr[K-1-k][j] = i: the property dgrad relies on), pairs/voxel matches the count of
    occupied neighbour cells computed independently (sorted-key membership test on the device)."""
    from pointcept_amd import ops

    ind, table, nbr3 = full_scene
    n = ind.shape[0]
    ar = torch.arange(n, device=cuda, dtype=torch.int32)
    for ks, nbr in ((3, nbr3), (5, ops.rulebook_subm(ind, 5, table))):
        kv = ks ** 3
        assert nbr.shape == (kv, n) and bool((nbr[kv // 2] == ar).all())
        assert int(nbr.max()) < n and int(nbr.min()) >= -1
        for k in (0, 1, kv // 3, kv // 2 - 1):   # spot-check the symmetry on a few offsets (all rows of those offsets)
            j = nbr[k].long()
            valid = j >= 0
            back = nbr[kv - 1 - k][j[valid]]
            assert bool((back == ar[valid]).all())
        # independent pair count for offset (+1, 0, 0): key membership
        ind64 = ind.long()
        key = ((ind64[:, 0] << 54) | (ind64[:, 1] << 36) | (ind64[:, 2] << 18) | ind64[:, 3])
        skey = torch.sort(key).values
        r = ks // 2
        q = key + (1 << 36)
        pos = torch.searchsorted(skey, q).clamp(max=n - 1)
        cnt = int((skey[pos] == q).sum())
        k_plus_x = ((1 + r) * ks + r) * ks + r
        assert int((nbr[k_plus_x] >= 0).sum()) == cnt


@pytest.mark.parametrize("c", [64, 96])
def test_full_size_conv_is_a_gather_for_one_hot_weights(cuda, full_scene, c):
    """W = identity at ONE offset, zero elsewhere: the convolution must reproduce in[nbr[k]] BIT-EXACTLY (products with
    1.0 and sums with 0.0 are exact in the MFMA): checks tables, gathers, chunk pipeline and epilogue at full size."""
    from pointcept_amd import ops

    ind, table, nbr = full_scene
    n = ind.shape[0]
    g = torch.Generator().manual_seed(c)
    x = torch.randn(n, c, generator=g).to(torch.bfloat16).to(cuda)
    for k in (13, 4, 26):
        w = torch.zeros(c, 27, c, dtype=torch.bfloat16)
        w[:, k, :] = torch.eye(c, dtype=torch.bfloat16)
        out = ops.spconv_fwd(x, w.to(cuda), None, nbr)
        idx = nbr[k].long()
        ref = torch.where((idx >= 0)[:, None], x[idx.clamp(min=0)], torch.zeros_like(x[:1]))
        assert torch.equal(out, ref), f"offset {k}"


def test_full_size_wgrad_counts_pairs(cuda, full_scene):
    """in = ones, dout = ones: dw[co][k][ci] = number of (output, input) pairs of offset k, exactly (integers < 2^24)."""
    from pointcept_amd import ops

    ind, table, nbr = full_scene
    n = ind.shape[0]
    ones = torch.ones(n, 64, dtype=torch.bfloat16, device=cuda)
    dw = ops.spconv_wgrad(ones, ones, nbr)
    cnt = (nbr >= 0).sum(1).float()
    assert dw.shape == (64, 27, 64)
    assert torch.equal(dw, cnt[None, :, None].expand(64, 27, 64))


def test_full_size_attention_properties(cuda):
    """800 sequences x 1024 x 4 heads: (a) V constant per sequence-head -> output equals that constant (softmax rows sum
    to one), (b) lse is invariant... shifts by log of the key count for zero queries, (c) backward of a constant V field
    gives dQ = dK = 0 (the gradient through a row-stochastic matrix applied to a constant vanishes)."""
    from pointcept_amd import ops

    n_seq, L, H = 800, 1024, 4
    T = n_seq * L
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(T, 3, H, 16, generator=g).to(torch.bfloat16)
    const = torch.randn(n_seq, 1, H, 16, generator=g).to(torch.bfloat16)
    qkv[:, 2] = const.expand(n_seq, L, H, 16).reshape(T, H, 16)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32)
    out, lse = ops.attn_varlen_fwd(qkv.to(cuda), cu.to(cuda), L, 0.25)
    ref = const.expand(n_seq, L, H, 16).reshape(T, H, 16).to(cuda)
    _close("attn_const_v", out, ref.float(), 2.0 ** -7, 1e-3)
    zq = qkv.clone()
    zq[:, 0] = 0
    _, lse0 = ops.attn_varlen_fwd(zq.to(cuda), cu.to(cuda), L, 0.25)
    _close("attn_lse_zero_q", lse0, torch.full_like(lse0, float(np.log(L))), 1e-4, 1e-3)
    dout = torch.randn(T, H, 16, generator=g).to(torch.bfloat16).to(cuda)
    dqkv = ops.attn_varlen_bwd(qkv.to(cuda), out, dout, lse, cu.to(cuda), L, 0.25)
    scale = float(dout.float().abs().max())
    assert float(dqkv[:, 0].float().abs().max()) <= 3e-2 * scale   # bf16 P / dP rounding noise only
    assert float(dqkv[:, 1].float().abs().max()) <= 3e-2 * scale


def test_full_size_attention_backward_forms_and_linearity(cuda, monkeypatch):
    """800 sequences x 1024 x 4 heads, the launch the bench step makes most: (a) the one-pass backward (attention_bwd1.h, the default at
    this size) is bit-reproducible and its dK / dV are bit-identical to the two split kernels'; dQ -- the same products, summed as eight
    per-wave partials -- agrees with theirs to the rounding of its bf16 output; (b) exact linearity in dO under a power-of-two factor
    (every product of the backward is linear in dO or in delta = rowsum(dO . O); scaling by 4 is exact in bf16 and fp32); (c) sequences are
    independent: reversing the order of the sequences reverses the gradient rows and changes nothing else."""
    from pointcept_amd import ops

    n_seq, L, H = 800, 1024, 4
    T = n_seq * L
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(T, 3, H, 16, generator=g).to(torch.bfloat16).to(cuda)
    dout = torch.randn(T, H, 16, generator=g).to(torch.bfloat16).to(cuda)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32).to(cuda)
    out, lse = ops.attn_varlen_fwd(qkv, cu, L, 0.25)
    monkeypatch.delenv("PTC_AT_BWD1", raising=False)
    d1 = ops.attn_varlen_bwd(qkv, out, dout, lse, cu, L, 0.25)                 # default at 3200 units: the one-pass kernel
    assert torch.equal(d1, ops.attn_varlen_bwd(qkv, out, dout, lse, cu, L, 0.25)), "not bit-reproducible"
    monkeypatch.setenv("PTC_AT_BWD1", "1")
    assert torch.equal(d1, ops.attn_varlen_bwd(qkv, out, dout, lse, cu, L, 0.25)), "the default at this size is not the one-pass kernel"
    monkeypatch.setenv("PTC_AT_BWD1", "0")
    d0 = ops.attn_varlen_bwd(qkv, out, dout, lse, cu, L, 0.25)
    monkeypatch.delenv("PTC_AT_BWD1", raising=False)
    assert torch.equal(d1[:, 1:], d0[:, 1:]), "dK / dV differ between the two forms"
    dq1, dq0 = d1[:, 0].float(), d0[:, 0].float()
    assert float((dq1 - dq0).abs().max()) <= 2.0 ** -7 * float(dq0.abs().max())
    assert float((dq1 - dq0).norm() / dq0.norm()) < 2.0 ** -9
    d4 = ops.attn_varlen_bwd(qkv, out, dout * 4, lse, cu, L, 0.25)
    assert torch.equal(d4.float(), d1.float() * 4), "the backward is not exactly linear in dO"
    rev = torch.arange(n_seq - 1, -1, -1, device=cuda)
    perm = (rev[:, None] * L + torch.arange(L, device=cuda)[None, :]).reshape(-1)
    dr = ops.attn_varlen_bwd(qkv[perm].contiguous(), out[perm].contiguous(), dout[perm].contiguous(), lse[:, perm].contiguous(), cu, L, 0.25)
    assert torch.equal(dr, d1[perm])


@pytest.mark.parametrize("n", [1, 5, 17, 33, 129])
def test_conv_tiny_inputs(cuda, n):
    """row counts below one tile / one workgroup, through the chunked-pipeline kernel and the weight gradient"""
    from pointcept_amd import ops

    g = torch.Generator().manual_seed(n)
    coords = torch.stack([torch.zeros(n, dtype=torch.int64), torch.arange(n) % 4, (torch.arange(n) // 4) % 4, torch.arange(n) // 16], 1)
    ind = coords.numpy().astype(np.int32)
    nbr = oops.subm_rulebook(ind, 3)
    got_nbr = ops.rulebook_subm(_t(ind, cuda), 3)
    assert np.array_equal(got_nbr.cpu().numpy(), nbr)
    feat = torch.randn(n, 64, generator=g).to(torch.bfloat16)
    w = (torch.randn(64, 27, 64, generator=g) * 0.05).to(torch.bfloat16)
    ref = oops.gather_conv(feat.float(), w.float(), None, nbr)
    got = ops.spconv_fwd(feat.to(cuda), w.to(cuda), None, got_nbr)
    _close("conv_tiny", got, ref, 1.0 / 128, 2e-3)
    dout = torch.randn(n, 64, generator=g).to(torch.bfloat16)
    wr = w.float().requires_grad_(True)
    oops.gather_conv(feat.float(), wr, None, nbr).backward(dout.float())
    dw = ops.spconv_wgrad(feat.to(cuda), dout.to(cuda), got_nbr)
    _close("wgrad_tiny", dw, wr.grad, 1e-4, 1e-3 * float(wr.grad.abs().max()) + 1e-6)


@pytest.mark.parametrize("counts,K", [([10, 3, 7], 4), ([1024, 1025, 5000, 1], 1024), ([48, 49, 100, 7], 48), ([2048], 1024)])
def test_attn_tables_match_index_algebra(cuda, counts, K):
    """ptc_attn_tables = the index algebra of SerializedAttention.forward (ptv3m1:184-188,216) and its backward"""
    from pointcept_amd import ops

    n = sum(counts)
    off = torch.tensor(np.cumsum(counts), dtype=torch.int64)
    g = torch.Generator().manual_seed(n + K)
    # a per-scene permutation as the serialization order
    order = torch.cat([torch.randperm(c, generator=g) + s for c, s in zip(counts, [0] + list(np.cumsum(counts)[:-1]))])
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(n)
    pad, unpad, cu, dup = ops.patch_pad_maps(off.to(cuda), off.tolist(), K)
    t1, t2, t3, t4 = ops.attn_tables(order.to(cuda), inverse.to(cuda), pad, unpad, dup)
    gidx = order.to(cuda)[pad]
    inv = unpad[inverse.to(cuda)]
    dup_of_point = dup[inverse.to(cuda)]
    slots = torch.arange(gidx.numel(), device=cuda)
    gidx_primary = torch.where(inv[gidx] == slots, gidx, torch.full_like(gidx, -1))
    assert torch.equal(t1[0].long(), gidx)
    assert torch.equal(t2[0].long(), inv) and torch.equal(t2[1].long(), dup_of_point)
    assert torch.equal(t3[0].long(), inv)
    assert torch.equal(t4[0].long(), gidx_primary)


# ---- Lovasz-Softmax (the second criterion of the ScanNet config) ------------------------------
def test_lovasz_softmax_matches_reference_golden_and_oracle(cuda):
    """fp32 logits: loss vs the REFERENCE module's value (tests/golden/lovasz.npz), gradient vs the fp64 oracle
    (the reference's own fp32 gradient carries the cancellation noise of lovasz.py:31-32, so it is compared looser)."""
    from oracle import losses
    from pointcept_amd import functional as PF
    from test_golden_cpu import lovasz_cases

    for ci, x, y, loss_ref, grad_ref in lovasz_cases():
        xe = x.clone().to(cuda).requires_grad_(True)
        loss = PF.lovasz_softmax(xe, y.to(cuda), -1)
        (loss * 2.5).backward()
        lo, do = losses.lovasz_softmax(x.numpy(), y.numpy(), -1)
        assert abs(loss.item() - loss_ref) <= 1e-4 * max(abs(loss_ref), 1e-3), (ci, loss.item(), loss_ref)
        assert abs(loss.item() - lo) <= 2e-6 * max(abs(lo), 1e-3), (ci, loss.item(), lo)
        g = xe.grad.cpu().numpy() / 2.5
        gmax = max(np.abs(do).max(), 1e-12)
        assert np.abs(g - do).max() <= 1e-4 * gmax, (ci, np.abs(g - do).max(), gmax)
        assert np.abs(g - grad_ref).max() <= 1e-3 * gmax, ci


@pytest.mark.parametrize("n", [5000, 56000])        # 56000 x 20 slots: more than LV_STEP_BLOCKS workgroups of 256 (the grid-stride loop of lovasz_step)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_lovasz_softmax_16bit_strided_and_edge_cases(cuda, dtype, n):
    from oracle import losses
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(11)
    c = 20
    wide = (torch.randn(n, 32, generator=g) * 2).to(dtype)
    y = torch.randint(0, 17, (n,), generator=g)        # classes 17..19 absent
    y[torch.rand(n, generator=g) < 0.1] = -1
    xe = wide.clone().to(cuda).requires_grad_(True)
    loss = PF.lovasz_softmax(xe[:, :c], y.to(cuda), -1)   # strided view of a wider head output
    loss.backward()
    lo, do = losses.lovasz_softmax(wide[:, :c].float().numpy(), y.numpy(), -1)
    assert abs(loss.item() - lo) <= 1e-5 * abs(lo)
    got = xe.grad[:, :c].float().cpu().numpy()
    assert np.abs(got - do).max() <= 2e-2 * np.abs(do).max()      # gradient rounded to the 16-bit input dtype
    assert float(xe.grad[:, c:].abs().max()) == 0.0
    # nothing counted -> 0 loss, 0 gradient; empty input -> 0
    x2 = torch.randn(300, c, generator=g).to(dtype).to(cuda).requires_grad_(True)
    l2 = PF.lovasz_softmax(x2, torch.full((300,), -1, dtype=torch.int64, device=cuda), -1)
    l2.backward()
    assert l2.item() == 0.0 and float(x2.grad.abs().max()) == 0.0
    assert PF.lovasz_softmax(torch.zeros(0, c, dtype=dtype, device=cuda), torch.zeros(0, dtype=torch.int64, device=cuda), -1).item() == 0.0


def test_lovasz_softmax_full_size_properties(cuda):
    """BASELINE batch (819200 points x 20 classes): bit-reproducible, invariant under a permutation of the points
    (the loss is a symmetric function of the (error, label) pairs), bounded by [0, 1], zero for a perfect prediction."""
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(12)
    n, c = 819200, 20
    x = torch.randn(n, c, generator=g).to(torch.bfloat16).to(cuda)
    y = torch.randint(0, c, (n,), generator=g)
    y[torch.rand(n, generator=g) < 0.05] = -1
    y = y.to(cuda)
    xa = x.clone().requires_grad_(True)
    la = PF.lovasz_softmax(xa, y, -1)
    la.backward()
    xb = x.clone().requires_grad_(True)
    lb = PF.lovasz_softmax(xb, y, -1)
    lb.backward()
    assert torch.equal(la, lb) and torch.equal(xa.grad, xb.grad)
    assert 0.0 < la.item() <= 1.0 and torch.isfinite(xa.grad).all()
    perm = torch.randperm(n, generator=g).to(cuda)
    lp = PF.lovasz_softmax(x[perm], y[perm], -1)
    assert abs(lp.item() - la.item()) <= 1e-6
    perfect = torch.full((n, c), -30.0, device=cuda)
    perfect[torch.arange(n, device=cuda), y.clamp(min=0)] = 30.0
    assert PF.lovasz_softmax(perfect, y, -1).item() <= 1e-6


def test_segmentor_ce_plus_lovasz(cuda):
    """criteria = [CrossEntropyLoss, LovaszLoss] summed (losses/builder.py:22-31), the ScanNet PTv3 configuration."""
    from oracle import losses
    from pointcept_amd.segmentor import DefaultSegmentorV2

    class Feat(torch.nn.Module):
        def forward(self, point):
            return point["feat"]

    torch.manual_seed(0)
    seg = DefaultSegmentorV2(20, 64, Feat(), criteria=("ce", "lovasz")).to(cuda).train()
    g = torch.Generator().manual_seed(13)
    feat = torch.randn(3000, 64, generator=g)
    y = torch.randint(-1, 20, (3000,), generator=g)
    out = seg(dict(feat=feat.to(cuda), segment=y.to(cuda), offset=torch.tensor([3000], device=cuda)))
    out["loss"].backward()
    logits = feat @ seg.seg_head.weight.detach().cpu().t() + seg.seg_head.bias.detach().cpu()
    ce = torch.nn.functional.cross_entropy(logits, y, ignore_index=-1).item()
    lv, _ = losses.lovasz_softmax(logits.numpy(), y.numpy(), -1)
    assert abs(out["loss"].item() - (ce + lv)) <= 2e-3 * (ce + lv)
    assert seg.seg_head.weight.grad is not None and torch.isfinite(seg.seg_head.weight.grad).all()


@pytest.mark.parametrize("c", [48, 36, 24])
def test_square_subm_conv_of_an_odd_width_on_the_block_staged_kernels(cuda, c, n_pts=9000):
    """round 5: SubMConv3d(c, c, 3) with 16 < c < 64, c != 32 (PT-v3m2's 48 at the Sonata widths, LitePT's 36) under 16-bit autocast is
    zero-padded to 32 / 64 channels and runs on conv7 / wgrad7 (square 32 | 64-channel kernels) where it ran on conv2: forward, input,
    weight and bias gradients against the oracle's gather convolution in fp32 on the bf16-rounded operands."""
    from pointcept_amd import spconv_api as sp

    ind = _scene_indices(n_pts)
    n = ind.shape[0]
    assert n >= 4096                      # the block-staged kernels' minimum
    nbr = oops.subm_rulebook(ind, 3)
    g = torch.Generator().manual_seed(c)
    conv = sp.SubMConv3d(c, c, 3, bias=True, indice_key="k").to(cuda)
    feat = (torch.randn(n, c, generator=g) * 0.5)
    probe = torch.randn(n, c, generator=g)
    fe = feat.to(cuda).requires_grad_(True)
    x = sp.SparseConvTensor(fe, torch.from_numpy(ind).int().to(cuda), [int(ind[:, 1:].max()) + 97] * 3, int(ind[:, 0].max()) + 1)
    with torch.autocast("cuda" if torch.device(cuda).type == "cuda" else "cpu", dtype=torch.bfloat16, enabled=torch.device(cuda).type == "cuda"):
        y = conv(x).features
    assert y.shape == (n, c)
    (y.float() * probe.to(cuda)).sum().backward()
    rnd = (lambda t: t.to(torch.bfloat16).float()) if torch.device(cuda).type == "cuda" else (lambda t: t)
    fr = rnd(feat).requires_grad_(True)
    wr = rnd(conv.weight.detach().cpu().reshape(c, 27, c)).requires_grad_(True)
    br = conv.bias.detach().cpu().clone().requires_grad_(True)
    ref = oops.gather_conv(fr, wr, br, nbr)
    (ref * rnd(probe)).sum().backward()
    rtol, atol = _tols(torch.bfloat16)
    _close("odd_conv_fwd", y, ref, rtol, atol * 4)
    _close("odd_conv_dx", fe.grad, fr.grad, rtol, 4 * atol * float(fr.grad.abs().max()))
    _close("odd_conv_dw", conv.weight.grad.reshape(c, 27, c), wr.grad, 2e-2, 1e-2 * float(wr.grad.abs().max()))
    _close("odd_conv_db", conv.bias.grad, br.grad, 2e-2, 1e-2 * float(br.grad.abs().max()))


def test_spconv_autograd_with_duplicate_voxels(cuda):
    """Mix3D batches list some voxels twice (or more).  Forward: the lowest row wins every lookup; the backward must be
    the exact adjoint of THAT map (copies that nothing reads get zero gradient, the gradients of identical output rows
    add up in the representative).  SubM k=3, strided k=2 s=2 and its inverse, chained, vs the oracle's plain autograd."""
    from oracle import shims
    from pointcept_amd import spconv_api as sp

    g = torch.Generator().manual_seed(21)
    base = torch.unique(torch.randint(0, 12, (400, 3), generator=g), dim=0)
    extra = base[torch.randperm(base.shape[0], generator=g)[:60]]
    coords = torch.cat([base, extra, extra[:15]])                       # 60 voxels twice, 15 of them three times
    coords = coords[torch.randperm(coords.shape[0], generator=g)]
    n = coords.shape[0]
    batch = (torch.arange(n) >= n // 2).int()
    indices = torch.cat([batch[:, None], coords.int()], dim=1).contiguous()
    feat = torch.randn(n, 16, generator=g)
    probe = torch.randn(n, 16, generator=g)

    def build(mod):
        torch.manual_seed(3)
        return [mod.SubMConv3d(16, 32, 3, bias=True, indice_key="a"), mod.SparseConv3d(32, 32, 2, stride=2, bias=False, indice_key="d"),
                mod.SubMConv3d(32, 32, 3, bias=False, indice_key="b"), mod.SparseInverseConv3d(32, 16, 2, bias=False, indice_key="d"),
                mod.SubMConv3d(16, 16, 3, bias=False, indice_key="a")]

    ref_layers, eng_layers = build(shims), build(sp)
    for a, b in zip(ref_layers, eng_layers):
        b.load_state_dict(a.state_dict())
        b.to(cuda)
    fo = feat.clone().requires_grad_(True)
    x = shims.SparseConvTensor(fo, indices, [108, 108, 108], 2)
    for layer in ref_layers:
        x = layer(x)
    (x.features * probe).sum().backward()
    fe = feat.to(cuda).requires_grad_(True)
    y = sp.SparseConvTensor(fe, indices.to(cuda), [108, 108, 108], 2)
    for layer in eng_layers:
        y = layer(y)
    (y.features * probe.to(cuda)).sum().backward()
    assert torch.allclose(y.features.detach().cpu(), x.features.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(fe.grad.cpu(), fo.grad, rtol=1e-3, atol=1e-4)
    assert float((fo.grad.abs().sum(1) == 0).float().mean()) > 0.05            # the unread copies
    for a, b in zip(ref_layers, eng_layers):
        assert torch.allclose(b.weight.grad.cpu(), a.weight.grad, rtol=1e-3, atol=1e-3), a.indice_key
    assert torch.allclose(eng_layers[0].bias.grad.cpu(), ref_layers[0].bias.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("row", [0, 2])
def test_pool_level_counts(cuda, row):
    """sizes of every pooled level from the stage-0 codes: equal to counting unique (code >> shift) per scene, for the
    Morton row and a Hilbert row alike (both curves are hierarchical), with duplicate voxels present."""
    from pointcept_amd import ops, synthetic

    a, b = synthetic.indoor_scene(3, 5000), synthetic.indoor_scene(4, 800)
    a = {k: np.concatenate([v, v[:100]]) for k, v in a.items()}          # 100 duplicate voxels
    batch = synthetic.collate([a, b, synthetic.indoor_scene(5, 1)])
    gc, off = batch["grid_coord"], batch["offset"]
    depth = int(gc.max() + 1).bit_length()
    bidx = omaps.offset2batch(off)
    code = osfc.encode_c(gc, bidx, depth, ("z", "z-trans", "hilbert", "hilbert-trans"))
    order = np.argsort(code, axis=1, kind="stable")
    shifts = [3, 6, 9, 15, 3 * depth]
    got = ops.pool_level_counts(torch.from_numpy(code[row]).to(cuda), torch.from_numpy(order[row]).to(cuda), 3 * depth, len(off),
                                shifts).cpu().numpy()
    for l, sh in enumerate(shifts):
        want = [len(np.unique(code[row][bidx == s] >> sh)) for s in range(len(off))]
        assert got[l].tolist() == want, (l, sh)
    z = [len(np.unique(code[0][bidx == s] >> 3)) for s in range(len(off))]
    h = [len(np.unique(code[2][bidx == s] >> 3)) for s in range(len(off))]
    assert z == h


# ---- device-side GridSample -------------------------------------------------------------------
def test_gridsample_matches_reference_golden_and_oracle(cuda):
    """voxel of every point, FNV keys, key order, voxel ids, counts: bit-exact vs the oracle; `inverse`, the voxel list
    in np.unique order and min_coord: equal to what the REFERENCE transform produced (tests/golden/gridsample.npz);
    with injected random offsets the picked representatives equal the oracle's (stable tie order)."""
    from oracle import voxelize
    from pointcept_amd.transform import GridSample
    from test_golden_cpu import gridsample_cases

    for ci, coord, grid, g in gridsample_cases():
        v = voxelize.voxels(coord, grid)
        gs = GridSample(grid_size=grid, mode="train", return_grid_coord=True, return_inverse=True, return_min_coord=True,
                        return_displacement=True)
        dev = gs.voxels(torch.from_numpy(coord).to(cuda))
        assert np.array_equal(dev["grid_coord"].cpu().numpy(), v["grid_coord"]), ci
        assert np.array_equal(dev["key"].cpu().numpy().view(np.uint64), v["key"]), ci
        assert np.array_equal(dev["idx_sort"].cpu().numpy(), v["idx_sort"]), ci
        assert np.array_equal(dev["inverse"].cpu().numpy(), g["inverse"]), ci
        assert np.array_equal(np.diff(dev["idx_ptr"].cpu().numpy()), v["count"]), ci
        rand = np.random.default_rng(ci).integers(0, 1 << 30, len(v["count"]))
        n = coord.shape[0]
        out = gs(dict(coord=torch.from_numpy(coord).to(cuda), segment=torch.arange(n, device=cuda),
                      index_valid_keys=["coord", "segment"]), rand=torch.from_numpy(rand).to(cuda))
        pick = voxelize.select_train(v, rand)
        assert np.array_equal(out["segment"].cpu().numpy(), pick), ci
        assert np.array_equal(out["grid_coord"].cpu().numpy(), g["voxels_keyorder"]), ci
        assert np.array_equal(out["coord"].cpu().numpy(), coord[pick]), ci
        assert np.allclose(out["min_coord"].cpu().numpy(), g["min_coord"]), ci
        assert np.array_equal(out["inverse"].cpu().numpy(), g["inverse"]), ci
        disp = out["displacement"].cpu().numpy()
        assert disp.shape == (len(pick), 3) and (disp >= -0.5).all() and (disp <= 0.5).all()
        assert out["index_valid_keys"] == ["coord", "segment", "grid_coord", "displacement"]


def test_gridsample_test_mode_and_full_size_properties(cuda):
    from pointcept_amd.transform import GridSample

    g = torch.Generator().manual_seed(31)
    coord = ((torch.rand(2_000_000, 3, generator=g) - 0.5) * torch.tensor([7.0, 5.0, 2.8])).to(cuda)
    gs = GridSample(grid_size=0.02, mode="train", return_grid_coord=True, return_inverse=True)
    out = gs(dict(coord=coord, index_valid_keys=["coord"]))
    gc, inv = out["grid_coord"], out["inverse"]
    v = gc.shape[0]
    assert torch.unique(gc, dim=0).shape[0] == v                       # one representative per voxel
    assert int(inv.max()) == v - 1 and int(inv.min()) == 0
    assert torch.equal(torch.floor(out["coord"].double() / 0.02).long() - torch.floor(coord.double() / 0.02).long().min(0).values, gc)
    small = coord[:5000] * 0.05
    parts = GridSample(grid_size=0.02, mode="test", return_grid_coord=True)(dict(coord=small, index_valid_keys=["coord"]))
    seen = torch.zeros(5000, dtype=torch.bool, device=cuda)
    for p in parts:
        assert torch.unique(p["grid_coord"], dim=0).shape[0] == p["grid_coord"].shape[0]
        seen[p["index"]] = True
    assert bool(seen.all())                                             # every point appears in some part (transform.py:916-949)


# ---- pointops subset ---------------------------------------------------------------------------
@pytest.mark.parametrize("nsample", [1, 3, 16, 20, 70])
def test_pointops_knn_query(cuda, nsample):
    """three scenes (one with fewer points than nsample, one whose queries straddle a workgroup boundary), duplicated
    points (exact distance ties): idx and dist bit-exact vs the oracle (ties: lower index first)."""
    from oracle import pointops as opo
    from pointcept_amd import pointops_api as po

    rng = np.random.default_rng(41)
    sizes, qsizes = [2500, 9, 1300], [700, 40, 300]
    xyz = rng.random((sum(sizes), 3)).astype(np.float32)
    xyz[100:150] = xyz[50:100]                                        # ties
    new_xyz = rng.random((sum(qsizes), 3)).astype(np.float32)
    new_xyz[:20] = xyz[:20]                                           # zero distances
    off, noff = np.cumsum(sizes), np.cumsum(qsizes)
    want_i, want_d = opo.knn_query(nsample, xyz, off, new_xyz, noff)
    got_i, got_d = po.knn_query(nsample, torch.from_numpy(xyz).to(cuda), torch.from_numpy(off).to(cuda),
                                torch.from_numpy(new_xyz).to(cuda), torch.from_numpy(noff).to(cuda))
    assert got_i.dtype == torch.int32 and got_i.shape == (sum(qsizes), nsample)
    assert np.array_equal(got_i.cpu().numpy(), want_i)
    assert np.array_equal(got_d.cpu().numpy(), want_d)
    self_i, self_d = po.knn_query(min(nsample, 8), torch.from_numpy(xyz).to(cuda), torch.from_numpy(off).to(cuda))
    assert float(self_d[:, 0].max()) == 0.0                           # every point is its own (or its twin's) nearest neighbour


def test_pointops_fps_grouping_interpolation(cuda):
    from oracle import pointops as opo
    from pointcept_amd import pointops_api as po

    rng = np.random.default_rng(42)
    sizes, picks = [3000, 1, 777], [200, 1, 64]
    xyz = rng.random((sum(sizes), 3)).astype(np.float32)
    off, noff = np.cumsum(sizes), np.cumsum(picks)
    x, o, no = torch.from_numpy(xyz).to(cuda), torch.from_numpy(off).to(cuda), torch.from_numpy(noff).to(cuda)
    got = po.farthest_point_sampling(x, o, no)
    assert np.array_equal(got.cpu().numpy(), opo.farthest_point_sampling(xyz, off, noff))
    # grouping (-1 slots -> zeros, relative coordinates) and its gradient
    feat = torch.randn(sum(sizes), 8, device=cuda, requires_grad=True)
    centres = x[got.long()]
    idx, _ = po.knn_query(4, x, o, centres, no)
    grouped = po.grouping(idx, feat, x, centres, with_xyz=True)
    assert grouped.shape == (noff[-1], 4, 11)
    slot = idx[0, 1].long()
    assert torch.equal(grouped[0, 1, 3:], feat[slot].detach()) and torch.allclose(grouped[0, 1, :3], x[slot] - centres[0])
    assert int((idx[picks[0]] == -1).sum()) == 3 and float(grouped[picks[0], 1:].abs().max()) == 0.0   # the 1-point scene
    grouped.sum().backward()
    counts = torch.bincount(idx[idx >= 0].long(), minlength=sum(sizes)).float()
    assert torch.allclose(feat.grad[:, 0], counts)
    # interpolation back to all points: exact at the sampled points, convex combination elsewhere
    vals = torch.randn(noff[-1], 5, device=cuda)
    up = po.interpolation(centres, x, vals, no, o, k=3)
    assert up.shape == (sum(sizes), 5)
    assert torch.allclose(up[got.long()], vals, atol=1e-4)
    assert float(up.abs().max()) <= float(vals.abs().max()) + 1e-4


@pytest.mark.parametrize("c,w_c", [(8, 4), (3, 1), (32, 8), (6, 2), (64, 64)])
def test_pointops_edge_operators(cuda, c, w_c):
    """grouping / interpolation / aggregation / subtraction on their kernels (csrc/pointops_edges.hip) against the restatements of the
    reference's CUDA kernels (oracle/pointops_c.py: C++ signatures, in-place outputs), forward and every gradient; -1 slots, repeated
    sources, channel counts that are not multiples of 4, gradients bit-reproducible (segmented sums, no atomics)."""
    from oracle import pointops_c as R
    from pointcept_amd import pointops_api as po

    g = torch.Generator().manual_seed(100 * c + w_c)
    n, m, ns = 700, 333, 7
    idx = torch.randint(0, n, (m, ns), generator=g).int()
    idx[::5, 3] = idx[::5, 2]                                   # repeated sources inside a row
    holes = idx.clone()
    holes[torch.rand(m, ns, generator=g) < 0.15] = -1           # empty neighbour slots
    feat, xyz, new_xyz = torch.randn(n, c, generator=g), torch.rand(n, 3, generator=g), torch.rand(m, 3, generator=g)

    def run(fn, *tensors):
        leaves = [t.clone().to(cuda).requires_grad_(True) for t in tensors]
        out = fn(*leaves)
        probe = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(cuda)
        (out * probe).sum().backward()
        return out.detach().cpu(), [t.grad.cpu() for t in leaves], probe.cpu()

    def close(name, a, b, tol=1e-5):
        assert a.shape == b.shape and a.dtype == b.dtype, (name, a.shape, b.shape, a.dtype, b.dtype)
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), name

    # grouping with holes and relative coordinates (functions/grouping.py:44-68 on grouping_forward_cuda)
    out, (df, dx, dn), probe = run(lambda f, x, q: po.grouping(holes.to(cuda), f, x, q, with_xyz=True), feat, xyz, new_xyz)
    mask = (holes >= 0).float()[:, :, None]
    safe = holes.clamp(min=0)
    want = torch.zeros(m, ns, c)
    R.grouping_forward_cuda(m, ns, c, feat, safe, want)
    want_xyz = torch.zeros(m, ns, 3)
    R.grouping_forward_cuda(m, ns, 3, xyz, safe, want_xyz)
    want = torch.cat([(want_xyz - new_xyz[:, None, :]) * mask, want * mask], -1)
    close("grouping", out, want, 0.0)
    gf, gx = torch.zeros(n, c), torch.zeros(n, 3)
    R.grouping_backward_cuda(m, ns, c, (probe[:, :, 3:] * mask).contiguous(), safe, gf)
    R.grouping_backward_cuda(m, ns, 3, (probe[:, :, :3] * mask).contiguous(), safe, gx)
    close("grouping d_feat", df, gf)
    close("grouping d_xyz", dx, gx)
    close("grouping d_new_xyz", dn, -(probe[:, :, :3] * mask).sum(1))
    out2, (df2,), _ = run(lambda f: po.grouping2(f, idx.to(cuda)), feat)
    want = torch.zeros(m, ns, c)
    R.grouping_forward_cuda(m, ns, c, feat, idx, want)
    close("grouping2", out2, want, 0.0)
    again = run(lambda f: po.grouping2(f, idx.to(cuda)), feat)[1][0]
    assert torch.equal(df2, again), "segmented-sum gradients must be bit-reproducible"

    # interpolation (interpolation_forward / backward_cuda) through its Function with explicit weights
    wgt = torch.rand(m, ns, generator=g)
    out, (df,), probe = run(lambda f: po._EdgeInterpolate.apply(f, idx.to(cuda), wgt.to(cuda)), feat)
    want, gf = torch.zeros(m, c), torch.zeros(n, c)
    R.interpolation_forward_cuda(m, c, ns, feat, idx, wgt, want)
    R.interpolation_backward_cuda(m, c, ns, probe, idx, wgt, gf)
    close("interpolation", out, want)
    close("interpolation d_feat", df, gf)

    # subtraction
    a = torch.randn(m, c, generator=g)
    out, (d1, d2), probe = run(lambda x, y: po.subtraction(x, y, idx.to(cuda)), a, feat)
    want, g1, g2 = torch.zeros(m, ns, c), torch.zeros(m, c), torch.zeros(n, c)
    R.subtraction_forward_cuda(m, ns, c, a, feat, idx, want)
    R.subtraction_backward_cuda(m, ns, c, idx, probe, g1, g2)
    close("subtraction", out, want, 0.0)
    close("subtraction d1", d1, g1)
    close("subtraction d2", d2, g2)

    # aggregation (PTv1 vector attention): c channels share w_c weight columns
    pos, w = torch.randn(m, ns, c, generator=g), torch.randn(m, ns, w_c, generator=g)
    idx_a = idx % m                                           # aggregation gathers from its own [m, c] input
    out, (di, dp, dw), probe = run(lambda x, p, q: po.aggregation(x, p, q, idx_a.to(cuda)), a, pos, w)
    want, gi, gp, gw = torch.zeros(m, c), torch.zeros(m, c), torch.zeros(m, ns, c), torch.zeros(m, ns, w_c)
    R.aggregation_forward_cuda(m, ns, c, w_c, a, pos, w, idx_a, want)
    R.aggregation_backward_cuda(m, ns, c, w_c, a, pos, w, idx_a, probe, gi, gp, gw)
    close("aggregation", out, want)
    close("aggregation d_input", di, gi)
    close("aggregation d_position", dp, gp)
    close("aggregation d_weight", dw, gw)


def test_pointops_through_compat_and_unsupported(cuda):
    import sys

    import pointcept_amd.compat as compat
    from pointcept_amd._lib import PtcoreError

    saved = {k: sys.modules.get(k) for k in ("spconv", "spconv.pytorch", "spconv.pytorch.modules", "flash_attn", "torch_scatter", "pointops")}
    try:
        compat.install(force=True)
        import pointops

        x = torch.rand(100, 3, device=cuda)
        o = torch.tensor([60, 100], device=cuda)
        idx, dist = pointops.knn_query(2, x, o)
        assert idx.shape == (100, 2) and bool((idx[:60] < 60).all()) and bool((idx[60:] >= 60).all())
        assert torch.equal(pointops.offset2batch(o), torch.cat([torch.zeros(60), torch.ones(40)]).long().to(cuda))
        bi, bd = pointops.ball_query(4, 0.2, 0.0, x, o)
        assert bi.shape == (100, 4) and bool((bi[:, 0] == torch.arange(100, device=cuda)).all())     # nearest in-range point = itself
        with pytest.raises(PtcoreError):
            pointops.ball_query(4, 0.1, 0.2, x, o)                                                     # min_radius >= max_radius (query.py:93)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# ------------------------------------------------------------------------------------------------
# libs/pointrope
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_rope3d_matches_reference_golden_and_oracle(cuda, dtype):
    """ptc_rope3d (csrc/rope.hip) through the pointrope operator API against (a) the golden outputs of the reference's own
    pointrope_cpu and (b) the numpy oracle on the same inputs; in place, as the extension."""
    import os

    from oracle import pointrope as orope
    from pointcept_amd import pointrope_api

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pointrope.npz"))
    for ci in range(int(g["n_cases"])):
        tok, pos, ref = g[f"tokens_{ci}"], g[f"pos_{ci}"], g[f"out_{ci}"]
        base, fwd = (float(v) for v in g[f"params_{ci}"])
        t = _t(tok, cuda).to(dtype).contiguous()
        t_in = t.float().cpu().numpy()
        p = _t(pos, cuda)
        ret = pointrope_api.pointrope(t, p, base, fwd)
        assert ret is None                                              # in place
        got = t.float().cpu().numpy()
        tol = {torch.float32: 2e-4, torch.bfloat16: 1.0 / 128, torch.float16: 1.0 / 512}[dtype] * np.abs(ref).max()
        assert np.abs(got - orope.pointrope(t_in, pos, base, fwd)).max() <= tol, ci      # oracle on the SAME rounded inputs
        if dtype == torch.float32:
            assert np.abs(got - ref).max() <= tol, ci                   # reference golden


def test_rope3d_autograd_is_the_inverse_rotation(cuda):
    """PointROPE_func (litept_v1.py:27-46): backward = the same kernel with -F0; checked against autograd through the
    oracle's formula in torch, and as a round trip."""
    from pointcept_amd import pointrope_api

    g = torch.Generator().manual_seed(5)
    B, H, N, D = 2, 3, 77, 24
    x = torch.randn(B, H, N, D, generator=g).to(cuda).requires_grad_(True)
    pos = torch.randint(0, 200, (B, N, 3), generator=g).to(cuda)
    rope = pointrope_api.PointROPE(freq=100.0, F0=1.0)
    y = rope(x, pos)
    w = torch.randn(y.shape, generator=g).to(cuda)
    (y * w).sum().backward()
    # reference formula in torch (differentiable)
    xr = x.detach().clone().requires_grad_(True)
    Q = D // 6
    inv = 1.0 / (100.0 ** (torch.arange(Q, device=cuda, dtype=torch.float32) / Q))
    parts = []
    for a in range(3):
        f = pos[:, None, :, a, None].float() * inv
        u, v = xr[..., a * 2 * Q:a * 2 * Q + Q], xr[..., a * 2 * Q + Q:a * 2 * Q + 2 * Q]
        parts += [u * f.cos() - v * f.sin(), v * f.cos() + u * f.sin()]
    yr = torch.cat(parts, dim=-1)
    (yr * w).sum().backward()
    _close("rope_fwd", y, yr, 1e-4, 1e-4)
    _close("rope_bwd", x.grad, xr.grad, 1e-4, 1e-4)


# ------------------------------------------------------------------------------------------------
# front end: SphereCrop
# ------------------------------------------------------------------------------------------------
def test_sphere_crop_matches_reference_golden(cuda):
    """device SphereCrop (radix sort of the fp32 distance bits) against the reference transform's own output: same points
    in the same (ascending distance) order -- the random coordinates have no distance ties."""
    import os

    from pointcept_amd.transform import SphereCrop

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spherecrop.npz"))
    for ci in range(int(g["n_cases"])):
        d = dict(coord=_t(g[f"coord_{ci}"], cuda), segment=_t(g[f"segment_{ci}"], cuda))
        c = int(g[f"center_index_{ci}"])
        out = SphereCrop(point_max=int(g[f"point_max_{ci}"]), mode=str(g[f"mode_{ci}"]))(d, center_index=None if c < 0 else c)
        assert np.array_equal(out["coord"].cpu().numpy(), g[f"out_coord_{ci}"]), ci
        assert np.array_equal(out["segment"].cpu().numpy(), g[f"out_segment_{ci}"]), ci
    small = dict(coord=torch.rand(100, 3, device=cuda), segment=torch.zeros(100, dtype=torch.long, device=cuda))
    assert SphereCrop(point_max=200)(small)["coord"].shape[0] == 100          # fewer points than point_max: untouched (:1031)


# ------------------------------------------------------------------------------------------------
# evaluation tail: arg-max + inverse + class histograms
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_seg_eval_hist_matches_the_reference_formula(cuda, dtype):
    """ptc_seg_eval_hist against intersection_and_union (pointcept/utils/misc.py:37-54, the numpy form of the GPU function
    at :57-69) applied to pred = logits.max(1)[1][inverse] (evaluator.py:139-147): exact integer counts; ignored points,
    classes absent from the labels, a strided logits view, predictions given directly."""
    from pointcept_amd import ops

    rng = np.random.default_rng(3)
    n, m, k = 5000, 23000, 20
    wide = torch.from_numpy(rng.standard_normal((n, 32)).astype(np.float32)).to(dtype).to(cuda)
    logits = wide[:, :k]                                                        # row pitch 32, as the padded head output
    inverse = rng.integers(0, n, size=m)
    target = rng.integers(-1, k - 3, size=m)                                    # -1 = ignore; the last classes never occur
    pred = logits.float().cpu().numpy().argmax(1)[inverse]

    def ref(output, tgt):
        mask = tgt != -1
        o, t = output[mask], tgt[mask]
        inter = np.histogram(o[o == t], bins=np.arange(k + 1))[0]
        ao, at = np.histogram(o, bins=np.arange(k + 1))[0], np.histogram(t, bins=np.arange(k + 1))[0]
        return inter, ao + at - inter, at

    want = ref(pred, target)
    got = ops.seg_eval_hist(logits, _t(target, cuda), k, -1, inverse=_t(inverse, cuda))
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b)
    got2 = ops.seg_eval_hist(None, _t(target, cuda), k, -1, pred=_t(pred, cuda))
    for a, b in zip(got2, want):
        assert np.array_equal(a.cpu().numpy(), b)
    got3 = ops.seg_eval_hist(logits, _t(target[:n], cuda), k, -1)              # no inverse: evaluated points = rows
    for a, b in zip(got3, ref(logits.float().cpu().numpy().argmax(1), target[:n])):
        assert np.array_equal(a.cpu().numpy(), b)


# ------------------------------------------------------------------------------------------------
# libs/pointops remainder
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nsample,rmax,rmin", [(16, 0.12, 0.0), (8, 0.3, 0.05), (64, 0.08, 0.0)])
def test_ball_query_and_random_ball_query(cuda, nsample, rmax, rmin):
    """ptc_ball_query against the numpy restatement of the CUDA kernels' loops (parity unpinned: CUDA-only in the
    reference): two scenes, fewer / more candidates than nsample, the uniform sub-sampling ranks in fp32, a caller-supplied
    permutation for the random variant.  Distances are computed without FMA contraction on both sides: bit-equal."""
    from oracle import pointops as opo
    from pointcept_amd import pointops_api as po

    rng = np.random.default_rng(nsample)
    xyz = rng.random((1500, 3)).astype(np.float32)
    offset = np.array([900, 1500], dtype=np.int32)
    new_xyz = np.concatenate([xyz[:900:7], xyz[900::5]]).astype(np.float32)
    new_offset = np.array([len(xyz[:900:7]), len(new_xyz)], dtype=np.int32)
    idx, dist = po.ball_query(nsample, rmax, rmin, _t(xyz, cuda), _t(offset, cuda), _t(new_xyz, cuda), _t(new_offset, cuda))
    ri, rd = opo.ball_query(nsample, rmax, rmin, xyz, offset, new_xyz, new_offset)
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(dist.cpu().numpy(), rd)
    order = np.concatenate([rng.permutation(900), 900 + rng.permutation(600)]).astype(np.int32)
    idx, dist = po.random_ball_query(nsample, rmax, rmin, _t(xyz, cuda), _t(offset, cuda), _t(new_xyz, cuda), _t(new_offset, cuda),
                                     order=_t(order, cuda))
    ri, rd = opo.ball_query(nsample, rmax, rmin, xyz, offset, new_xyz, new_offset, order=order)
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(dist.cpu().numpy(), rd)
    idx, _ = po.random_ball_query(nsample, rmax, rmin, _t(xyz, cuda), _t(offset, cuda))       # self query, fresh permutation
    got = idx.cpu().numpy()
    assert ((got >= -1) & (got < 1500)).all() and (got[:900][got[:900] >= 0] < 900).all()      # neighbours stay inside the scene


def test_ptv1_ptv2_operators_match_their_definitions(cuda):
    """subtraction / aggregation / attention_relation_step / attention_fusion_step (libs/pointops/src/{subtraction,
    aggregation,attention}) against the kernels' loops written out in numpy; the fusion step is a segmented sum here."""
    from pointcept_amd import pointops_api as po

    rng = np.random.default_rng(1)
    n, ns, c, wc, g, m = 300, 8, 12, 4, 3, 2000
    f1, f2 = rng.standard_normal((n, c)).astype(np.float32), rng.standard_normal((n, c)).astype(np.float32)
    idx = rng.integers(0, n, size=(n, ns)).astype(np.int32)
    sub = po.subtraction(_t(f1, cuda), _t(f2, cuda), _t(idx, cuda)).cpu().numpy()
    assert np.allclose(sub, f1[:, None, :] - f2[idx], atol=1e-6)
    pos, w = rng.standard_normal((n, ns, c)).astype(np.float32), rng.standard_normal((n, ns, wc)).astype(np.float32)
    agg = po.aggregation(_t(f1, cuda), _t(pos, cuda), _t(w, cuda), _t(idx, cuda)).cpu().numpy()
    ref = ((f1[idx] + pos) * w[:, :, np.arange(c) % wc]).sum(1)
    assert np.allclose(agg, ref, atol=1e-4)
    q, k = rng.standard_normal((n, g, c)).astype(np.float32), rng.standard_normal((n, g, c)).astype(np.float32)
    wt = rng.standard_normal(c).astype(np.float32)
    it, ir = rng.integers(0, n, size=m).astype(np.int32), rng.integers(0, n, size=m).astype(np.int32)
    rel = po.attention_relation_step(_t(q, cuda), _t(k, cuda), _t(wt, cuda), _t(it, cuda), _t(ir, cuda)).cpu().numpy()
    assert np.allclose(rel, (q[it] * k[ir] * wt).sum(-1), atol=1e-4)
    aw = rng.standard_normal((m, g)).astype(np.float32)
    fus = po.attention_fusion_step(_t(aw, cuda), _t(k, cuda), _t(it, cuda), _t(ir, cuda)).cpu().numpy()
    ref = np.zeros((n, g, c), np.float64)
    np.add.at(ref, it, (aw[:, :, None] * k[ir]).astype(np.float64))
    assert np.allclose(fus, ref, atol=1e-4)


@pytest.mark.parametrize("c", [12, 16, 5])
def test_pair_list_attention_steps_and_their_gradients(cuda, c):
    """attention_relation_step / attention_fusion_step of libs/pointops (functions/attention.py:11-120) on ptc_pair_dot_weighted /
    ptc_pair_segment_sum: forward and ALL five gradients against the oracle's restatement of the reference's CUDA loops
    (oracle/pointops_c.py: attention_*_{forward,backward}_cuda), including rows no pair touches, repeated pairs, channel counts that are
    and are not multiples of 4; two runs are bit-identical (the reference's atomicAdd sums are not)."""
    from oracle import pointops_c as oc
    from pointcept_amd import pointops_api as po

    g = torch.Generator().manual_seed(c)
    n, G, m = 400, 3, 5000
    q, k = torch.randn(n, G, c, generator=g), torch.randn(n, G, c, generator=g)
    wt = torch.randn(c, generator=g)
    it = torch.randint(0, n - 40, (m,), generator=g).int()          # rows n-40.. receive nothing
    ir = torch.randint(0, n, (m,), generator=g).int()
    it[:50], ir[:50] = it[50:100], ir[50:100]                       # repeated pairs
    go = torch.randn(m, G, generator=g)
    # relation step
    ref = torch.zeros(m, G)
    oc.attention_relation_step_forward_cuda(m, G, c, q, k, wt, it, ir, ref)
    gq, gk, gw = torch.zeros(n, G, c), torch.zeros(n, G, c), torch.zeros(c)
    oc.attention_relation_step_backward_cuda(m, G, c, q, gq, k, gk, wt, gw, it, ir, go)
    outs = []
    for _ in range(2):
        qd, kd, wd = (t.clone().to(cuda).requires_grad_(True) for t in (q, k, wt))
        rel = po.attention_relation_step(qd, kd, wd, it.to(cuda), ir.to(cuda))
        rel.backward(go.to(cuda))
        outs.append((rel.detach(), qd.grad, kd.grad, wd.grad))
    assert rel.shape == (m, G) and rel.dtype == torch.float32
    for name, a, b in zip(("relation", "d query", "d key", "d weight"), outs[0], (ref, gq, gk, gw)):
        assert torch.allclose(a.cpu(), b, rtol=1e-4, atol=1e-4 * float(b.abs().max())), name
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "bit-reproducible"
    # fusion step
    aw = torch.randn(m, G, generator=g)
    ref = torch.zeros(n, G, c)
    oc.attention_fusion_step_forward_cuda(m, G, c, aw, k, it, ir, ref)
    gout = torch.randn(n, G, c, generator=g)
    gaw, gv = torch.zeros(m, G), torch.zeros(n, G, c)
    oc.attention_fusion_step_backward_cuda(m, G, c, aw, gaw, k, gv, it, ir, gout)
    outs = []
    for _ in range(2):
        awd, vd = aw.clone().to(cuda).requires_grad_(True), k.clone().to(cuda).requires_grad_(True)
        fus = po.attention_fusion_step(awd, vd, it.to(cuda), ir.to(cuda))
        fus.backward(gout.to(cuda))
        outs.append((fus.detach(), awd.grad, vd.grad))
    assert fus.shape == (n, G, c) and float(fus[n - 40:].abs().max()) == 0.0
    for name, a, b in zip(("fusion", "d weight", "d value"), outs[0], (ref, gaw, gv)):
        assert torch.allclose(a.cpu(), b, rtol=1e-4, atol=1e-4 * float(b.abs().max())), name
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "bit-reproducible"


def test_weight_layout_cache_one_launch_refresh(cuda):
    """functional._CastCache.layout: the [c_in][taps][c_out] layouts the input-gradient GEMMs read (mirrored taps for a
    submanifold convolution, transpose for nn.Linear, the repeated matrix of the kv = 2 gather-fused Linear) equal the torch
    permutations they replace, follow the weights through optimizer-style in-place updates, and die with the parameter."""
    from pointcept_amd import functional as PF

    cache = PF._CastCache()
    g = torch.Generator().manual_seed(4)
    conv = torch.nn.Parameter(torch.randn(64, 27, 32, generator=g).to(cuda))
    lin = torch.nn.Parameter(torch.randn(96, 32, generator=g).to(cuda))
    down = torch.nn.Parameter(torch.randn(48, 8, 16, generator=g).to(cuda))

    def check():
        sc, sl, sd = cache.get(conv, torch.bfloat16), cache.get(lin, torch.bfloat16), cache.get(down, torch.bfloat16)
        assert torch.equal(cache.layout(sc, "mirror"), sc.permute(2, 1, 0).flip(1).contiguous())
        assert torch.equal(cache.layout(sd, "keep"), sd.permute(2, 1, 0).contiguous())
        assert torch.equal(cache.layout(sl, "mirror"), sl.t().contiguous()[:, None, :])
        assert torch.equal(cache.layout(sl, "repeat", 2), sl.t().contiguous()[:, None, :].expand(-1, 2, -1).contiguous())
        assert cache.layout(sl.float(), "mirror") is None and cache.layout(sc[:, :, :16].contiguous(), "mirror") is None
        return cache.layout(sl, "mirror").data_ptr()

    p0 = check()
    with torch.no_grad():
        for p in (conv, lin, down):
            p.mul_(1.5).add_(0.25)
    assert check() == p0                      # same persistent buffers, new contents
    n = len(cache.layouts)
    del lin
    import gc
    gc.collect()
    assert len(cache.layouts) == n - 2        # both layouts of the Linear went with it


def test_weight_shadows_of_two_autocast_dtypes_in_one_process(cuda):
    """one model, bf16 autocast, fp16 autocast, optimizer-style weight updates in between: the 16-bit weight shadows of BOTH kinds are
    refreshed in their own dtype (round 5 bug: the mixed refresh went through torch._foreach_copy_, which wrote bf16 bit patterns
    into the f16 shadows -- SpUNet's stem was 31x off under fp16 after a bf16 run, tools/fp16_stem_probe.py)"""
    from pointcept_amd import functional as PF
    from pointcept_amd import nn as PNN

    torch.manual_seed(0)
    lin = PNN.Linear(64, 96).to(cuda)
    x = torch.randn(3000, 64, device=cuda)
    for step in range(3):
        ref = torch.nn.functional.linear(x, lin.weight, lin.bias)
        for dt, bar in ((torch.bfloat16, 2e-2), (torch.float16, 3e-3), (torch.bfloat16, 2e-2)):
            with torch.autocast("cuda", dtype=dt):
                y = lin(x)
            assert y.dtype == dt
            assert float((y.float() - ref).abs().max() / ref.abs().max()) < bar, (step, dt)
            assert torch.equal(PF._cast_cache.get(lin.weight, dt), lin.weight.detach().to(dt)), (step, dt)
        with torch.no_grad():
            lin.weight.mul_(1.25).add_(0.01)


# ------------------------------------------------------------------------------------------------
# M. libs/pointops2 (Stratified Transformer operators)
# ------------------------------------------------------------------------------------------------
def _p2_case(seed, M=80000, N=3500, hdim=16, h=6, L=31):
    """the workload of the reference's operator tests (libs/pointops2/functions/test_relative_pos_encoding_op_step1_v3.py:13-47)"""
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.rand(N, h, hdim, generator=g) for _ in range(3))
    tq, tk, tv = (torch.rand(L, h, hdim, 3, generator=g) for _ in range(3))
    i0 = torch.sort((torch.rand(M, generator=g) * N).long()).values
    i1 = (torch.rand(M, generator=g) * N).long()
    rel = (torch.rand(M, 3, generator=g) * L).long()
    attn = torch.rand(M, h, generator=g)
    return q, k, v, tq, tk, tv, i0, i1, rel, attn


def _p2_grads(fn, inputs, w):
    leaves = [t.clone().requires_grad_(True) for t in inputs]
    out = fn(*leaves)
    (out * w.to(out.device, out.dtype)).sum().backward()
    return out.detach(), [t.grad for t in leaves]


@pytest.mark.parametrize("seed", [1, 2])
def test_pointops2_pair_operators_match_the_reference_formulations(cuda, seed):
    """Every name of libs/pointops2/functions/pointops.py that Stratified Transformer calls, forward and gradients, against the torch
    formulations of the reference's own operator tests (oracle/pointops2.py, fp64).  v1 / v2 / v3 forms must agree with each other."""
    from oracle import pointops2 as orc
    from pointcept_amd import pointops2_api as p2

    q, k, v, tq, tk, tv, i0, i1, rel, attn = _p2_case(seed)
    N, M = q.shape[0], i0.numel()
    off = orc.offsets_of(i0, N)
    n_max = int((off[1:] - off[:-1]).max())
    dev = lambda *ts: [t.to(cuda) for t in ts]  # noqa: E731
    i0d, i1d, reld, offd = i0.int().to(cuda), i1.int().to(cuda), rel.int().to(cuda), off.to(cuda)
    g = torch.Generator().manual_seed(seed + 10)
    w_mh, w_n = torch.rand(M, q.shape[1], generator=g), torch.rand(N, q.shape[1], q.shape[2], generator=g)
    d64 = lambda *ts: [t.double() for t in ts]  # noqa: E731

    def cmp(tag, got, ref, rtol=2e-5):
        scale = float(ref.abs().max()) + 1e-30
        err = float((got.detach().cpu().double() - ref).abs().max())
        assert err <= rtol * scale, f"{tag}: max err {err:.3e} vs scale {scale:.3e}"

    # attention_step1 / _v2
    ref, rg = _p2_grads(lambda a, b: orc.attention_step1(a, b, i0, i1), d64(q, k), w_mh)
    for tag, fn in (("step1", lambda a, b: p2.attention_step1(a, b, i0d, i1d)), ("step1_v2", lambda a, b: p2.attention_step1_v2(a, b, i1d, offd, n_max))):
        out, gr = _p2_grads(fn, dev(q, k), w_mh)
        cmp(tag, out, ref)
        for name, a, b in zip(("dq", "dk"), gr, rg):
            cmp(f"{tag}.{name}", a, b, 1e-4)
    # dot_prod_with_idx (one table)
    ref, rg = _p2_grads(lambda a, t: orc.dot_prod_with_idx(a, i0, t, rel), d64(q, tq), w_mh)
    out, gr = _p2_grads(lambda a, t: p2.dot_prod_with_idx(a, i0d, t, reld), dev(q, tq), w_mh)
    cmp("dot_prod", out, ref)
    cmp("dot_prod.dq", gr[0], rg[0], 1e-4)
    cmp("dot_prod.dtable", gr[1], rg[1], 1e-4)
    # _v2 / _v3 (both tables)
    ref, rg = _p2_grads(lambda a, b, t, u: orc.dot_prod_with_idx_v3(a, i0, b, i1, t, u, rel), d64(q, k, tq, tk), w_mh)
    for tag, fn in (("dot_prod_v2", lambda a, b, t, u: p2.dot_prod_with_idx_v2(a, i0d, b, i1d, t, u, reld)),
                    ("dot_prod_v3", lambda a, b, t, u: p2.dot_prod_with_idx_v3(a, offd, n_max, b, i1d, t, u, reld))):
        out, gr = _p2_grads(fn, dev(q, k, tq, tk), w_mh)
        cmp(tag, out, ref)
        for name, a, b in zip(("dq", "dk", "dtq", "dtk"), gr, rg):
            cmp(f"{tag}.{name}", a, b, 1e-4)
    # attention_step2 / with_rel_pos_value / _v2
    ref, rg = _p2_grads(lambda a, b: orc.attention_step2(a, b, i0, i1, N), d64(attn, v), w_n)
    out, gr = _p2_grads(lambda a, b: p2.attention_step2(a, b, i0d, i1d), dev(attn, v), w_n[: int(i0.max()) + 1])
    cmp("step2", out, ref[: int(i0.max()) + 1], 1e-4)
    ref, rg = _p2_grads(lambda a, b, t: orc.attention_step2(a, b, i0, i1, N, t, rel), d64(attn, v, tv), w_n)
    for tag, fn in (("step2_rel", lambda a, b, t: p2.attention_step2_with_rel_pos_value(a, b, i0d, i1d, t, reld)),
                    ("step2_rel_v2", lambda a, b, t: p2.attention_step2_with_rel_pos_value_v2(a, b, offd, n_max, i1d, t, reld))):
        out, gr = _p2_grads(fn, dev(attn, v, tv), w_n)
        cmp(tag, out, ref, 1e-4)
        for name, a, b in zip(("dattn", "dv", "dtable"), gr, rg):
            cmp(f"{tag}.{name}", a, b, 1e-4)
    # the offsets form is a fixed-order segment loop: bit-reproducible
    a1 = p2.attention_step2_with_rel_pos_value_v2(attn.to(cuda), v.to(cuda), offd, n_max, i1d, tv.to(cuda), reld)
    a2 = p2.attention_step2_with_rel_pos_value_v2(attn.to(cuda), v.to(cuda), offd, n_max, i1d, tv.to(cuda), reld)
    assert torch.equal(a1, a2)


def test_pointops2_shared_families_and_import_names(cuda):
    """furthestsampling / knnquery / grouping / queryandgroup / interpolation under pointops2's argument order, through
    `import pointops2.pointops` as the Stratified Transformer files import it."""
    import pointcept_amd.compat as compat

    compat.install()
    import pointops2.pointops as pointops
    from oracle import pointops as orc1

    g = torch.Generator().manual_seed(3)
    xyz = torch.rand(3000, 3, generator=g)
    feat = torch.randn(3000, 8, generator=g)
    offset = torch.tensor([1400, 3000], dtype=torch.int32)
    new_offset = torch.tensor([350, 750], dtype=torch.int32)
    idx = pointops.furthestsampling(xyz.to(cuda), offset.to(cuda), new_offset.to(cuda))
    ref_idx = orc1.farthest_point_sampling(xyz.numpy(), offset.numpy(), new_offset.numpy())
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    new_xyz = xyz[idx.cpu().long()]
    nidx, dist = pointops.knnquery(8, xyz.to(cuda), new_xyz.to(cuda), offset.to(cuda), new_offset.to(cuda))
    ridx, rdist = orc1.knn_query(8, xyz.numpy(), offset.numpy(), new_xyz.numpy(), new_offset.numpy())
    assert np.array_equal(nidx.cpu().numpy(), ridx) and np.allclose(dist.cpu().numpy(), rdist, atol=1e-6)
    grouped = pointops.queryandgroup(8, xyz.to(cuda), new_xyz.to(cuda), feat.to(cuda), None, offset.to(cuda), new_offset.to(cuda), use_xyz=True)
    ref_g = torch.cat([xyz[torch.from_numpy(ridx).long()] - new_xyz[:, None], feat[torch.from_numpy(ridx).long()]], -1)
    assert torch.allclose(grouped.cpu(), ref_g, atol=1e-6)
    assert torch.equal(pointops.grouping(feat.to(cuda), nidx).cpu(), feat[torch.from_numpy(ridx).long()])
    up = pointops.interpolation(new_xyz.to(cuda), xyz.to(cuda), feat[idx.cpu().long()].to(cuda), new_offset.to(cuda), offset.to(cuda))
    assert up.shape == (3000, 8) and torch.isfinite(up).all()

