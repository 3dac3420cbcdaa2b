#!/usr/bin/env python
"""Per-operator timings of the hot path at the PTv3-base / ScanNet shapes (8 scenes x 102400 voxels),
engine kernel vs the PyTorch-ROCm library op for the same math.  HIP-event timing on torch's current
stream (the stream the engine launches on).  Writes gpurun_out/bench_ops.json and prints a table.

    python tools/bench_ops.py [--quick]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pointcept_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
HBM_PEAK = 8.0e12
MFMA_PEAK = 2.5e15


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3  # seconds


def roof(bytes_, flops, t):
    t_roof = max(bytes_ / HBM_PEAK, flops / MFMA_PEAK)
    return {"us": round(t * 1e6, 1), "GBps": round(bytes_ / t / 1e9, 1), "TFLOPs": round(flops / t / 1e12, 1),
            "roof_frac": round(t_roof / t, 3)}


def bench_linear(rows, n, cin, cout, results):
    dt = torch.bfloat16
    x = torch.randn(n, cin, device=DEV).to(dt)
    w = (torch.randn(cout, cin, device=DEV) / cin ** 0.5).to(dt)
    b = torch.randn(cout, device=DEV)
    g = torch.randn(n, cout, device=DEV).to(dt)
    wt = w.t().contiguous()[:, None, :]
    e = 2
    by_f = n * (cin + cout) * e + cin * cout * e
    fl = 2.0 * n * cin * cout
    r = {"shape": [n, cin, cout]}
    r["own_fwd"] = roof(by_f, fl, timeit(lambda: ops.spconv_fwd(x, w[:, None, :], b, None)))
    r["lib_fwd"] = roof(by_f, fl, timeit(lambda: F.linear(x, w, b.to(dt))))
    r["own_dgrad"] = roof(by_f, fl, timeit(lambda: ops.spconv_fwd(g, wt, None, None)))
    r["lib_dgrad"] = roof(by_f, fl, timeit(lambda: g @ w))
    by_w = n * (cin + cout) * e + cin * cout * 4
    r["own_wgrad"] = roof(by_w, fl, timeit(lambda: ops.spconv_wgrad(x, g, None, want_bias=True)))
    r["lib_wgrad"] = roof(by_w, fl, timeit(lambda: (g.t() @ x, g.sum(0))))
    results.append(r)
    rows.append(f"linear n={n:7d} {cin:4d}->{cout:4d} | fwd own {r['own_fwd']['us']:8.1f} lib {r['lib_fwd']['us']:8.1f} | "
                f"dgrad own {r['own_dgrad']['us']:8.1f} lib {r['lib_dgrad']['us']:8.1f} | "
                f"wgrad own {r['own_wgrad']['us']:8.1f} lib {r['lib_wgrad']['us']:8.1f} | own fwd roof {r['own_fwd']['roof_frac']}")


def bench_mlp(rows, n, c, results, dt=torch.bfloat16):
    """csrc/mlp.hip against the split kernels it replaces (fc1 + GELU, fc2 + joint; GELU' dgrad, fc1 dgrad, two weight gradients)"""
    hid = 4 * c
    x = torch.randn(n, c, device=DEV).to(dt)
    w1 = (torch.randn(hid, c, device=DEV) / c ** 0.5).to(dt)
    w2 = (torch.randn(c, hid, device=DEV) / hid ** 0.5).to(dt)
    b1, b2 = torch.randn(hid, device=DEV) * 0.3, torch.randn(c, device=DEV)
    a = torch.randn(n, c, device=DEV)
    dm = torch.randn(n, c, device=DEV).to(dt)
    w2t, w1t = w2.t().contiguous(), w1.t().contiguous()[:, None, :].contiguous()
    h, act = ops.linear_gelu_fwd(x, w1, b1)
    dh = ops.linear_gelu_bwd_input(dm, w2t, h)

    def split_fwd():
        _, a_ = ops.linear_gelu_fwd(x, w1, b1)
        return ops.linear_joint_fwd(a_, w2, b2, None, a, None, None, dt)

    def split_bwd():
        d = ops.linear_gelu_bwd_input(dm, w2t, h)
        ops.spconv_fwd(d, w1t, None, None)
        ops.spconv_wgrad(act, dm, None, want_bias=True)
        ops.spconv_wgrad(x, d, None, want_bias=True)

    r = {"shape": [n, c]}
    fl_f, fl_b = 4.0 * n * c * hid, 10.0 * n * c * hid
    r["fused_fwd"] = roof(n * c * 12, fl_f, timeit(lambda: ops.mlp_fwd(x, w1, b1, w2, b2, a, None)))
    r["split_fwd"] = roof(n * c * 12, fl_f, timeit(split_fwd))
    r["fused_bwd"] = roof(n * c * 6, fl_b, timeit(lambda: ops.mlp_bwd(dm, x, w1, b1, w2t)))
    r["split_bwd"] = roof(n * c * 6, fl_b, timeit(split_bwd))
    results.append(r)
    rows.append(f"mlp n={n:7d} c={c:3d} {str(dt)[6:]:8s} | fwd fused {r['fused_fwd']['us']:7.1f} us ({r['fused_fwd']['GBps']:.0f} GB/s alg, {fl_f / r['fused_fwd']['us'] / 1e6:.0f} TF/s) "
                f"split {r['split_fwd']['us']:7.1f} | bwd fused {r['fused_bwd']['us']:7.1f} us ({r['fused_bwd']['GBps']:.0f} GB/s alg, {fl_b / r['fused_bwd']['us'] / 1e6:.0f} TF/s) "
                f"split {r['split_bwd']['us']:7.1f}")


def bench_ln(rows, n, c, results):
    x = torch.randn(n, c, device=DEV)
    gm, bt = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV)
    dy = torch.randn(n, c, device=DEV)
    dyb = dy.to(torch.bfloat16)
    y, mean, rstd = ops.layer_norm_fwd(x, gm, bt, 1e-5, torch.float32)
    r = {"shape": [n, c]}
    r["own_fwd_f32"] = roof(n * c * 8, 0, timeit(lambda: ops.layer_norm_fwd(x, gm, bt, 1e-5, torch.float32)))
    r["own_fwd_bf16out"] = roof(n * c * 6, 0, timeit(lambda: ops.layer_norm_fwd(x, gm, bt, 1e-5, torch.bfloat16)))
    r["lib_fwd_f32"] = roof(n * c * 8, 0, timeit(lambda: F.layer_norm(x, (c,), gm, bt, 1e-5)))
    r["own_bwd_f32"] = roof(n * c * 12, 0, timeit(lambda: ops.layer_norm_bwd(dy, x, mean, rstd, gm)))
    r["own_bwd_bf16dy"] = roof(n * c * 10, 0, timeit(lambda: ops.layer_norm_bwd(dyb, x, mean, rstd, gm)))
    xr = x.clone().requires_grad_(True)
    gr, br = gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (c,), gr, br, 1e-5)
    r["lib_bwd_f32"] = roof(n * c * 12, 0, timeit(lambda: torch.autograd.grad(yr, (xr, gr, br), dy, retain_graph=True)))
    results.append(r)
    rows.append(f"layernorm n={n:7d} c={c:4d} | fwd own {r['own_fwd_f32']['us']:7.1f} ({r['own_fwd_f32']['GBps']:.0f} GB/s) "
                f"bf16out {r['own_fwd_bf16out']['us']:7.1f} lib {r['lib_fwd_f32']['us']:7.1f} | bwd own {r['own_bwd_f32']['us']:7.1f} "
                f"({r['own_bwd_f32']['GBps']:.0f} GB/s) bf16dy {r['own_bwd_bf16dy']['us']:7.1f} lib {r['lib_bwd_f32']['us']:7.1f}")


def bench_attention(rows, n_seq, H, results, L=1024):
    T = n_seq * L
    qkv = torch.randn(T, 3, H, 16, device=DEV).to(torch.bfloat16)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=DEV)
    sc = 0.25
    out, lse = ops.attn_varlen_fwd(qkv, cu, L, sc)
    do = torch.randn_like(out)
    fl_f = 4.0 * L * L * 16 * n_seq * H
    by_f = T * H * 16 * 2 * 4
    r = {"shape": [n_seq, L, H]}
    r["fwd"] = roof(by_f, fl_f, timeit(lambda: ops.attn_varlen_fwd(qkv, cu, L, sc), iters=10))
    r["bwd"] = roof(T * H * 16 * 2 * 8, 10.0 * L * L * 16 * n_seq * H,
                    timeit(lambda: ops.attn_varlen_bwd(qkv, out, do, lse, cu, L, sc), iters=10))
    # the two forms of the backward (attention.hip: PTC_AT_BWD1 is read per call): split dQ | dK dV kernels, one-pass kernel
    prev = os.environ.get("PTC_AT_BWD1")
    for key, val in (("bwd_split", "0"), ("bwd_one_pass", "1")):
        os.environ["PTC_AT_BWD1"] = val
        r[key] = roof(T * H * 16 * 2 * 8, 10.0 * L * L * 16 * n_seq * H,
                      timeit(lambda: ops.attn_varlen_bwd(qkv, out, do, lse, cu, L, sc), iters=10))
    if prev is None:
        del os.environ["PTC_AT_BWD1"]
    else:
        os.environ["PTC_AT_BWD1"] = prev
    results.append(r)
    rows.append(f"attention n_seq={n_seq:4d} L={L} H={H:2d} | fwd {r['fwd']['us']:8.1f} us {r['fwd']['TFLOPs']:7.1f} TF/s | "
                f"bwd {r['bwd']['us']:8.1f} us {r['bwd']['TFLOPs']:7.1f} TF/s (split kernels {r['bwd_split']['us']:8.1f} us, one-pass "
                f"{r['bwd_one_pass']['us']:8.1f} us)")


def bench_spconv(rows, results, scenes=8, points=102400):
    from pointcept_amd import synthetic

    b = synthetic.to_torch(synthetic.indoor_batch(scenes, points), DEV)
    n = b["grid_coord"].shape[0]
    batch = torch.repeat_interleave(torch.arange(scenes, device=DEV), torch.diff(b["offset"], prepend=b["offset"].new_zeros(1)))
    ind = torch.cat([batch[:, None].int(), b["grid_coord"].int()], 1).contiguous()
    r = {"n": n}
    t_hash = timeit(lambda: ops.HashTable(ind), iters=5)
    table = ops.HashTable(ind)
    for ks in (3, 5):
        r[f"rulebook_k{ks}_us"] = round(timeit(lambda: ops.rulebook_subm(ind, ks, table), iters=5) * 1e6, 1)
    r["hash_us"] = round(t_hash * 1e6, 1)
    nbr = ops.rulebook_subm(ind, 3, table)
    pairs = int((nbr >= 0).sum())
    r["pairs_k3"] = pairs
    for c in (32, 64):
        x = torch.randn(n, c, device=DEV).to(torch.bfloat16)
        w = (torch.randn(c, 27, c, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(c, device=DEV)
        g = torch.randn(n, c, device=DEV).to(torch.bfloat16)
        by = n * c * 2 * 2 + 4 * 27 * n + 27 * c * c * 2
        fl = 2.0 * pairs * c * c
        r[f"fwd_c{c}"] = roof(by, fl, timeit(lambda: ops.spconv_fwd(x, w, bias, nbr), iters=10))
        r[f"wgrad_c{c}"] = roof(by, fl, timeit(lambda: ops.spconv_wgrad(x, g, nbr), iters=10))
        rows.append(f"spconv k3 n={n} c={c} pairs={pairs} | fwd {r[f'fwd_c{c}']['us']:8.1f} us ({r[f'fwd_c{c}']['GBps']:.0f} GB/s alg, "
                    f"{r[f'fwd_c{c}']['TFLOPs']:.1f} TF/s) | wgrad {r[f'wgrad_c{c}']['us']:8.1f} us")
    rows.append(f"rulebook n={n}: hash {r['hash_us']} us, k3 {r['rulebook_k3_us']} us, k5 {r['rulebook_k5_us']} us")
    # serialization + sort
    code_t = timeit(lambda: ops.serialize_encode(b["grid_coord"], batch, 8, ("z", "z-trans", "hilbert", "hilbert-trans")), iters=10)
    code = ops.serialize_encode(b["grid_coord"], batch, 8, ("z", "z-trans", "hilbert", "hilbert-trans"))
    sort_t = timeit(lambda: ops.sort_keys(code, 0, 28), iters=10)
    r["encode"] = roof(n * 48, 0, code_t)
    r["sort4"] = roof(4 * n * (16 * 4 + 16), 0, sort_t)
    rows.append(f"serialize n={n}: encode(4 orders) {r['encode']['us']} us ({r['encode']['GBps']} GB/s), sort 4x28bit {r['sort4']['us']} us")
    idx = torch.randperm(n, device=DEV)
    for c in (96, 192):
        src = torch.randn(n, c, device=DEV).to(torch.bfloat16)
        t = timeit(lambda: ops.gather_rows(src, idx), iters=10)
        r[f"gather_c{c}"] = roof(n * c * 4 + n * 8, 0, t)
        rows.append(f"gather_rows n={n} c={c} bf16: {r[f'gather_c{c}']['us']} us ({r[f'gather_c{c}']['GBps']} GB/s)")
    results.append(r)


def bench_spconv_stages(rows, results, scenes=8, points=102400, outdoor=False):
    """CPE convolution (k=3 SubM, C -> C) forward at the five PTv3 stages of the bench batch, rows in
    Hilbert order as in the model (config.SORT_POINTS), conv2 vs conv3."""
    from pointcept_amd import synthetic

    if outdoor:   # BASELINE configs[4]: LiDAR sweeps on a depth-12 grid; pooling thins these scenes far less than the indoor ones
        b = synthetic.to_torch(synthetic.collate([synthetic.outdoor_scene(5000 + i, azimuth_steps=3300) for i in range(scenes)]), DEV)
    else:
        b = synthetic.to_torch(synthetic.indoor_batch(scenes, points), DEV)
    depth = 12 if outdoor else 8
    batch = torch.repeat_interleave(torch.arange(scenes, device=DEV), torch.diff(b["offset"], prepend=b["offset"].new_zeros(1)))
    gc = b["grid_coord"]
    for s, chans in enumerate(((32, 64), (64,), (128,), (256,), (512,)) if outdoor else ((32, 64, (128, 96), 96), (64, 32, 128), (128,), (256,), (512,))):
        key = (batch << 48) | ((gc[:, 0] >> s) << 32) | ((gc[:, 1] >> s) << 16) | (gc[:, 2] >> s)
        uk = torch.unique(key)
        bb = uk >> 48
        cc = torch.stack([(uk >> 32) & 0xffff, (uk >> 16) & 0xffff, uk & 0xffff], 1)
        code = ops.serialize_encode(cc, bb, depth - s, ("hilbert",))
        order, _ = ops.sort_keys(code, 0, 3 * (depth - s) + 3)
        cc, bb = cc[order[0]], bb[order[0]]
        ind = torch.cat([bb[:, None].int(), cc.int()], 1).contiguous()
        n = ind.shape[0]
        table = ops.HashTable(ind)
        nbr = ops.rulebook_subm(ind, 3, table)
        pairs = int((nbr >= 0).sum())
        t_h = timeit(lambda: ops.HashTable(ind), iters=5)
        t_3 = timeit(lambda: ops.rulebook_subm(ind, 3, table), iters=5)
        t_5 = timeit(lambda: ops.rulebook_subm(ind, 5, table), iters=5) if s == 0 else 0.0
        rows.append(f"rulebook stage {s} n={n:7d} (curve order): hash {t_h * 1e6:7.1f} us | k3 {t_3 * 1e6:7.1f} us "
                    f"({27 * n * 4 / t_3 / 1e9:.0f} GB/s table out) | k5 {t_5 * 1e6:7.1f} us")
        for cc in chans:
            ci, c = cc if isinstance(cc, tuple) else (cc, cc)
            x = torch.randn(n, ci, device=DEV).to(torch.bfloat16)
            w = (torch.randn(c, 27, ci, device=DEV) * 0.05).to(torch.bfloat16)
            bias = torch.randn(c, device=DEV)
            by = n * (ci + c) * 2 + 4 * 27 * n + 27 * ci * c * 2
            fl = 2.0 * pairs * ci * c
            r = {"stage": s, "n": n, "c_in": ci, "c": c, "pairs": pairs}
            r["conv"] = roof(by, fl, timeit(lambda: ops.spconv_fwd(x, w, bias, nbr), iters=10))   # conv5 (c_in 32 / 64) | conv3
            plan = ops.block_plan(ci, c, 27, torch.bfloat16)
            c4 = ""
            if plan is not None:
                t_b = timeit(lambda: ops.BlockTables(nbr, *plan), iters=5)
                blk = ops.BlockTables(nbr, *plan)
                r["conv7"] = roof(by, fl, timeit(lambda: ops.spconv_fwd(x, w, bias, nbr, blk), iters=10))
                r["blocks_us"] = round(t_b * 1e6, 1)
                r["blocks_overflow"] = int(blk.n_overflow.item())
                r["halo_mean"] = round(float(blk.hcnt.float().mean()), 1)
                c4 = (f" | block-staged {r['conv7']['us']:8.1f} us ({r['conv7']['GBps']:.0f} GB/s alg, {r['conv7']['TFLOPs']:.1f} TF/s; tables {r['blocks_us']:.1f} us, "
                      f"halo {r['halo_mean']:.0f}/{plan[0]}, {r['blocks_overflow']} overflow)")
            g = torch.randn(n, c, device=DEV).to(torch.bfloat16)
            r["wgrad"] = roof(by, fl, timeit(lambda: ops.spconv_wgrad(x, g, nbr), iters=10))
            w7 = ""
            if plan is not None:   # the accumulator-stationary weight gradient on the block tables (csrc/wgrad7.h; channel slices above 64)
                r["wgrad7"] = roof(by, fl, timeit(lambda: ops.spconv_wgrad(x, g, nbr, blk=blk), iters=10))
                w7 = f" | block-staged wgrad {r['wgrad7']['us']:8.1f} us ({r['wgrad7']['TFLOPs']:.1f} TF/s)"
            results.append(r)
            rows.append(f"conv stage {s} n={n:7d} {ci:3d}->{c:3d} pairs/pt={pairs / n:5.2f} | global gathers {r['conv']['us']:8.1f} us "
                        f"({r['conv']['GBps']:.0f} GB/s alg, {r['conv']['TFLOPs']:.1f} TF/s){c4} | wgrad {r['wgrad']['us']:8.1f} us ({r['wgrad']['TFLOPs']:.1f} TF/s){w7}")


def bench_losses(rows, results, n=819200, c=20):
    """CE and Lovasz-Softmax on the seg-head output of the BASELINE batch (bf16, strided [N, 32] storage)."""
    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(0)
    wide = torch.randn(n, 32, generator=g).to(torch.bfloat16).to(DEV)
    y = torch.randint(-1, c, (n,), generator=g).to(DEV)
    logits = wide[:, :c]
    t_ce = timeit(lambda: ops.cross_entropy_fwd(logits, y, -1), 10, 2)
    t_lv = timeit(lambda: ops.lovasz_softmax(logits, y, -1), 10, 2)
    slot_bytes = 190.0 * n * c   # keys 8w + sort 4 passes x (8r + 8r + 12w) + order 8w+8r + fg 4w+4r + scan 8w+8r + g 4w+4r ...

    def aten():
        p = logits.float().softmax(1)
        tot = 0
        for k in range(c):
            fg = (y == k).float()
            e, perm = torch.sort((fg - p[:, k]).abs(), descending=True)
            tot = tot + e.sum() + fg[perm].cumsum(0)[-1]
        return tot

    t_at = timeit(aten, 3, 1)
    r = {"n": n, "c": c, "ce_fwd_us": round(t_ce * 1e6, 1), "lovasz_us": round(t_lv * 1e6, 1),
         "lovasz_GBps": round(slot_bytes / t_lv / 1e9, 1), "aten_sorts_only_us": round(t_at * 1e6, 1)}
    results.append(r)
    rows.append(f"losses [{n} x {c}] bf16: CE fwd {r['ce_fwd_us']:.1f} us | Lovasz fwd+grad {r['lovasz_us']:.1f} us "
                f"({r['lovasz_GBps']:.0f} GB/s of ~190 B/slot) | ATen softmax + {c} sorts + cumsums alone {r['aten_sorts_only_us']:.1f} us")


def bench_front_end(rows, results):
    """the 8(f) rows: device GridSample on a raw 2M-point scan, kNN at the evaluator's shape (predictions of a 100k-voxel
    scene carried to its 250k raw points, evaluator.py:569), k = 16 self-query, farthest point sampling 100k -> 2048."""
    from pointcept_amd import pointops_api as po
    from pointcept_amd.transform import GridSample

    g = torch.Generator().manual_seed(0)
    raw = ((torch.rand(2_000_000, 3, generator=g) - 0.5) * torch.tensor([7.0, 5.0, 2.8])).to(DEV)
    gs = GridSample(grid_size=0.02, mode="train", return_grid_coord=True)
    t_gs = timeit(lambda: gs(dict(coord=raw, index_valid_keys=["coord"])), 5, 1)
    vox = torch.rand(100_000, 3, generator=g).to(DEV)
    pts = torch.rand(250_000, 3, generator=g).to(DEV)
    o1, o2 = torch.tensor([100_000], device=DEV), torch.tensor([250_000], device=DEV)
    t_k1 = timeit(lambda: po.knn_query(1, vox, o1, pts, o2), 5, 1)
    t_k16 = timeit(lambda: po.knn_query(16, vox, o1), 3, 1)
    t_fps = timeit(lambda: po.farthest_point_sampling(vox, o1, torch.tensor([2048], device=DEV)), 3, 1)
    r = {"gridsample_2M_us": round(t_gs * 1e6, 1), "knn1_100k_x_250k_us": round(t_k1 * 1e6, 1),
         "knn16_100k_self_us": round(t_k16 * 1e6, 1), "fps_100k_2048_us": round(t_fps * 1e6, 1),
         "knn1_Gdist_per_s": round(100_000 * 250_000 / t_k1 / 1e9, 1), "knn16_Gdist_per_s": round(100_000 * 100_000 / t_k16 / 1e9, 1)}
    results.append(r)
    rows.append(f"front end: GridSample 2M pts {r['gridsample_2M_us']:.0f} us | kNN k=1 100k x 250k {r['knn1_100k_x_250k_us']:.0f} us "
                f"({r['knn1_Gdist_per_s']:.0f} G dist/s) | kNN k=16 100k self {r['knn16_100k_self_us']:.0f} us "
                f"({r['knn16_Gdist_per_s']:.0f} G dist/s) | FPS 100k -> 2048 {r['fps_100k_2048_us']:.0f} us")


def bench_attention_hd(rows, n_seq, H, D, results, L=1024):
    """head_dim 17..64 kernels (attention_hd.h) at a PT-v3m3 / LitePT-like shape, beside the SDPA library path."""
    T = n_seq * L
    qkv = torch.randn(T, 3, H, D, device=DEV).to(torch.bfloat16)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=DEV)
    sc = D ** -0.5
    out, lse = ops.attn_varlen_fwd(qkv, cu, L, sc)
    do = torch.randn_like(out)
    fl = L * L * D * n_seq * H
    r = {"shape": [n_seq, L, H, D]}
    r["fwd"] = roof(T * H * D * 8, 4.0 * fl, timeit(lambda: ops.attn_varlen_fwd(qkv, cu, L, sc), iters=10))
    r["bwd"] = roof(T * H * D * 16, 10.0 * fl, timeit(lambda: ops.attn_varlen_bwd(qkv, out, do, lse, cu, L, sc), iters=10))
    blk = qkv.reshape(n_seq, L, 3, H, D).permute(2, 0, 3, 1, 4).contiguous()
    r["sdpa_fwd"] = roof(T * H * D * 8, 4.0 * fl, timeit(lambda: F.scaled_dot_product_attention(blk[0], blk[1], blk[2], scale=sc), iters=10))
    rope = ""
    if ops.attn_rope_supported(D, L):   # the rotation of q / k as its own pass (+ its inverse on the gradient) against the fused prologue / epilogue
        xyz = torch.rand(T, 3, device=DEV) * 40.0
        inv_freq = (1.0 / (100.0 ** (torch.arange(D // 6, dtype=torch.float32, device=DEV) / (D // 6))))
        t_pass = timeit(lambda: ops.rope3d_xyz(qkv, xyz, inv_freq, 2, 1.0, torch.bfloat16), iters=10)
        t_ff = timeit(lambda: ops.attn_rope_fwd(qkv, xyz, inv_freq, cu, L, sc), iters=10)
        t_fb = timeit(lambda: ops.attn_rope_bwd(qkv, out, do, lse, xyz, inv_freq, cu, L, sc), iters=10)
        r["rope_pass_us"], r["rope_fused_fwd_us"], r["rope_fused_bwd_us"] = round(t_pass * 1e6, 1), round(t_ff * 1e6, 1), round(t_fb * 1e6, 1)
        rope = (f" | rotation pass {t_pass * 1e6:6.1f} us per direction; with the rotation fused: fwd {t_ff * 1e6:8.1f} us "
                f"(two-pass {r['fwd']['us'] + t_pass * 1e6:8.1f}), bwd {t_fb * 1e6:8.1f} us (two-pass {r['bwd']['us'] + t_pass * 1e6:8.1f})")
    results.append(r)
    rows.append(f"attention_hd n_seq={n_seq:4d} L={L} H={H:2d} D={D:2d} | fwd {r['fwd']['us']:8.1f} us {r['fwd']['TFLOPs']:7.1f} TF/s | "
                f"bwd {r['bwd']['us']:8.1f} us {r['bwd']['TFLOPs']:7.1f} TF/s | library SDPA fwd on pre-gathered [n,H,L,D] {r['sdpa_fwd']['us']:8.1f} us{rope}")


def bench_attention_rpe(rows, n_seq, H, L, results):
    """A13: RPE attention kernels (attention_rpe.h) beside the reference's dense formulation (ptv3m1:190-206) in torch on the GPU."""
    bnd = int((4 * L) ** (1 / 3) * 2)
    T = n_seq * L
    qkv = torch.randn(T, 3, H, 16, device=DEV).to(torch.bfloat16)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=DEV)
    gc = torch.randint(0, 2 * bnd, (T, 3), device=DEV, dtype=torch.int32)
    table = torch.randn(3 * (2 * bnd + 1), H, device=DEV) * 0.02
    sc = 0.25
    out, lse = ops.attn_rpe_fwd(qkv, cu, L, sc, gc, table, bnd)
    do = torch.randn_like(out)
    fl = L * L * 16 * n_seq * H
    r = {"shape": [n_seq, L, H], "pos_bnd": bnd}
    r["fwd"] = roof(T * H * 16 * 8, 4.0 * fl, timeit(lambda: ops.attn_rpe_fwd(qkv, cu, L, sc, gc, table, bnd), iters=5))
    r["bwd"] = roof(T * H * 16 * 16, 10.0 * fl, timeit(lambda: ops.attn_rpe_bwd(qkv, out, do, lse, cu, L, sc, gc, table, bnd), iters=5))

    def dense():
        q, k, v = qkv.reshape(n_seq, L, 3, H, 16).permute(2, 0, 3, 1, 4).unbind(0)
        g = gc.reshape(n_seq, L, 3).long()
        rel = g.unsqueeze(2) - g.unsqueeze(1)
        idx = rel.clamp(-bnd, bnd) + bnd + torch.arange(3, device=DEV) * (2 * bnd + 1)
        bias = table.index_select(0, idx.reshape(-1)).view(idx.shape + (-1,)).sum(3).permute(0, 3, 1, 2)
        attn = torch.softmax((q.float() * sc) @ k.float().transpose(-2, -1) + bias, dim=-1).to(qkv.dtype)
        return (attn @ v).transpose(1, 2)

    try:
        r["dense_fwd_us"] = timeit(dense, iters=3) * 1e6
    except RuntimeError as e:   # out of memory at large shapes is the point of the comparison
        r["dense_fwd_us"] = float("nan")
        r["dense_error"] = str(e)[:80]
    results.append(r)
    rows.append(f"attention_rpe n_seq={n_seq:4d} L={L} H={H:2d} bnd={bnd} | fwd {r['fwd']['us']:8.1f} us | bwd {r['bwd']['us']:8.1f} us | "
                f"dense torch formulation fwd {r['dense_fwd_us']:10.1f} us ([P,H,K,K] fp32 = {n_seq * H * L * L * 4 / 1e9:.2f} GB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="comma list of sections: linear,ln,attn,attn_hd,spconv,stages,losses,front")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    want = lambda name: not only or name in only  # noqa: E731
    rows, res = [], {"linear": [], "ln": [], "attn": [], "spconv": [], "stages": [], "losses": [], "front": []}
    stages = [(819200, 32), (202560, 64), (49256, 128), (11400, 256), (2640, 512)]
    if args.quick:
        stages = stages[:2]
    if want("linear"):
        for n, c in stages:
            for cin, cout in ((c, 3 * c), (c, c), (c, 4 * c), (4 * c, c)):
                bench_linear(rows, n, cin, cout, res["linear"])
        bench_linear(rows, 819200, 64, 192, res["linear"])
        bench_linear(rows, 819200, 64, 256, res["linear"])
        bench_linear(rows, 819200, 256, 64, res["linear"])
        bench_linear(rows, 819200, 64, 32, res["linear"])   # seg head (20 padded to 32)
    if want("mlp"):
        res["mlp"] = []
        for n, c in ((819200, 64), (819200, 32), (202560, 64), (202560, 32)):
            bench_mlp(rows, n, c, res["mlp"])
        bench_mlp(rows, 819200, 64, res["mlp"], torch.float16)
    if "linprobe" in only:   # shape sweep at the stage-0 row count: which of (c_in, c_out) costs the bandwidth?
        res["linprobe"] = []
        for cin, cout in ((32, 32), (32, 64), (32, 128), (32, 256), (64, 32), (64, 64), (64, 128), (64, 192), (64, 256), (128, 64),
                          (128, 128), (128, 256), (256, 64), (256, 128)):
            bench_linear(rows, 819200, cin, cout, res["linprobe"])
    if "rope" in only:   # PT-v3m3 / LitePT rotation on the packed qkv rows: ptc_rope3d_xyz against the torch formulation it replaces
        from pointcept_amd import functional as PF
        res["rope"] = []
        for n, H, D in ((819200, 3, 18), (204800, 6, 18), (51200, 12, 18), (819200, 2, 24)):
            qkv = torch.randn(n, 3, H, D, device=DEV).to(torch.bfloat16)
            xyz = torch.rand(n, 3, device=DEV) * 8.0
            f = (1.0 / (10.0 ** (torch.arange(0, D // 3, 2).float() / (D // 3)))).to(DEV)
            t_k = timeit(lambda: ops.rope3d_xyz(qkv, xyz, f, 2, 1.0, torch.bfloat16))
            t_t = timeit(lambda: PF.rope_xyz_torch(qkv, xyz, f))
            by = n * 3 * H * D * 2 * 2 + n * 12
            res["rope"].append({"shape": [n, H, D], "kernel": roof(by, 0.0, t_k), "torch_ops": roof(by, 0.0, t_t)})
            rows.append(f"rope n={n:7d} H={H:2d} D={D:2d} | kernel {t_k * 1e6:8.1f} us ({by / t_k / 1e9:6.0f} GB/s)  torch ops {t_t * 1e6:8.1f} us")
    if "wgrad_small" in only:   # the weight gradients of the deep stages (~103 launches of ~29 us per step), own vs library
        res["wgrad_small"] = []
        dt = torch.bfloat16
        for n, cin, cout in ((11400, 256, 768), (11400, 256, 256), (11400, 256, 1024), (11400, 1024, 256), (2640, 512, 1536), (2640, 512, 512),
                             (49256, 128, 384), (49256, 128, 128), (49256, 128, 512), (49256, 512, 128)):
            x, g = torch.randn(n, cin, device=DEV).to(dt), torch.randn(n, cout, device=DEV).to(dt)
            t_own = timeit(lambda: ops.spconv_wgrad(x, g, None, want_bias=True), iters=50)
            t_lib = timeit(lambda: (g.t() @ x, g.sum(0)), iters=50)
            by, fl = n * (cin + cout) * 2 + cin * cout * 4, 2.0 * n * cin * cout
            res["wgrad_small"].append({"shape": [n, cin, cout], "own": roof(by, fl, t_own), "lib": roof(by, fl, t_lib)})
            rows.append(f"wgrad n={n:6d} {cin:4d}->{cout:4d} | own {t_own * 1e6:7.1f} us  lib {t_lib * 1e6:7.1f} us | roof frac own "
                        f"{res['wgrad_small'][-1]['own']['roof_frac']}")
    if want("ln"):
        for n, c in stages:
            bench_ln(rows, n, c, res["ln"])
        bench_ln(rows, 819200, 64, res["ln"])
    if want("attn"):
        for n_seq, H in ((800, 2), (800, 4), (200, 4), (48, 8), (16, 16), (12, 16), (8, 16), (3, 32)):
            bench_attention(rows, n_seq, H, res["attn"])
    if "attn_rpe" in only:
        res["attn_rpe"] = []
        for n_seq, H, L in ((64, 4, 256), (200, 4, 1024), (800, 4, 1024)):
            bench_attention_rpe(rows, n_seq, H, L, res["attn_rpe"])
    if want("attn_hd"):
        res["attn_hd"] = []
        for n_seq, H, D, L in ((800, 3, 18, 1024), (200, 6, 18, 1024), (48, 12, 18, 1024), (200, 4, 32, 1024), (200, 4, 48, 672), (200, 4, 64, 512)):
            bench_attention_hd(rows, n_seq, H, D, res["attn_hd"], L)
    if want("spconv"):
        bench_spconv(rows, res["spconv"])
    if want("stages"):
        bench_spconv_stages(rows, res["stages"])
    if "stages_outdoor" in only:   # the same table on the LiDAR geometry of BASELINE configs[4] (deep stages keep 19-43 % of the voxels) + its dense Linears
        bench_spconv_stages(rows, res["stages"], outdoor=True)
        for n, c in ((437000, 256), (274000, 512)):
            for cin, cout in ((c, 3 * c), (c, c), (c, 4 * c), (4 * c, c)):
                bench_linear(rows, n, cin, cout, res["linear"])
    if want("losses"):
        bench_losses(rows, res["losses"])
    if want("front"):
        bench_front_end(rows, res["front"])
    print("\n".join(rows))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_ops.json"), "w") as f:
        json.dump(res, f, indent=1)
    with open(os.path.join(ROOT, "gpurun_out", "bench_ops.txt"), "w") as f:
        f.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
