#!/usr/bin/env python
"""Launch ONLY the gather-table convolution family at the bench shapes (PT-v3m1 base, 8 x 102400 voxels, rows in
curve order as in the model) a few times each.  Meant to run under rocprofv3 (`--kernel-trace --stats`, and separate
`--pmc ...` passes): tools/pmc_summary.py turns the output into profiles/*_conv_pmc.json.

    PTC_CK_CASE=s0   stage 0 (N = 819200): conv 64->64 (dec0 CPE), conv 32->32 (enc0 CPE), their weight gradients,
                     the gather-fused qkv GEMMs 32->96 / 64->192, rulebook k=3 / k=5 and the hash build
    PTC_CK_CASE=s1   stage 1 (N ~ 202k): conv 64->64 and its weight gradient, rulebook k=3
Algorithmic bytes / flops per launch are printed so the summary can put traffic beside them.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcept_amd import ops, synthetic  # noqa: E402

DEV = torch.device("cuda:0")
CASE = os.environ.get("PTC_CK_CASE", "s0")
ITERS = int(os.environ.get("PTC_CK_ITERS", "4"))
scenes, points = 8, 102400


def stage_indices(s):
    cache = f"/tmp/ptc_ck_stage{s}.pt"
    if os.path.exists(cache):
        return torch.load(cache).to(DEV)
    b = synthetic.to_torch(synthetic.indoor_batch(scenes, points), DEV)
    batch = torch.repeat_interleave(torch.arange(scenes, device=DEV), torch.diff(b["offset"], prepend=b["offset"].new_zeros(1)))
    gc = b["grid_coord"]
    key = (batch << 48) | ((gc[:, 0] >> s) << 32) | ((gc[:, 1] >> s) << 16) | (gc[:, 2] >> s)
    uk = torch.unique(key)
    bb = uk >> 48
    cc = torch.stack([(uk >> 32) & 0xffff, (uk >> 16) & 0xffff, uk & 0xffff], 1)
    code = ops.serialize_encode(cc, bb, 8 - s, ("hilbert",))
    order, _ = ops.sort_keys(code, 0, 3 * (8 - s) + 3)
    cc, bb = cc[order[0]], bb[order[0]]
    ind = torch.cat([bb[:, None].int(), cc.int()], 1).contiguous()
    torch.save(ind.cpu(), cache)
    return ind


def main():
    s = 0 if CASE == "s0" else 1
    ind = stage_indices(s)
    n = ind.shape[0]
    info = {"case": CASE, "n": n}
    table = ops.HashTable(ind)
    nbr = ops.rulebook_subm(ind, 3, table)
    pairs = int((nbr >= 0).sum())
    info["pairs_k3"] = pairs
    # tile-level occupancy of the table: (16-row tile, tap) cells with at least one neighbour = MFMA work issued by the
    # tile-skipping kernels, vs pairs = useful work
    v = (nbr >= 0)
    nt = (n + 15) // 16
    vp = torch.zeros(27, nt * 16, dtype=torch.bool, device=DEV)
    vp[:, :n] = v
    info["tile16_tap_cells_nonempty"] = int(vp.view(27, nt, 16).any(2).sum())
    info["tile16_tap_cells"] = 27 * nt
    chans = (64, 32) if s == 0 else (64,)
    g = torch.Generator(device="cpu").manual_seed(0)
    for _ in range(ITERS):
        t2 = ops.HashTable(ind)
        ops.rulebook_subm(ind, 3, t2)
        if s == 0:
            ops.rulebook_subm(ind, 5, t2)
    for c in chans:
        x = torch.randn(n, c, generator=g).to(torch.bfloat16).to(DEV)
        w = (torch.randn(c, 27, c, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
        bias = torch.randn(c, generator=g).to(DEV)
        go = torch.randn(n, c, generator=g).to(torch.bfloat16).to(DEV)
        info[f"conv_c{c}"] = {"alg_bytes": n * c * 2 * 2 + 4 * 27 * n + 27 * c * c * 2, "alg_flops": 2.0 * pairs * c * c,
                              "gathered_bytes": pairs * c * 2}
        blk = ops.BlockTables(nbr)     # conv7's block tables (rulebook_blocks_kernel)
        info[f"conv_c{c}"]["halo_rows_mean"] = round(float(blk.hcnt.float().mean()), 1)
        info[f"conv_c{c}"]["alg_bytes_pair_list"] = n * c * 2 * 2 + 8 * pairs + 27 * c * c * 2      # SURVEY 8(d)'s formula
        for _ in range(ITERS):
            ops.spconv_fwd(x, w, bias, nbr)            # conv5: global gathers
            ops.spconv_fwd(x, w, bias, nbr, blk)       # conv7: register weights + DMA-staged halo
            ops.spconv_wgrad(x, go, nbr)
            ops.spconv_wgrad(x, go, nbr, blk=blk)      # wgrad7: accumulator-stationary, operands from the staged block images
        if s == 0 and c == 64:
            # round 6: SpUNet's widest level-0 shape on conv8 (block-staged, weights as register fragments) and on conv3 (global gathers)
            ci, co = 128, 96
            xw = torch.randn(n, ci, generator=g).to(torch.bfloat16).to(DEV)
            ww = (torch.randn(co, 27, ci, generator=g) * 0.03).to(torch.bfloat16).to(DEV)
            bw = torch.randn(co, generator=g).to(DEV)
            info["conv_wide_128_96"] = {"alg_flops": 2.0 * pairs * ci * co, "alg_bytes_pair_list": n * (ci + co) * 2 + 8 * pairs + 27 * ci * co * 2,
                                        "blocks": int(blk.hcnt.numel()), "blocks_beyond_416_rows": int(((blk.hcnt < 0) | (blk.hcnt > 416)).sum())}
            for _ in range(ITERS):
                ops.spconv_fwd(xw, ww, bw, nbr, blk)   # conv8 (+ its conv3 follow-up for the blocks beyond the image)
            os.environ["PTC_CONV8"] = "0"
            for _ in range(ITERS):
                ops.spconv_fwd(xw, ww, bw, nbr, blk)   # conv3
            os.environ["PTC_CONV8"] = "1"
            del xw, ww
        if s == 0:
            # the gather-fused qkv GEMM: kv = 1 table = a permutation (serialization order)
            perm = torch.randperm(n, generator=g).int().to(DEV)[None].contiguous()
            wq = (torch.randn(3 * c, 1, c, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
            bq = torch.randn(3 * c, generator=g).to(DEV)
            info[f"qkv_c{c}"] = {"alg_bytes": n * c * 2 + n * 3 * c * 2 + 4 * n}
            for _ in range(ITERS):
                ops.spconv_fwd(x, wq, bq, perm)
    torch.cuda.synchronize()
    print("CONVKERNELS " + json.dumps(info))


if __name__ == "__main__":
    main()
