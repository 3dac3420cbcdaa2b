#!/usr/bin/env python
"""Event-time the weight gradient of the 3^3 submanifold convolution at the bench's stage-0 / stage-1 shapes (rows in curve order, bf16):
wgrad2 (global gathers, 14 tap groups) against wgrad7 (block-staged, accumulator-stationary; csrc/wgrad7.h), each INCLUDING its
reduction launch, and the agreement of the two results."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pointcept_amd import ops  # noqa: E402
import conv_kernels  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(f, iters=20, warm=5):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for stage in (0, 1):
    ind = conv_kernels.stage_indices(stage)
    n = ind.shape[0]
    nbr = ops.rulebook_subm(ind, 3, ops.HashTable(ind))
    blk = ops.BlockTables(nbr)
    pairs = int((nbr >= 0).sum())
    g = torch.Generator(device="cpu").manual_seed(0)
    for c in ((64, 32) if stage == 0 else (64,)):
        x = torch.randn(n, c, generator=g).to(torch.bfloat16).to(DEV)
        go = torch.randn(n, c, generator=g).to(torch.bfloat16).to(DEV)
        t2 = timeit(lambda: ops.spconv_wgrad(x, go, nbr))
        t7 = timeit(lambda: ops.spconv_wgrad(x, go, nbr, blk=blk))
        a, b = ops.spconv_wgrad(x, go, nbr), ops.spconv_wgrad(x, go, nbr, blk=blk)
        rel = float((a - b).abs().max() / a.abs().max())
        nbytes = n * c * 2 * 2 + 8 * pairs + 27 * c * c * 4
        print(f"stage {stage} n={n} C={c}: wgrad2 {t2:7.1f} us   wgrad7 {t7:7.1f} us ({nbytes / t7 / 1e3:6.1f} GB/s by 8(d)'s bytes = {nbytes / t7 / 1e3 / 8000:.3f} of HBM; "
              f"{2.0 * pairs * c * c / t7 / 1e6:6.1f} TF/s)   max |diff| / max |dw| = {rel:.2e}   overflow blocks {int(blk.n_overflow)}", flush=True)
