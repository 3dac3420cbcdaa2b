#!/usr/bin/env python
"""TEMPORARY experiment driver: conv7 with parts switched off (PTC_CONV7_DBG bits: 1 no DMA of the next block, 2 no MFMA loop, 4 no epilogue)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pointcept_amd import ops
import conv_kernels as ck
DEV = torch.device("cuda:0")
ind = ck.stage_indices(0)
n = ind.shape[0]
nbr = ops.rulebook_subm(ind, 3, ops.HashTable(ind))
blk = ops.BlockTables(nbr)
def timeit(fn, iters=20, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for c in (64, 32):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, c, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(c, 27, c, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    bias = torch.randn(c, generator=g).to(DEV)
    for dbg in (0, 1, 2, 4, 3, 6, 5, 7):
        os.environ["PTC_CONV7_DBG"] = str(dbg)
        print(f"C={c} dbg={dbg} (1: no DMA, 2: no MFMA loop, 4: no epilogue): {timeit(lambda: ops.spconv_fwd(x, w, bias, nbr, blk)):8.1f} us", flush=True)
    os.environ.pop("PTC_CONV7_DBG")
