#!/usr/bin/env python
"""Stem convolution (SubM k = 5, 6 -> 32 channels, 8 x 102400 voxels in curve order): input padded to 16 channels on conv2 (round-1
form), padded to 8 on conv2, padded to 8 on conv3's four-table-rows-per-step form; and the weight gradient at 16 / 8 channels."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("PTC_CK_CASE", "s0")
import conv_kernels as ck  # noqa: E402
from pointcept_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


ind = ck.stage_indices(0)
n = ind.shape[0]
nbr = ops.rulebook_subm(ind, 5, ops.HashTable(ind))
pairs = int((nbr >= 0).sum())
g = torch.Generator().manual_seed(0)
x6 = torch.randn(n, 6, generator=g)
w6 = torch.randn(32, 125, 6, generator=g) * 0.05
go = torch.randn(n, 32, generator=g).to(torch.bfloat16).to(DEV)
res = {}
for pad in (16, 8):
    x = torch.nn.functional.pad(x6, (0, pad - 6)).to(torch.bfloat16).to(DEV)
    w = torch.nn.functional.pad(w6, (0, pad - 6)).to(torch.bfloat16).to(DEV)
    for off in (("1", "0") if pad == 8 else ("1",)):
        if off == "1":
            os.environ["PTC_CONV3_C8_OFF"] = "1"
        else:
            os.environ.pop("PTC_CONV3_C8_OFF", None)
        tag = f"pad{pad}_" + ("conv2" if off == "1" else "conv3x4")
        res[tag] = (timeit(lambda: ops.spconv_fwd(x, w, None, nbr)), ops.spconv_fwd(x, w, None, nbr).float())
    res[f"pad{pad}_wgrad"] = (timeit(lambda: ops.spconv_wgrad(x, go, nbr)), None)
os.environ.pop("PTC_CONV3_C8_OFF", None)
ref = res["pad16_conv2"][1]
print(f"stem n={n} pairs={pairs} ({pairs / n:.1f} per voxel of 125)")
for k, (t, o) in res.items():
    d = "" if o is None else f"  max|diff| vs pad16_conv2 {float((o - ref).abs().max()):.3e}"
    print(f"  {k:16s} {t:8.1f} us{d}")
