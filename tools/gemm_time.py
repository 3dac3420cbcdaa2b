#!/usr/bin/env python
"""Event-timed launches of the deep-stage GEMMs (gemm3.h / wgrad3.h) at the shapes of PT-v3m1's 128 / 256 / 512-channel Blocks.  One line per
shape: us per launch, TFLOP/s.  PTC_LIB_VARIANT selects an ablation build (d_G3_ABLATE_n: wrong results, timing only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pointcept_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
print("variant", os.environ.get("PTC_LIB_VARIANT", "") or "default", "PTC_GEMM3", os.environ.get("PTC_GEMM3", "1"))
shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ.get("PTC_GT_SHAPES", "68000x128,20000x256,5500x512").split(",")]
for rows, c in shapes:
    x = torch.randn(rows, c, generator=g).to(torch.bfloat16).to(dev)
    h = torch.randn(rows, 4 * c, generator=g).to(torch.bfloat16).to(dev)
    for name, cin, cout, inp in (("proj", c, c, x), ("qkv", c, 3 * c, x), ("fc2", 4 * c, c, h)):
        w = (torch.randn(cout, 1, cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        b = torch.zeros(cout, device=dev)
        ms = bench._time_launches(lambda: ops.spconv_fwd(inp, w, b, None), iters=20, warm=5)
        fl = 2.0 * rows * cin * cout
        print(f"  N={rows:6d} {name:5s} {cin:4d}->{cout:4d}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    w1 = (torch.randn(4 * c, c, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b1 = torch.zeros(4 * c, device=dev)
    ms = bench._time_launches(lambda: ops.linear_gelu_fwd(x, w1, b1), iters=20, warm=5)
    print(f"  N={rows:6d} fc1+G {c:4d}->{4 * c:4d}: {ms * 1e3:7.1f} us  {2.0 * rows * c * 4 * c / ms / 1e9:7.1f} TFLOP/s")
    ms = bench._time_launches(lambda: ops.spconv_wgrad(x, h, None, want_bias=True), iters=20, warm=5)
    print(f"  N={rows:6d} wgrad {c:4d}->{4 * c:4d}: {ms * 1e3:7.1f} us  {2.0 * rows * c * 4 * c / ms / 1e9:7.1f} TFLOP/s (kernel + reduction)")
