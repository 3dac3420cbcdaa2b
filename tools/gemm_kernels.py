#!/usr/bin/env python
"""Launch ONLY the deep-stage GEMM kernels (gemm3.h / wgrad3.h) at the shapes of a 256-channel PT-v3m1 Block with ~20000 rows, a few times
each, for rocprofv3 --kernel-trace / --pmc passes (tools/gpu_session.sh gemmpmc): qkv 256 -> 768, fc1 + GELU 256 -> 1024, fc2 1024 -> 256,
the weight gradient of fc1."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcept_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rows, c = int(os.environ.get("PTC_GK_ROWS", "20000")), int(os.environ.get("PTC_GK_C", "256"))
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(rows, c, generator=g).to(torch.bfloat16).to(dev)
h = torch.randn(rows, 4 * c, generator=g).to(torch.bfloat16).to(dev)
w_qkv = (torch.randn(3 * c, 1, c, generator=g) * 0.05).to(torch.bfloat16).to(dev)
b_qkv = torch.zeros(3 * c, device=dev)
w1 = (torch.randn(4 * c, c, generator=g) * 0.05).to(torch.bfloat16).to(dev)
b1 = torch.zeros(4 * c, device=dev)
w2 = (torch.randn(c, 1, 4 * c, generator=g) * 0.03).to(torch.bfloat16).to(dev)
b2 = torch.zeros(c, device=dev)
case = os.environ.get("PTC_GK_CASE", "all")
for _ in range(int(os.environ.get("PTC_GK_ITERS", "6"))):
    if case in ("all", "qkv"):
        ops.spconv_fwd(x, w_qkv, b_qkv, None)
    if case in ("all", "fc1"):
        ops.linear_gelu_fwd(x, w1, b1)
    if case in ("all", "fc2"):
        ops.spconv_fwd(h, w2, b2, None)
    if case in ("all", "wgrad"):
        ops.spconv_wgrad(x, h, None, want_bias=True)
torch.cuda.synchronize()
print("GEMMKERNELS", rows, c, case)
