#!/usr/bin/env python
"""Launch ONLY linear2_kernel (nn.Linear on [N, C] rows) at N = 819200 for a few (c_in, c_out) pairs, a few times each, for
rocprofv3 --kernel-trace / --pmc passes (tools/gpu_session.sh linpmc): which resource separates the 0.75-of-HBM shapes (32 -> 128)
from the 0.45-0.55 ones (64 -> 128, 64 -> 256)?  One shape per process: PTC_LK_SHAPE=cin,cout."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcept_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
cin, cout = (int(v) for v in os.environ.get("PTC_LK_SHAPE", "64,128").split(","))
n = 819200
g = torch.Generator().manual_seed(0)
x = torch.randn(n, cin, generator=g).to(torch.bfloat16).to(DEV)
w = (torch.randn(cout, 1, cin, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
b = torch.randn(cout, generator=g).to(DEV)
for _ in range(int(os.environ.get("PTC_LK_ITERS", "6"))):
    ops.spconv_fwd(x, w, b, None)
torch.cuda.synchronize()
print("LINEARKERNELS", cin, cout, n)
