#!/usr/bin/env python
"""Launch ONLY the roofline kernel of bench.py (attn_fwd_kernel at the dec0/enc0 shape) a few times.
Meant to run under rocprofv3 (`--kernel-trace --stats`, and separate `--pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` passes) so that the per-launch duration and HBM traffic of exactly that kernel
can be read from the profile; tools/pmc_summary.py turns the CSVs into profiles/*.json."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcept_amd import ops  # noqa: E402

scenes, points, L, H, D = 8, 102400, 1024, int(os.environ.get("PTC_ROOF_H", "4")), 16
n_seq = scenes * ((points + L - 1) // L)
T = n_seq * L
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
qkv = torch.randn(T, 3, H, D, generator=g).to(torch.bfloat16).to(dev)
cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=dev)
out, lse = ops.attn_varlen_fwd(qkv, cu, L, D ** -0.5)
do = torch.randn_like(out)
for _ in range(int(os.environ.get("PTC_ROOF_ITERS", "10"))):
    out, lse = ops.attn_varlen_fwd(qkv, cu, L, D ** -0.5)
    ops.attn_varlen_bwd(qkv, out, do, lse, cu, L, D ** -0.5)
torch.cuda.synchronize()
print("ok", n_seq, H)
