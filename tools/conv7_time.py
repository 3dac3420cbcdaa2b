#!/usr/bin/env python
"""Time the block-staged convolution (conv7) alone at the bench's stage-0 shape (N = 8 x 102400 rows in curve order, 64 -> 64 and
32 -> 32, bf16) with the library variant named by PTC_LIB_VARIANT (timing ablations: `python -m pointcept_amd.build --variant
d_C7_ABLATE_<bits>`).  With no variant set and --all: runs itself once per libptcore_d_C7_ABLATE_*.so found and prints a table."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--all" in sys.argv:
    variants = [""] + sorted(os.path.basename(p)[len("libptcore_"):-3] for p in glob.glob(os.path.join(ROOT, "pointcept_amd", "libptcore_*.so")) if "hostprobe" not in p)
    for v in variants:
        env = dict(os.environ, PTC_LIB_VARIANT=v)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        print(f"{v or 'product':24s} " + ("\n".join(r.stdout.strip().splitlines()[-3:]) if r.stdout.strip() else r.stderr[-300:]), flush=True)
    sys.exit(0)

import torch  # noqa: E402

from pointcept_amd import ops  # noqa: E402
import conv_kernels  # noqa: E402

DEV = torch.device("cuda:0")
ind = conv_kernels.stage_indices(0)
n = ind.shape[0]
nbr = ops.rulebook_subm(ind, 3, ops.HashTable(ind))
blk = ops.BlockTables(nbr)
g = torch.Generator(device="cpu").manual_seed(0)
out = []
for c in (64, 32):
    x = torch.randn(n, c, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(c, 27, c, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    bias = torch.randn(c, generator=g).to(DEV)
    for _ in range(3):
        ops.spconv_fwd(x, w, bias, nbr, blk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.spconv_fwd(x, w, bias, nbr, blk)
    e1.record()
    torch.cuda.synchronize()
    out.append(f"C={c}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
    if "64" in os.environ.get("PTC_LIB_VARIANT", "").split("_")[-1:]:
        y = ops.spconv_fwd(x, w, bias, nbr, blk)
        torch.cuda.synchronize()
        ph = y[:256].contiguous().view(torch.int64).view(256, -1)[:, :8].double().cpu()      # [workgroup][phase] cycles of wave 0
        names = ["dma issue", "reset+first gathers", "tap loop", "acc->scratch", "wait dma/scratch", "barrier 1", "add+store", "barrier 2"]
        tot = ph.sum(1)
        print(f"C={c} phases of wave 0, cycles per block (mean over workgroups; 25 blocks each): " +
              ", ".join(f"{nm} {ph[:, i].mean() / 25:.0f}" for i, nm in enumerate(names)) + f"; sum {tot.mean() / 25:.0f} (min {tot.min() / 25:.0f}, max {tot.max() / 25:.0f})")
print(f"n={n}  " + "   ".join(out) + "   (conv7 + the empty conv5 launch behind it)")
