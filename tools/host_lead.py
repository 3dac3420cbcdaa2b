#!/usr/bin/env python
"""How far does the host run ahead of the GPU in the bench step?  Enqueue time (step() returns, no sync) vs completed time per step:
if the two are close the step is host-bound and 8 ranks sharing one host will scale worse than the GPU work suggests."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=8)
ap.add_argument("--points", type=int, default=102400)
mine = ap.parse_args()
sys.argv = [sys.argv[0], "--batch", str(mine.scenes), "--points", str(mine.points)]
args = bench.parse()
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model, opt, batch, loss_of = bench.build_ptv3(args, dev, 0)
step = bench.make_step(model, opt, batch, args.amp, loss_of, dev)
for _ in range(3):
    step()
torch.cuda.synchronize()
enq, tot = [], []
ms0 = torch.cuda.memory_stats()
for _ in range(8):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
from pointcept_amd import config  # noqa: E402

ms1 = torch.cuda.memory_stats()
# device allocations / frees inside the timed steps: a hipFree synchronises the device (a step whose allocator keeps growing and
# trimming shows up here, not in the kernel profile)
print(f"allocator over the 8 timed steps: {ms1['num_device_alloc'] - ms0['num_device_alloc']} device allocations, "
      f"{ms1['num_device_free'] - ms0['num_device_free']} device frees, {ms1['num_alloc_retries'] - ms0['num_alloc_retries']} retries; "
      f"reserved {ms1['reserved_bytes.all.current'] / 2**30:.2f} GiB, peak allocated {ms1['allocated_bytes.all.peak'] / 2**30:.2f} GiB")

print(f"{mine.scenes} x {mine.points} voxels, block executor {'on' if config.EXEC_BLOCK else 'off'}: ", end="")
print(f"host enqueue per step: mean {sum(enq) / len(enq):.1f} ms (min {min(enq):.1f}); step complete: mean {sum(tot) / len(tot):.1f} ms; "
      f"host lead {sum(tot) / len(tot) - sum(enq) / len(enq):.1f} ms; host syncs inside the step make enqueue >= the GPU time up to the last sync")
