#!/usr/bin/env python
"""Which call sites launch the small fill / copy kernels of a step?  torch.profiler with stacks over two steps of the bench model,
aten::fill_ / zero_ / zeros / copy_ / clone events grouped by (op, shape, nearest frame of this repo or autograd node)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = [sys.argv[0], "--batch", "2", "--points", "20000"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model, opt, batch, loss_of = bench.build_ptv3(args, dev, 0)
step = bench.make_step(model, opt, batch, args.amp, loss_of, dev)
for _ in range(4):
    step()
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
WANT = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy",
        "aten::full", "aten::ones")
groups = collections.Counter()
for ev in prof.events():
    if ev.name not in WANT:
        continue
    frame = "?"
    for fr in ev.stack or []:
        if "pointcept_amd" in fr or "bench.py" in fr or "autograd" in fr or "optim" in fr:
            frame = fr.strip()[-110:]
            break
    parent = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
    groups[(ev.name, str(ev.input_shapes)[:60], parent[:50], frame)] += 1
print(f"per step (over {STEPS} steps), events with >= 2 per step:")
for (name, shapes, parent, frame), cnt in sorted(groups.items(), key=lambda kv: -kv[1]):
    if cnt / STEPS >= 2:
        print(f"{cnt / STEPS:7.1f}  {name:18s} {shapes:60s} parent={parent:50s} {frame}")
