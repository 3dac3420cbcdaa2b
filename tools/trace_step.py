#!/usr/bin/env python
"""Attribute the small ATen kernels of one training step (copies, fills, casts, index ops) to the engine's
Python source lines: torch.profiler with stacks over ONE step of bench.py's PTv3 workload.
    python tools/trace_step.py > gpurun_out/trace_step.txt
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pointcept_amd import synthetic  # noqa: E402
from pointcept_amd.point_transformer_v3 import PointTransformerV3  # noqa: E402
from pointcept_amd.segmentor import DefaultSegmentorV2  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = DefaultSegmentorV2(20, 64, PointTransformerV3(**bench.PTV3_BASE)).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)
batch = synthetic.to_torch(synthetic.indoor_batch(8, 102400, rank=0), dev)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(dict(batch))["loss"]
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

def dev_time(e):
    for name in ("self_device_time_total", "self_cuda_time_total"):
        if hasattr(e, name):
            return getattr(e, name)
    return 0.0


rows = []
for e in prof.key_averages(group_by_stack_n=12):
    t = dev_time(e)
    if t <= 0 or not e.key.startswith("aten::"):
        continue
    frame = "?"
    for fr in e.stack or []:
        if "pointcept_amd" in fr or "bench.py" in fr:
            frame = fr.strip().replace(ROOT + "/", "")
            break
    rows.append((t, e.count, e.key, frame))
tot = collections.defaultdict(lambda: [0, 0.0])
for t, cnt, key, frame in rows:
    tot[key][0] += cnt
    tot[key][1] += t
print("== ATen ops by name (self device ms, count)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{v[1] / 1e3:8.3f} ms {v[0]:5d}  {k}")
print("== by (op, first engine frame)")
agg = collections.defaultdict(lambda: [0, 0.0])
for t, cnt, key, frame in rows:
    agg[(key, frame)][0] += cnt
    agg[(key, frame)][1] += t
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
    print(f"{v[1] / 1e3:8.3f} ms {v[0]:5d}  {k[0]:26s} {k[1][:130]}")
