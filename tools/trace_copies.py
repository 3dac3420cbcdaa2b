#!/usr/bin/env python
"""Which engine source lines launch the small ATen kernels of one training step?  A TorchDispatchMode logs every
aten op outside a short allow-list together with the innermost pointcept_amd / bench.py frame of the Python stack
(forward AND backward: custom Function.backward bodies run as Python on the autograd thread).  Complements
tools/trace_step.py, whose profiler stacks are empty for device ops on this build.

    python tools/trace_copies.py > gpurun_out/trace_copies.txt
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pointcept_amd import synthetic  # noqa: E402
from pointcept_amd.point_transformer_v3 import PointTransformerV3  # noqa: E402
from pointcept_amd.segmentor import DefaultSegmentorV2  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
SPUNET = "--model" in sys.argv and sys.argv[sys.argv.index("--model") + 1] == "spunet"
if SPUNET:       # BASELINE configs[1]: SpUNet-v1m1 + CE + SGD, 8 x 100000 voxels (bench.build_spunet)
    from pointcept_amd import functional as PF
    from pointcept_amd.sparse_unet import SpUNetBase

    net = SpUNetBase(6, 20, **bench.SPUNET_BASE).to(dev).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
    batch = synthetic.to_torch(synthetic.indoor_batch(8, 100000, rank=0), dev)

    class _M(torch.nn.Module):
        def forward(self, d):
            return {"loss": PF.cross_entropy(net(d), d["segment"], -1)}

    model = _M()
else:
    model = DefaultSegmentorV2(20, 64, PointTransformerV3(**bench.PTV3_BASE), criteria=("ce", "lovasz")).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)
    batch = synthetic.to_torch(synthetic.indoor_batch(8, 102400, rank=0), dev)
SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::t", "aten::transpose", "aten::permute", "aten::slice", "aten::select",
        "aten::detach", "aten::alias", "aten::expand", "aten::as_strided", "aten::unsqueeze", "aten::squeeze", "aten::empty", "aten::size",
        "aten::stride", "aten::is_", "aten::sym_", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::unbind", "aten::split",
        "aten::chunk", "aten::narrow", "aten::new_empty", "aten::empty_like", "aten::empty_strided", "aten::result_type", "aten::item")


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(dict(batch))["loss"]
    loss.backward()
    opt.step()


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.defaultdict(lambda: [0, 0])
        self.shapes = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.name() if hasattr(func, "name") else str(func)
        if not name.startswith(SKIP):
            frame = "?"
            for fs in reversed(traceback.extract_stack(limit=40)):
                if "pointcept_amd" in fs.filename or fs.filename.endswith("bench.py"):
                    frame = f"{os.path.relpath(fs.filename, ROOT)}:{fs.lineno} {fs.name}"
                    break
            n = 0
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    n += t.numel() * t.element_size()
            if n > 0 or name.startswith(("aten::copy_", "aten::fill_", "aten::zero_")):
                e = self.rows[(name, frame)]
                e[0] += 1
                e[1] += n
                if n >= (4 << 20):
                    ins = [f"{tuple(a.shape)} {str(a.dtype)[6:]}" for a in args if isinstance(a, torch.Tensor)]
                    o = out[0] if isinstance(out, (tuple, list)) else out
                    self.shapes[(name, frame)][" , ".join(ins) + f" -> {str(o.dtype)[6:]}"] += 1
        return out


for _ in range(2):
    step()
torch.cuda.synchronize()
log = Log()
with log:
    step()
torch.cuda.synchronize()
tot = collections.defaultdict(lambda: [0, 0])
for (name, frame), (c, b) in log.rows.items():
    tot[name][0] += c
    tot[name][1] += b
print("== aten ops of one step (count, output MB)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{v[0]:6d} {v[1] / 1e6:10.1f} MB  {k}")
print("== by (op, innermost engine frame), sorted by count")
for (name, frame), (c, b) in sorted(log.rows.items(), key=lambda kv: -kv[1][0])[:120]:
    print(f"{c:6d} {b / 1e6:10.1f} MB  {name:34s} {frame}")

print("== operands of the ops that write >= 4 MB (count x inputs -> output dtype), by site")
for (name, frame), ctr in sorted(log.shapes.items(), key=lambda kv: -log.rows[kv[0]][1]):
    print(f"{name:28s} {frame}   [{log.rows[(name, frame)][1] / 1e6:.1f} MB]")
    for sh, c in ctr.most_common(12):
        print(f"      {c:3d} x {sh}")
