#!/usr/bin/env python
"""bisect: SpUNet stem conv under fp16 autocast (tools/spunet_amp_probe.py saw rel err 31)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ptv3_model as om
from pointcept_amd import synthetic, ops, functional as PF
from pointcept_amd import spconv_api as spconv
from pointcept_amd.structure import offset2batch

dev = torch.device("cuda:0")
order = sys.argv[1] if len(sys.argv) > 1 else "f16,bf16,f16"
batch = synthetic.to_torch(synthetic.collate([synthetic.indoor_scene(61, 20000)]), dev)
conv = spconv.SubMConv3d(6, 32, kernel_size=5, padding=1, bias=False, indice_key="stem").to(dev)
torch.manual_seed(0)
with torch.no_grad():
    conv.weight.normal_(0, 0.05)
ind = torch.cat([offset2batch(batch["offset"]).unsqueeze(-1).int(), batch["grid_coord"].int()], 1).contiguous()
shape = (batch["grid_coord"].max(0).values + 96).tolist()
def run(dt):
    x = spconv.SparseConvTensor(batch["feat"], ind, shape, 1)
    with torch.autocast("cuda", dtype=dt or torch.bfloat16, enabled=dt is not None):
        return conv(x).features.float()
ref = run(None)
print("fp32 absmax", float(ref.abs().max()))
for name in order.split(","):
    dt = {"f16": torch.float16, "bf16": torch.bfloat16}[name]
    y = run(dt)
    print(name, "rel", float((y - ref).norm() / ref.norm()))
    with torch.no_grad():
        conv.weight.mul_(1.0)       # version bump: shadows stale
# direct kernel call
rb = ops.rulebook_subm(ind, 5, None) if False else None
