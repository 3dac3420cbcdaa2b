#!/usr/bin/env python
"""Per-module forward comparison engine (GPU) vs oracle (CPU) on a small batch: prints the relative error of
point.feat after every encoder / decoder sub-module (run with PTC_SORT_POINTS=0 so rows line up at stage 0).

    PTC_SORT_POINTS=0 python tools/debug_stage_diff.py [rpe|dense|flash]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ptv3_model as om  # noqa: E402
from pointcept_amd import synthetic  # noqa: E402
from pointcept_amd.point_transformer_v3 import PointTransformerV3  # noqa: E402

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")
mode = sys.argv[1] if len(sys.argv) > 1 else "dense"
cfg = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 1, 1), dec_depths=(1, 1, 1, 1), enc_patch_size=(256,) * 5,
           dec_patch_size=(256,) * 4, drop_path=0.0, shuffle_orders=False)
if mode in ("dense", "rpe"):
    cfg.update(enable_flash=False, upcast_attention=True, upcast_softmax=True, enable_rpe=mode == "rpe")
torch.manual_seed(0)
orc, eng = om.PointTransformerV3(**cfg), PointTransformerV3(**cfg)
sd = om.deterministic_state_dict(orc, 2)
orc.load_state_dict(sd)
eng.load_state_dict(sd)
eng = eng.cuda().eval()
orc.eval()
batch = synthetic.collate([synthetic.indoor_scene(23, 900), synthetic.indoor_scene(24, 200)])
cap = {"o": {}, "e": {}}


def hook(tag, name):
    def f(mod, inp, out):
        feat = out.feat if hasattr(out, "feat") else (out["feat"] if isinstance(out, dict) else out)
        feat = getattr(feat, "features", feat)
        if not torch.is_tensor(feat):
            return
        extra = getattr(mod, "patch_size", None)
        cap[tag][name] = (feat.detach().float().cpu(), extra)
    return f


pre = {"o": {}, "e": {}}


def pre_hook(tag, name):
    def f(mod, inp):
        pt = inp[0]
        d = {"feat_in": pt["feat"].detach().float().cpu()}
        for key in ("pad", "unpad", "grid_coord", "offset", "batch"):
            if key in pt.keys():
                d[key] = pt[key].detach().cpu()
        d["order0"] = pt["serialized_order"][0].detach().cpu()
        pre[tag][name] = d
    return f


for tag, net in (("o", orc), ("e", eng)):
    for name, m in net.named_modules():
        if name.endswith(".attn"):
            m.register_forward_pre_hook(pre_hook(tag, name))
for tag, net in (("o", orc), ("e", eng)):
    for name, m in net.named_modules():
        if name and name.count(".") <= 3 and not name.split(".")[-1].isdigit():
            m.register_forward_hook(hook(tag, name))
with torch.no_grad():
    torch.manual_seed(5)   # pooling shuffles the order rows with the CPU generator
    orc({k: torch.from_numpy(v) for k, v in batch.items()})
    torch.manual_seed(5)
    eng(synthetic.to_torch(batch, "cuda"))
for name, (fo, ko) in cap["o"].items():
    if name not in cap["e"]:
        print(f"{name:40s} (no engine capture)")
        continue
    fe, ke = cap["e"][name]
    if fe.shape != fo.shape:
        print(f"{name:40s} shape {tuple(fe.shape)} vs {tuple(fo.shape)}")
        continue
    rel = float((fe - fo).abs().max() / fo.abs().max().clamp(min=1e-12))
    print(f"{name:40s} rel {rel:9.3e}  rows {fo.shape[0]:5d}  K eng/orc {ke}/{ko}")

print("---- attention inputs (pre-hook) ----")
for name, do in pre["o"].items():
    de = pre["e"].get(name)
    if de is None:
        continue
    msg = []
    for key, vo in do.items():
        ve = de.get(key)
        if ve is None:
            msg.append(f"{key}: missing")
        elif ve.shape != vo.shape:
            msg.append(f"{key}: shape {tuple(ve.shape)} vs {tuple(vo.shape)}")
        elif vo.is_floating_point():
            msg.append(f"{key}: rel {float((ve - vo).abs().max() / vo.abs().max().clamp(min=1e-12)):.2e}")
        else:
            msg.append(f"{key}: {'eq' if torch.equal(ve.long(), vo.long()) else 'DIFF'}")
    print(f"{name:28s} " + " | ".join(msg))
