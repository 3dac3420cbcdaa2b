#!/usr/bin/env python
"""Where does a 16-bit autocast run of SpUNet leave the fp32 run?  Per-module relative Frobenius distance of the output features
(forward hooks), engine vs engine, on one indoor scene.  usage: python tools/spunet_amp_probe.py [n_voxels]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic
    from pointcept_amd.sparse_unet import SpUNetBase

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dev = torch.device("cuda:0")
    kw = dict(channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2))
    torch.manual_seed(0)
    eng = SpUNetBase(6, 20, **kw)
    sd = om.deterministic_state_dict(eng, 6)
    eng.load_state_dict(sd)
    eng = eng.to(dev).train()
    batch = synthetic.to_torch(synthetic.collate([synthetic.indoor_scene(61, n)]), dev)
    recs = {}

    def hook(name, store):
        def fn(m, inp, out):
            f = out.features if hasattr(out, "features") else out
            if torch.is_tensor(f):
                store[name] = f.detach().float()
        return fn

    outs = {}
    for mode, dtype in (("fp32", None), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
        eng.load_state_dict(sd)
        store = {}
        hs = [m.register_forward_hook(hook(k, store)) for k, m in eng.named_modules() if k and not list(m.children())]
        with torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None):
            eng(dict(batch))
        for h in hs:
            h.remove()
        outs[mode] = store
    print(f"{'module':40s} {'absmax fp32':>12s} {'bf16 rel':>10s} {'fp16 rel':>10s}")
    for k, ref in outs["fp32"].items():
        r = [float((outs[m][k] - ref).norm() / ref.norm().clamp(min=1e-30)) if k in outs[m] else float("nan") for m in ("bf16", "fp16")]
        print(f"{k:40s} {float(ref.abs().max()):12.4e} {r[0]:10.3e} {r[1]:10.3e} {'nonfinite' if not torch.isfinite(outs['fp16'].get(k, ref)).all() else ''}")


if __name__ == "__main__":
    main()
