#!/usr/bin/env python
"""Golden vectors of LitePT-v1 (pointcept/models/litept/litept_v1.py), generated IN THE AUTHORING CONTAINER by importing the
reference's own model file (oracle/ref_import.py on oracle/shims.py) with `pointrope` = the reference's OWN pointrope_cpu compiled
from libs/pointrope/pointrope.cpp (oracle/build_ref.py): the file then uses its PointROPE_func / PointROPE classes (:27-59), the path
training runs.  /root/reference does not exist on the GPU box, the .npz travels.

    python tests/golden/make_golden_litept.py   ->  tests/golden/litept_tiny.npz
        the default stage layout (convolution blocks in stages 0-2, PointROPE attention blocks in stages 3-4, un-pooling decoder),
        36 / 72 / 72 / 144 / 144 channels with 18 per head (the reference's default is 36 / 72 / 144 / 252 / 504, 18 per head),
        two scenes (3000 + 900 voxels); eval features (every 8th row), train-mode loss and the gradient norm of every parameter,
        the state-dict key list.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import build_ref  # noqa: E402
from oracle import ptv3_model as om  # noqa: E402
from oracle import ref_import  # noqa: E402
from pointcept_amd import synthetic  # noqa: E402

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")
LITEPT_CFG = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 2, 1), enc_channels=(36, 72, 72, 144, 144), enc_num_head=(2, 4, 4, 8, 8),
                  enc_patch_size=(128,) * 5, dec_channels=(36, 72, 72, 144), dec_num_head=(2, 4, 4, 8), dec_patch_size=(128,) * 4,
                  drop_path=0.0, shuffle_orders=False)
SCENES = ((81, 3000), (82, 900))


def main():
    ref_import.load()
    sys.modules["pointrope"] = build_ref.load_pointrope()
    pkg = types.ModuleType("pointcept.models.litept")
    pkg.__path__ = [ref_import.REF + "/pointcept/models/litept"]
    sys.modules["pointcept.models.litept"] = pkg
    R = importlib.import_module("pointcept.models.litept.litept_v1")
    assert hasattr(R, "PointROPE_func")
    torch.manual_seed(0)
    ref = R.LitePT(**LITEPT_CFG)
    sd = om.deterministic_state_dict(ref, 43)
    ref.load_state_dict(sd)
    batch = synthetic.collate([synthetic.indoor_scene(s, n) for s, n in SCENES])
    inp = {k: torch.from_numpy(v) for k, v in batch.items()}
    inp["grid_size"] = 0.02
    ref.eval()
    torch.manual_seed(5)
    with torch.no_grad():
        out_eval = ref(dict(inp)).feat.numpy()
    ref.train()
    torch.manual_seed(5)
    f = ref(dict(inp)).feat
    loss = (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean()
    loss.backward()
    names = [k for k, _ in ref.named_parameters()]
    path = os.path.join(OUT, "litept_tiny.npz")
    np.savez_compressed(
        path, scene_seeds=np.asarray([s for s, _ in SCENES]), n_points=np.asarray([n for _, n in SCENES]),
        input_checksum=np.asarray([batch["grid_coord"].sum(), float(batch["feat"].astype(np.float64).sum())]),
        weight_checksum=np.asarray(float(sum(float(v.double().abs().sum()) for v in sd.values()))),
        state_keys=np.asarray(list(sd.keys())), feat_rows=out_eval[::8].astype(np.float32), feat_absmax=np.asarray(np.abs(out_eval).max()),
        loss=np.asarray(float(loss.detach())), grad_names=np.asarray(names),
        grad_norms=np.asarray([float(p.grad.norm()) for _, p in ref.named_parameters()], dtype=np.float64))
    print("written", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
