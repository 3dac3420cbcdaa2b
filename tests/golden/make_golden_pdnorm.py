#!/usr/bin/env python
"""Golden vectors of the PPT configuration of PT-v3m1 (prompt-driven normalisation: pdnorm_bn + pdnorm_ln, decoupled, adaptive;
pointcept/models/point_prompt_training/prompt_driven_normalization.py, configs/*/semseg-pt-v3m1-*-ppt-*.py), generated IN THE
AUTHORING CONTAINER by importing the reference's own model file (oracle/ref_import.py on oracle/shims.py); /root/reference does not
exist on the GPU box, the .npz travels.

    python tests/golden/make_golden_pdnorm.py   ->  tests/golden/ptv3_pdnorm_tiny.npz
        two scenes (2000 + 600 voxels), condition "S3DIS" of ("ScanNet", "S3DIS", "Structured3D"), context [1, 256] seeded;
        eval features (every 4th row), train-mode features, loss = mean((feat * ramp)^2), gradient norm of every parameter
        (-1 = no gradient: the norm layers of the conditions that were not selected), the state-dict key list.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import ptv3_model as om  # noqa: E402
from oracle import ref_import  # noqa: E402
from pointcept_amd import synthetic  # noqa: E402

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")
PDNORM_CFG = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 1, 1), dec_depths=(1, 1, 1, 1), enc_patch_size=(1024,) * 5,
                  dec_patch_size=(1024,) * 4, drop_path=0.0, shuffle_orders=False, pdnorm_bn=True, pdnorm_ln=True,
                  pdnorm_decouple=True, pdnorm_adaptive=True, pdnorm_conditions=("ScanNet", "S3DIS", "Structured3D"))
SCENES = ((61, 2000), (62, 600))


def inputs():
    batch = synthetic.collate([synthetic.indoor_scene(s, n) for s, n in SCENES])
    inp = {k: torch.from_numpy(v) for k, v in batch.items()}
    inp["condition"] = "S3DIS"
    inp["context"] = torch.randn(1, 256, generator=torch.Generator().manual_seed(5))
    return batch, inp


def main():
    R = ref_import.load()
    torch.manual_seed(0)
    ref = R["ptv3"].PointTransformerV3(**PDNORM_CFG)
    ref.load_state_dict(om.deterministic_state_dict(ref, 23))
    batch, inp = inputs()
    ref.eval()
    torch.manual_seed(5)
    with torch.no_grad():
        feat_eval = ref(dict(inp)).feat.numpy()
    ref.train()
    torch.manual_seed(6)
    feat = ref(dict(inp)).feat
    loss = (feat * torch.linspace(-1, 1, feat.shape[1])).pow(2).mean()
    loss.backward()
    names = [k for k, _ in ref.named_parameters()]
    np.savez_compressed(
        os.path.join(OUT, "ptv3_pdnorm_tiny.npz"), input_checksum=np.asarray([batch["grid_coord"].sum()]),
        feat_eval_rows=feat_eval[::4].astype(np.float32), feat_absmax=np.asarray(float(np.abs(feat_eval).max())),
        feat_train_rows=feat.detach().numpy()[::4].astype(np.float32), loss=np.asarray(float(loss.detach())),
        param_names=np.asarray(names), state_keys=np.asarray(list(ref.state_dict().keys())),
        grad_norms=np.asarray([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in ref.named_parameters()]))
    print("written", os.path.join(OUT, "ptv3_pdnorm_tiny.npz"), os.path.getsize(os.path.join(OUT, "ptv3_pdnorm_tiny.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
