"""Generate the golden fixtures under tests/golden/ by running the REFERENCE'S OWN code
(/root/reference, imported unmodified through oracle/ref_import.py).  Only runnable in the
authoring container; the .npz outputs are committed and travel to the GPU box.

    python tests/golden/make_golden.py

Fixtures
  serialization.npz : grid_coord / batch / depth -> serialization.encode codes for the 4 orders
                      (pointcept/models/utils/serialization/default.py:8-24)
  padmaps.npz       : offsets, K -> SerializedAttention.get_padding_and_inverse (ptv3m1:114-170)
  pooling.npz       : Point -> SerializedPooling.forward maps (ptv3m1:371-444), shuffle off
  ptv3_tiny.npz     : BASELINE config 1 (PTv3 depths 1/1/1/1/1 + 1/1/1/1, one 8192-voxel scene):
                      backbone output of the reference model running its FLASH branch on the CPU
                      stand-ins of oracle/shims.py, deterministic weights (oracle.ptv3_model.
                      deterministic_state_dict), CPU RNG seeded with 5 before the forward.
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
TINY_CFG = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 1, 1), dec_depths=(1, 1, 1, 1),
                enc_patch_size=(1024,) * 5, dec_patch_size=(1024,) * 4, drop_path=0.0, shuffle_orders=False)


def main():
    R = ref_import.load()
    ser, ptv3 = R["serialization"], R["ptv3"]
    RefPoint = R["structure"].Point

    # ---- serialization -------------------------------------------------------------------------
    rng = np.random.default_rng(2024)
    blobs = {}
    for depth in (1, 2, 5, 8, 9, 13, 16):
        n = 300
        gc = rng.integers(0, 1 << depth, size=(n, 3), dtype=np.int64)
        b = rng.integers(0, 5, size=n, dtype=np.int64)
        code = torch.stack([ser.encode(torch.from_numpy(gc), torch.from_numpy(b), depth, o) for o in ORDERS]).numpy()
        blobs[f"gc_{depth}"], blobs[f"batch_{depth}"], blobs[f"code_{depth}"] = gc, b, code
    np.savez_compressed(os.path.join(OUT, "serialization.npz"), **blobs)

    # ---- pad maps ------------------------------------------------------------------------------
    blobs = {}
    cases = [([10, 3, 7], 4), ([1024, 1025, 5000, 1], 1024), ([48, 49, 100, 7], 48), ([330, 1425, 2048], 1024),
             ([2047, 2049, 1023, 1024, 1], 1024), ([5], 1024), ([129, 128, 127, 256, 1000], 128)]
    for ci, (counts, K) in enumerate(cases):
        attn = ptv3.SerializedAttention(channels=16, num_heads=1, patch_size=K, enable_flash=True,
                                        upcast_attention=False, upcast_softmax=False)
        p = RefPoint(offset=torch.tensor(np.cumsum(counts)))
        pad, unpad, cu = attn.get_padding_and_inverse(p)
        blobs[f"counts_{ci}"] = np.asarray(counts, dtype=np.int64)
        blobs[f"K_{ci}"] = np.asarray(K)
        blobs[f"pad_{ci}"], blobs[f"unpad_{ci}"], blobs[f"cu_{ci}"] = pad.numpy(), unpad.numpy(), cu.numpy()
    blobs["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(OUT, "padmaps.npz"), **blobs)

    # ---- pooling maps --------------------------------------------------------------------------
    b = synthetic.collate([synthetic.indoor_scene(11, 1500), synthetic.indoor_scene(12, 900)])
    p = RefPoint({k: torch.from_numpy(v) for k, v in b.items()})
    p.serialization(order=ORDERS, shuffle_orders=False)
    import torch.nn as nn
    from functools import partial
    pool = ptv3.SerializedPooling(6, 8, stride=2, norm_layer=partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01),
                                  act_layer=nn.GELU, shuffle_orders=False)
    child = pool(p)
    np.savez_compressed(
        os.path.join(OUT, "pooling.npz"), grid_coord=b["grid_coord"], offset=b["offset"],
        parent_code=p.serialized_code.numpy(), parent_depth=np.asarray(p.serialized_depth),
        cluster=child.pooling_inverse.numpy(), child_code=child.serialized_code.numpy(),
        child_order=child.serialized_order.numpy(), child_inverse=child.serialized_inverse.numpy(),
        child_grid_coord=child.grid_coord.numpy(), child_batch=child.batch.numpy(),
        child_coord=child.coord.numpy(), child_depth=np.asarray(child.serialized_depth))

    # ---- PTv3 tiny (BASELINE config 1) ---------------------------------------------------------
    torch.manual_seed(0)
    ref = ptv3.PointTransformerV3(enable_flash=True, **TINY_CFG)
    sd = om.deterministic_state_dict(ref, 0)
    ref.load_state_dict(sd)
    ref.eval()
    scene = synthetic.collate([synthetic.indoor_scene(7, 8192)])
    inp = {k: torch.from_numpy(v) for k, v in scene.items()}
    torch.manual_seed(5)
    with torch.no_grad():
        out = ref(inp).feat.numpy()
    key_sum = float(sum(float(v.double().abs().sum()) for v in sd.values()))
    np.savez_compressed(
        os.path.join(OUT, "ptv3_tiny.npz"), scene_seed=np.asarray(7), n_points=np.asarray(8192),
        input_checksum=np.asarray([scene["grid_coord"].sum(), float(scene["feat"].astype(np.float64).sum())]),
        weight_checksum=np.asarray(key_sum), feat_rows=out[::16].astype(np.float32),
        feat_row_norm=np.linalg.norm(out.astype(np.float64), axis=1).astype(np.float32),
        feat_absmax=np.asarray(np.abs(out).max()))
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
