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
  spunet_tiny.npz   : SpUNet-v1m1 (the reference file spconv_unet_v1m1_base.py on oracle/shims.py), channels
                      (16,32,48,64,64,48,32,32), layers (1,2,1,1,1,1,2,1), two scenes (3000 + 1200 voxels):
                      eval-mode and train-mode logits (every 4th row), CE loss and the gradient norm of every parameter.
  ptv3_rpe.npz      : the reference's dense attention branch (enable_flash=False, enable_rpe=True, both upcasts), patch
                      256 on scenes of 900 + 200 voxels (=> K shrinks to the smallest scene at every stage):
                      eval output, train-mode output rows, loss = mean(feat^2) and all gradient norms.
  gridsample.npz    : coord -> GridSample(hash_type="fnv", mode="train") of pointcept/datasets/transform.py: inverse,
                      the voxel set in np.unique (ascending key) order, min_coord and the picked representatives.
  ptv3_enc_mode.npz : reference PT-v3m1 with enc_mode=True followed by the parent-chain concatenation of
                      DefaultSegmentorV2.forward (default.py:69-74): [N, 32+64+128+256+512] features (every 32nd row, column norms).
  ptv3m2_tiny.npz   : the reference's point_transformer_v3m2_sonata.py (GridPooling / GridUnpooling / LayerScale / Linear stem) on
                      oracle/shims.py: depths 1/1/1/2/1 + 1/1/1/1, layer_scale 0.5, two scenes (2500 + 700 voxels): eval features
                      (every 8th row), train-mode loss and the gradient norm of every parameter, the state-dict key list.
  spherecrop.npz    : coord / segment -> SphereCrop(point_max, mode) of pointcept/datasets/transform.py:1014-1057, three cases
                      (mode "center" twice, mode "random" with numpy's generator seeded and the drawn centre recorded).
  pointrope.npz     : tokens / positions -> pointrope_cpu of libs/pointrope/pointrope.cpp:13-49 (compiled from the reference tree by
                      oracle/build_ref.py), four shapes (D = 18, 24, 48, 6; F0 = +-1).
  lovasz.npz        : logits / labels -> LovaszLoss(mode="multiclass", ignore_index=-1) loss and gradient
                      (pointcept/models/losses/lovasz.py), five shapes incl. absent classes and a single point.
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


RPE_CFG = dict(TINY_CFG, enc_patch_size=(256,) * 5, dec_patch_size=(256,) * 4, enable_flash=False, enable_rpe=True,
               upcast_attention=True, upcast_softmax=True)
ENC_CFG = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 1, 1), enc_channels=(32, 64, 128, 256, 512),
               enc_num_head=(2, 4, 8, 16, 32), enc_patch_size=(128,) * 5, drop_path=0.0, shuffle_orders=False)
M2_CFG = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 2, 1), enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32),
              dec_depths=(1, 1, 1, 1), dec_channels=(64, 64, 128, 256), dec_num_head=(4, 4, 8, 16), enc_patch_size=(128,) * 5,
              dec_patch_size=(128,) * 4, drop_path=0.0, shuffle_orders=False, layer_scale=0.5)
SPUNET_CFG = dict(base_channels=16, channels=(16, 32, 48, 64, 64, 48, 32, 32), layers=(1, 2, 1, 1, 1, 1, 2, 1))


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
    # ---- SpUNet tiny ---------------------------------------------------------------------------
    spunet = R["spunet"]
    torch.manual_seed(0)
    ref = spunet.SpUNetBase(6, 20, **SPUNET_CFG)
    sd = om.deterministic_state_dict(ref, 1)
    ref.load_state_dict(sd)
    batch = synthetic.collate([synthetic.indoor_scene(31, 3000), synthetic.indoor_scene(32, 1200)])
    inp = {k: torch.from_numpy(v) for k, v in batch.items()}
    ref.eval()
    with torch.no_grad():
        logits_eval = ref(inp).numpy()
    ref.train()
    logits = ref(inp)
    loss = torch.nn.functional.cross_entropy(logits, inp["segment"], ignore_index=-1)
    loss.backward()
    names = [k for k, _ in ref.named_parameters()]
    np.savez_compressed(
        os.path.join(OUT, "spunet_tiny.npz"), scene_seeds=np.asarray([31, 32]), n_points=np.asarray([3000, 1200]),
        input_checksum=np.asarray([batch["grid_coord"].sum(), float(batch["feat"].astype(np.float64).sum())]),
        logits_eval=logits_eval[::4].astype(np.float32), logits_train=logits.detach().numpy()[::4].astype(np.float32),
        logits_absmax=np.asarray(float(np.abs(logits_eval).max())),
        loss=np.asarray(float(loss.detach())), param_names=np.asarray(names),
        grad_norms=np.asarray([float(p.grad.double().norm()) for _, p in ref.named_parameters()]),
        grad_conv_input=ref.conv_input[0].weight.grad.numpy().astype(np.float32),
        grad_final=ref.final.weight.grad.numpy().astype(np.float32),
        n_state=np.asarray(len(sd)))
    # ---- PTv3 dense attention branch + RPE (ptv3m1:29-48,173-206) -------------------------------
    torch.manual_seed(0)
    ref = ptv3.PointTransformerV3(**RPE_CFG)
    sd = om.deterministic_state_dict(ref, 2)
    ref.load_state_dict(sd)
    batch = synthetic.collate([synthetic.indoor_scene(23, 900), synthetic.indoor_scene(24, 200)])
    inp = {k: torch.from_numpy(v) for k, v in batch.items()}
    ref.eval()
    torch.manual_seed(5)     # SerializedPooling shuffles the order rows with the CPU generator even when the model's
    with torch.no_grad():    # shuffle_orders is False (ptv3m1:624-632 builds it with its default shuffle_orders=True)
        feat_eval = ref(dict(inp)).feat.numpy()
    ref.train()
    torch.manual_seed(6)
    feat = ref(dict(inp)).feat
    loss = feat.pow(2).mean()
    loss.backward()
    names = [k for k, _ in ref.named_parameters()]
    np.savez_compressed(
        os.path.join(OUT, "ptv3_rpe.npz"), scene_seeds=np.asarray([23, 24]), n_points=np.asarray([900, 200]),
        input_checksum=np.asarray([batch["grid_coord"].sum()]), feat_eval_rows=feat_eval[::4].astype(np.float32), feat_absmax=np.asarray(float(np.abs(feat_eval).max())),
        feat_train_rows=feat.detach().numpy()[::4].astype(np.float32), loss=np.asarray(float(loss.detach())),
        param_names=np.asarray(names),
        grad_norms=np.asarray([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in ref.named_parameters()]),
        grad_rpe_dec0=ref.dec.dec0.block0.attn.rpe.rpe_table.grad.numpy().astype(np.float32))

    # ---- GridSample (voxelisation) --------------------------------------------------------------
    tr = ref_import.load_transform()
    blobs = {}
    gcases = [(3000, 0.02, 0.6, 0), (5000, 0.05, 3.0, 1), (64, 0.1, 0.05, 2)]
    for ci, (n, grid, extent, seed) in enumerate(gcases):
        coord = om.gridsample_case(seed, n, extent)
        gs = tr.GridSample(grid_size=grid, hash_type="fnv", mode="train", return_grid_coord=True, return_inverse=True,
                           return_min_coord=True)
        np.random.seed(seed)
        d = gs(dict(coord=coord.copy(), segment=np.arange(n), index_valid_keys=["coord", "segment"]))
        order = np.lexsort(d["grid_coord"].T[::-1])
        blobs[f"params_{ci}"] = np.asarray([n, grid, extent, seed], dtype=np.float64)
        blobs[f"coord_sum_{ci}"] = np.asarray(float(coord.astype(np.float64).sum()))
        blobs[f"inverse_{ci}"] = d["inverse"].astype(np.int32)
        blobs[f"voxels_sorted_{ci}"] = d["grid_coord"][order].astype(np.int32)       # the voxel set
        blobs[f"voxels_keyorder_{ci}"] = d["grid_coord"].astype(np.int32)            # ... in ascending-key (np.unique) order
        blobs[f"min_coord_{ci}"] = d["min_coord"]
        blobs[f"picked_{ci}"] = d["segment"].astype(np.int32)                        # the reference's own representatives
    blobs["n_cases"] = np.asarray(len(gcases))
    np.savez_compressed(os.path.join(OUT, "gridsample.npz"), **blobs)

    # ---- PTv3 enc_mode: the pooling_parent / pooling_inverse chain consumed at models/default.py:69-74 ----------
    torch.manual_seed(0)
    ref = ptv3.PointTransformerV3(enc_mode=True, **ENC_CFG)
    ref.load_state_dict(om.deterministic_state_dict(ref, 3))
    ref.eval()
    batch = synthetic.collate([synthetic.indoor_scene(27, 1500), synthetic.indoor_scene(28, 400)])
    torch.manual_seed(5)
    with torch.no_grad():
        pt = ref({k: torch.from_numpy(v) for k, v in batch.items()})
        sizes = [pt.feat.shape[0]]
        while "pooling_parent" in pt.keys():
            parent, inverse = pt.pop("pooling_parent"), pt.pop("pooling_inverse")
            parent.feat = torch.cat([parent.feat, pt.feat[inverse]], dim=-1)
            pt = parent
            sizes.append(pt.feat.shape[0])
    feat = pt.feat.numpy()
    np.savez_compressed(
        os.path.join(OUT, "ptv3_enc_mode.npz"), scene_seeds=np.asarray([27, 28]), n_points=np.asarray([1500, 400]),
        input_checksum=np.asarray([batch["grid_coord"].sum()]), stage_sizes=np.asarray(sizes[::-1]),
        feat_rows=feat[::32].astype(np.float32), feat_absmax=np.asarray(float(np.abs(feat).max())),
        feat_col_norm=np.linalg.norm(feat.astype(np.float64), axis=0).astype(np.float32))

    # ---- PT-v3m2 (Sonata backbone) tiny -----------------------------------------------------------
    import importlib
    m2 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m2_sonata")
    torch.manual_seed(0)
    ref = m2.PointTransformerV3(**M2_CFG)
    sd = om.deterministic_state_dict(ref, 33)
    ref.load_state_dict(sd)
    batch = synthetic.collate([synthetic.indoor_scene(61, 2500), synthetic.indoor_scene(62, 700)])
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
    np.savez_compressed(
        os.path.join(OUT, "ptv3m2_tiny.npz"), scene_seeds=np.asarray([61, 62]), n_points=np.asarray([2500, 700]),
        input_checksum=np.asarray([batch["grid_coord"].sum(), float(batch["feat"].astype(np.float64).sum())]),
        weight_checksum=np.asarray(float(sum(float(v.double().abs().sum()) for v in sd.values()))),
        state_keys=np.asarray(list(sd.keys())), feat_rows=out_eval[::8].astype(np.float32), feat_absmax=np.asarray(np.abs(out_eval).max()),
        loss=np.asarray(float(loss.detach())), grad_names=np.asarray(names),
        grad_norms=np.asarray([float(p.grad.norm()) for _, p in ref.named_parameters()], dtype=np.float64))

    # ---- libs/pointrope: the reference's own pointrope_cpu, compiled by oracle/build_ref.py ------------
    from oracle import build_ref
    ext = build_ref.build_pointrope()
    blobs = {}
    cases = [(2, 37, 2, 18, 100.0, 1.0), (1, 200, 4, 24, 100.0, -1.0), (3, 16, 3, 48, 10.0, 1.0), (1, 5, 1, 6, 100.0, 1.0)]
    for ci, (B, N, H, D, base, fwd) in enumerate(cases):
        g = torch.Generator().manual_seed(700 + ci)
        tok = torch.randn(B, N, H, D, generator=g)
        pos = torch.randint(0, 300, (B, N, 3), generator=g)
        out = tok.clone()
        ext.pointrope(out, pos, base, fwd)
        blobs[f"shape_{ci}"] = np.asarray([B, N, H, D], dtype=np.int64)
        blobs[f"params_{ci}"] = np.asarray([base, fwd], dtype=np.float64)
        blobs[f"tokens_{ci}"], blobs[f"pos_{ci}"], blobs[f"out_{ci}"] = tok.numpy(), pos.numpy(), out.numpy()
    blobs["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(OUT, "pointrope.npz"), **blobs)

    # ---- SphereCrop (pointcept/datasets/transform.py:1014-1057) ----------------------------------------
    TR = ref_import.load_transform()
    blobs = {}
    rng = np.random.default_rng(11)
    cases = [(5000, 1200, "center"), (3000, 2999, "center"), (2000, 500, "random")]
    for ci, (n, pmax, mode) in enumerate(cases):
        coord = (rng.random((n, 3)) * np.array([7.0, 5.0, 2.8])).astype(np.float32)
        seg = rng.integers(0, 20, size=n).astype(np.int64)
        d = dict(coord=coord.copy(), segment=seg.copy())
        np.random.seed(123 + ci)
        ci_center = None
        if mode == "random":     # the centre the transform is about to draw
            st = np.random.get_state()
            ci_center = int(np.random.randint(n))
            np.random.set_state(st)
        out = TR.SphereCrop(point_max=pmax, mode=mode)(d)
        blobs[f"coord_{ci}"], blobs[f"segment_{ci}"], blobs[f"point_max_{ci}"] = coord, seg, np.asarray(pmax)
        blobs[f"mode_{ci}"], blobs[f"center_index_{ci}"] = np.asarray(mode), np.asarray(-1 if ci_center is None else ci_center)
        blobs[f"out_coord_{ci}"], blobs[f"out_segment_{ci}"] = out["coord"], out["segment"]
    blobs["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(OUT, "spherecrop.npz"), **blobs)

    # ---- Lovasz-Softmax ------------------------------------------------------------------------
    import types
    pkg = types.ModuleType("pointcept.models.losses")
    pkg.__path__ = [ref_import.REF + "/pointcept/models/losses"]
    sys.modules["pointcept.models.losses"] = pkg
    lov = importlib.import_module("pointcept.models.losses.lovasz")
    crit = lov.LovaszLoss(mode="multiclass", ignore_index=-1, loss_weight=1.0)
    blobs = {}
    cases = [(1000, 20, 0.05, 20, 1.0), (2049, 13, 0.3, 9, 4.0), (257, 20, 0.0, 3, 0.5), (64, 5, 0.9, 5, 2.0), (1, 20, 0.0, 20, 1.0)]
    for ci, (n, c, p_ignore, n_used, spread) in enumerate(cases):
        x, y = om.lovasz_case(ci, n, c, p_ignore, n_used, spread)   # seeded; regenerated by the tests, checksummed here
        x.requires_grad_(True)
        loss = crit(x, y)
        loss.backward()
        blobs[f"shape_{ci}"] = np.asarray([n, c, n_used], dtype=np.int64)
        blobs[f"params_{ci}"] = np.asarray([p_ignore, spread], dtype=np.float64)
        blobs[f"logits_sum_{ci}"], blobs[f"labels_{ci}"] = np.asarray(float(x.detach().double().sum())), y.numpy().astype(np.int8)
        blobs[f"loss_{ci}"], blobs[f"grad_{ci}"] = np.asarray(float(loss.detach())), x.grad.numpy()
    blobs["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(OUT, "lovasz.npz"), **blobs)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
