"""-m gpu: parity at the BASELINE sizes and depths (VERDICT r1 "no full-size, full-depth parity").

BASELINE.json configs[2] / [1] / [4] at THEIR OWN size, engine (libptcore.so) vs the CPU oracle on the same seeded
inputs and weights:
  (i)   PT-v3m1 BASE depths / channels, eval mode, one 102400-voxel indoor scene: every index map of every stage
        bit-exact (enc_mode chain), logits within the stated tolerance -- fp32 engine (algorithm parity) and the bf16
        autocast path the benchmark runs;
  (ii)  the same model in train mode (drop_path 0) on a 20480-voxel scene: loss, per-parameter and per-stage gradients;
        bars 3 % per stage / 6 % per parameter against the all-fp32 oracle, and at every stage the deviation must lie
        inside the envelope that the kernels' bf16 roundings ALONE produce when emulated inside the CPU oracle
        (oracle/ops.py _AttnKernelRounding): the evidence that the deviation is attention rounding and nothing else;
  (iii) SpUNet-v1m1 BASE layers (2,3,4,6,2,2,2,2) on one 100000-voxel scene, forward;
  (iv)  PT-v3m1 outdoor (in_channels 4, depth 12 grid) on one ~120k-voxel LiDAR-like scene, forward;
  (v)   one optimizer step under fp16 autocast + torch.amp.GradScaler exactly as engines/train.py:203-231;
  (vi)  (i) again at B = 2 with ragged scene sizes 102400 + 77777: the padding-borrow path of the patch maps at stage 0.
The oracle needs ~10-40 s per case on the GPU box's host cores; sizes were chosen so the whole file stays under 5 min.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")
# the CPU dry run of these bodies (tests/test_gpu_tests_dry_run_cpu.py) shrinks the scenes; the -m gpu run uses 1.0
SCALE = float(os.environ.get("PTC_FULLSIZE_SCALE", "1"))


def _n(points):
    return max(1200, int(points * SCALE))


BASE = dict(  # configs/scannet/semseg-pt-v3m1-0-base.py:15-47 with drop_path 0 / fixed orders (parity needs no RNG)
    in_channels=6, order=ORDERS, stride=(2, 2, 2, 2), enc_depths=(2, 2, 2, 6, 2), enc_channels=(32, 64, 128, 256, 512),
    enc_num_head=(2, 4, 8, 16, 32), enc_patch_size=(1024,) * 5, dec_depths=(2, 2, 2, 2), dec_channels=(64, 64, 128, 256),
    dec_num_head=(4, 4, 8, 16), dec_patch_size=(1024,) * 4, mlp_ratio=4, qkv_bias=True, drop_path=0.0, shuffle_orders=False)


def _pair(cfg, seed=0):
    from oracle import ptv3_model as om
    from pointcept_amd.point_transformer_v3 import PointTransformerV3

    torch.manual_seed(0)
    orc = om.PointTransformerV3(**cfg)
    eng = PointTransformerV3(**cfg)
    sd = om.deterministic_state_dict(orc, seed)
    orc.load_state_dict(sd)
    eng.load_state_dict(sd)
    return orc, eng


def _rel_max(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def _rel_fro(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def _report(name, lines):
    """gpurun_out/<name> on an MI355X run; gpurun_out/dryrun_<name> when the bodies run on the CPU stand-ins (scaled scenes): the
    dry run must never overwrite the hardware evidence"""
    os.makedirs("gpurun_out", exist_ok=True)
    on_gpu = torch.cuda.is_available() and SCALE == 1.0
    head = [f"# {'MI355X run (' + torch.cuda.get_device_name(0) + ')' if on_gpu else 'CPU dry run on oracle stand-ins, scene scale ' + str(SCALE)}"]
    with open(os.path.join("gpurun_out", name if on_gpu else "dryrun_" + name), "w") as f:
        f.write("\n".join(head + lines) + "\n")


def _chain(point):
    out = [point]
    while "pooling_parent" in point.keys():
        point = point["pooling_parent"]
        out.append(point)
    return out[::-1]   # stage 0 first


def test_ptv3_base_one_full_scene_maps_and_logits(cuda):
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic
    from pointcept_amd.segmentor import DefaultSegmentorV2

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    scene = synthetic.collate([synthetic.indoor_scene(7, _n(102400))])
    assert scene["grid_coord"].shape[0] == _n(102400)
    host = {k: torch.from_numpy(v) for k, v in scene.items()}
    lines = []
    # ---- index maps of every stage: encoder-only models return the deepest Point with its parent chain
    enc_cfg = {k: v for k, v in BASE.items() if not k.startswith("dec_")}
    orc_e, eng_e = _pair(dict(enc_cfg, enc_mode=True))
    eng_e = eng_e.to(cuda).eval()
    orc_e.eval()
    with torch.no_grad():
        torch.manual_seed(11)
        pe = eng_e(synthetic.to_torch(scene, cuda))
        torch.manual_seed(11)
        po = orc_e(dict(host))
    ce, co = _chain(pe), _chain(po)
    assert len(ce) == len(co) == 5
    for s, (a, b) in enumerate(zip(ce, co)):
        assert a.feat.shape[0] == b.feat.shape[0], f"stage {s}: {a.feat.shape[0]} vs {b.feat.shape[0]} points"
        for key in ("serialized_code", "serialized_order", "serialized_inverse", "grid_coord", "batch", "offset"):
            assert torch.equal(a[key].cpu().long(), b[key].long()), f"stage {s}: {key}"
        for key in ("pad", "unpad", "cu_seqlens_key"):
            assert torch.equal(a[key].cpu().long(), b[key].long()), f"stage {s}: {key}"
        if s > 0:
            assert torch.equal(a["pooling_inverse"].cpu().long(), b["pooling_inverse"].long()), f"stage {s}: pooling_inverse"
        lines.append(f"stage {s}: n = {a.feat.shape[0]}, feat rel_max {_rel_max(a.feat, b.feat):.3e} rel_fro {_rel_fro(a.feat, b.feat):.3e}")
        assert _rel_max(a.feat, b.feat) < 2e-2, f"stage {s} encoder features"
    del orc_e, eng_e, pe, po, ce, co
    # ---- full model, logits
    orc_b, eng_b = _pair(BASE)
    torch.manual_seed(1)
    orc = om.SegmentorV2(20, 64, orc_b).eval()
    eng = DefaultSegmentorV2(20, 64, eng_b)
    eng.seg_head.load_state_dict(orc.seg_head.state_dict())
    eng = eng.to(cuda).eval()
    dev_in = synthetic.to_torch(scene, cuda)
    with torch.no_grad():
        torch.manual_seed(11)
        lo = orc({k: v for k, v in host.items() if k != "segment"})["seg_logits"]
        torch.manual_seed(11)
        le = eng({k: v for k, v in dev_in.items() if k != "segment"})["seg_logits"]
        torch.manual_seed(11)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            la = eng({k: v for k, v in dev_in.items() if k != "segment"})["seg_logits"]
    assert torch.isfinite(le).all() and torch.isfinite(la).all()
    agree32 = float((le.argmax(1).cpu() == lo.argmax(1)).float().mean())
    agree16 = float((la.float().argmax(1).cpu() == lo.argmax(1)).float().mean())
    lines += [f"logits fp32 engine  vs oracle: rel_max {_rel_max(le, lo):.3e} rel_fro {_rel_fro(le, lo):.3e} argmax agreement {agree32:.4f}",
              f"logits bf16 autocast vs oracle: rel_max {_rel_max(la, lo):.3e} rel_fro {_rel_fro(la, lo):.3e} argmax agreement {agree16:.4f}"]
    _report("fullsize_ptv3_forward.txt", lines)
    assert _rel_max(le, lo) < 2e-2, lines[-2]
    # bf16 activations through 22 blocks: per-element bar 8e-2 of the logit range, mean-square bar 3e-2, and the
    # PREDICTIONS (what mIoU is computed from) must agree on >= 97 % of the voxels
    assert _rel_max(la, lo) < 8e-2 and _rel_fro(la, lo) < 3e-2, lines[-1]
    assert agree32 > 0.995 and agree16 > 0.97, lines[-2:]


def test_ptv3_base_two_ragged_full_scenes_padding_borrow(cuda):
    """base depth, B = 2 with scene sizes 102400 and 77777 -- 77777 = 75 x 1024 + 977, so the last patch of the
    second scene BORROWS 47 ranks from the previous patch at stage 0 (ptv3m1:144-154), and every deeper stage pads both scenes.
    Pad / unpad / cu_seqlens of every stage bit-exact against the oracle, logits (fp32 engine and the bf16 autocast path) inside the
    bars of the one-scene test."""
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic
    from pointcept_amd.segmentor import DefaultSegmentorV2

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    scene = synthetic.collate([synthetic.indoor_scene(21, _n(102400)), synthetic.indoor_scene(22, _n(77777))])
    host = {k: torch.from_numpy(v) for k, v in scene.items()}
    lines = [f"scenes: {_n(102400)} + {_n(77777)} voxels"]
    enc_cfg = {k: v for k, v in BASE.items() if not k.startswith("dec_")}
    orc_e, eng_e = _pair(dict(enc_cfg, enc_mode=True), seed=3)
    eng_e = eng_e.to(cuda).eval()
    orc_e.eval()
    with torch.no_grad():
        torch.manual_seed(11)
        pe = eng_e(synthetic.to_torch(scene, cuda))
        torch.manual_seed(11)
        po = orc_e(dict(host))
    for s_, (a, b) in enumerate(zip(_chain(pe), _chain(po))):
        assert a.feat.shape[0] == b.feat.shape[0], f"stage {s_}"
        for key in ("serialized_order", "serialized_inverse", "offset", "pad", "unpad", "cu_seqlens_key"):
            assert torch.equal(a[key].cpu().long(), b[key].long()), f"stage {s_}: {key}"
        n_pad = int(a["pad"].numel())
        lines.append(f"stage {s_}: n = {a.feat.shape[0]}, padded slots {n_pad}, sequences {int(a['cu_seqlens_key'].numel()) - 1}, feat rel_max {_rel_max(a.feat, b.feat):.3e}")
        assert _rel_max(a.feat, b.feat) < 2e-2, f"stage {s_} encoder features"
    if SCALE == 1.0:
        assert int(_chain(pe)[0]["pad"].numel()) == 102400 + 76 * 1024        # the borrow path: 77777 -> 76 full patches
    del orc_e, eng_e, pe, po
    orc_b, eng_b = _pair(BASE, seed=3)
    orc = om.SegmentorV2(20, 64, orc_b).eval()
    eng = DefaultSegmentorV2(20, 64, eng_b)
    eng.seg_head.load_state_dict(orc.seg_head.state_dict())
    eng = eng.to(cuda).eval()
    dev_in = synthetic.to_torch(scene, cuda)
    with torch.no_grad():
        torch.manual_seed(11)
        lo = orc({k: v for k, v in host.items() if k != "segment"})["seg_logits"]
        torch.manual_seed(11)
        le = eng({k: v for k, v in dev_in.items() if k != "segment"})["seg_logits"]
        torch.manual_seed(11)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            la = eng({k: v for k, v in dev_in.items() if k != "segment"})["seg_logits"]
    agree32 = float((le.argmax(1).cpu() == lo.argmax(1)).float().mean())
    agree16 = float((la.float().argmax(1).cpu() == lo.argmax(1)).float().mean())
    lines += [f"logits fp32 engine  vs oracle: rel_max {_rel_max(le, lo):.3e} rel_fro {_rel_fro(le, lo):.3e} argmax agreement {agree32:.4f}",
              f"logits bf16 autocast vs oracle: rel_max {_rel_max(la, lo):.3e} rel_fro {_rel_fro(la, lo):.3e} argmax agreement {agree16:.4f}"]
    _report("fullsize_ptv3_two_scenes_forward.txt", lines)
    assert _rel_max(le, lo) < 2e-2, lines[-2]
    assert _rel_max(la, lo) < 8e-2 and _rel_fro(la, lo) < 3e-2, lines[-1]
    assert agree32 > 0.995 and agree16 > 0.97, lines[-2:]


def _stage_of(name):
    for tag in ("embedding", "enc.enc0", "enc.enc1", "enc.enc2", "enc.enc3", "enc.enc4", "dec.dec3", "dec.dec2", "dec.dec1", "dec.dec0",
                "seg_head"):
        if tag in name:
            return tag
    return "other"


def test_ptv3_base_train_step_gradients_vs_oracle(cuda):
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic
    from pointcept_amd.segmentor import DefaultSegmentorV2

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    scene = synthetic.collate([synthetic.indoor_scene(9, _n(20480))])
    host = {k: torch.from_numpy(v) for k, v in scene.items()}
    orc_b, eng_b = _pair(BASE, seed=2)
    torch.manual_seed(1)
    orc = om.SegmentorV2(20, 64, orc_b).train()
    eng = DefaultSegmentorV2(20, 64, eng_b)
    eng.seg_head.load_state_dict(orc.seg_head.state_dict())
    eng = eng.to(cuda).train()
    torch.manual_seed(13)
    le = eng(synthetic.to_torch(scene, cuda))["loss"]
    le.backward()
    ge = {n: p.grad.detach().float().cpu() for n, p in eng.named_parameters()}
    results, grads = {}, {}
    for mode in ("fp32", "kernel"):
        for m in orc.modules():
            if type(m).__name__ == "SerializedAttention":
                m.attn_rounding = "kernel" if mode == "kernel" else None
        orc.zero_grad(set_to_none=True)
        torch.manual_seed(13)
        lo = orc(dict(host))["loss"]
        lo.backward()
        grads[mode] = {n: p.grad.detach().clone() for n, p in orc.named_parameters()}
        results[mode] = float(lo.detach())

    def compare(ga, gb):
        """per-parameter and per-stage relative Frobenius distance of gradient set ga from gb"""
        gmax = max(float(g.abs().max()) for g in gb.values())
        per_param, per_stage = [], {}
        for n, g in gb.items():
            d = float((ga[n] - g).norm())
            r = float(g.norm())
            tiny = float(g.abs().max()) < 1e-5 * gmax          # true gradient zero (bias in front of a batch-stat BN)
            per_param.append((n, d / max(r, 1e-30), r, tiny, d))
            st = per_stage.setdefault(_stage_of(n), [0.0, 0.0])
            st[0] += d * d
            st[1] += r * r
        return per_param, {k: (v[0] ** 0.5) / max(v[1] ** 0.5, 1e-30) for k, v in per_stage.items()}, gmax

    eng_pp, eng_ps, gmax = compare(ge, grads["fp32"])               # engine vs exact-arithmetic oracle
    emu_pp, emu_ps, _ = compare(grads["kernel"], grads["fp32"])     # what the kernels' 16-bit roundings ALONE do to the oracle
    lines = [f"engine loss {float(le):.6f}; oracle loss fp32 {results['fp32']:.6f}, with the kernels' roundings {results['kernel']:.6f}",
             "stage        engine-vs-fp32-oracle   rounding-emulation-vs-fp32-oracle (rel. Frobenius)"]
    lines += [f"   {k:10s} {eng_ps[k]:.3e}               {emu_ps[k]:.3e}" for k in eng_ps]
    worst = sorted([p for p in eng_pp if not p[3]], key=lambda p: -p[1])[:8]
    lines += [f"   worst engine {p[1]:.3e} (|g| {p[2]:.3e}) {p[0]}" for p in worst]
    _report("fullsize_ptv3_grad_report.txt", lines)
    assert abs(float(le) - results["fp32"]) < 5e-3 * abs(results["fp32"]), (float(le), results["fp32"])
    on_gpu = torch.device(cuda).type == "cuda"
    for n, rel, rn, tiny, d in eng_pp:
        if tiny:
            assert d <= 1e-4 * gmax * max(1.0, ge[n].numel() ** 0.5), n
        else:
            assert rel < (6e-2 if on_gpu else 2e-2), (n, rel)        # r1 accepted 10 % here (tiny model); measured 3.9e-2
    assert max(eng_ps.values()) < (3e-2 if on_gpu else 5e-3), eng_ps   # measured 2.2e-2 on MI355X
    if on_gpu:
        # The engine's deviation is bf16 attention rounding and nothing else: rounding P / dS / the output to bf16 in the
        # fp32 CPU oracle exactly where the kernels round (oracle/ops.py _AttnKernelRounding) moves the ORACLE's own
        # gradients by 3-5 % per stage; the engine must sit inside that envelope at every stage.  (The two cannot agree
        # rounding for rounding: a different fp32 summation order flips individual bf16 roundings.)
        for k in eng_ps:
            assert eng_ps[k] <= 1.3 * emu_ps[k] + 5e-3, (k, eng_ps[k], emu_ps[k])       # measured ratios 0.8 - 1.15 (round 3); 2.0 until round 4


def test_spunet_base_one_full_scene_forward(cuda):
    from oracle import ptv3_model as om
    from oracle import spunet_model as osp
    from pointcept_amd import synthetic
    from pointcept_amd.sparse_unet import SpUNetBase

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    scene = synthetic.collate([synthetic.indoor_scene(5, _n(100000))])
    kw = dict(channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2))   # scannet/semseg-spunet-v1m1-0-base.py:16-17
    torch.manual_seed(0)
    orc = osp.SpUNetBase(6, 20, **kw)
    eng = SpUNetBase(6, 20, **kw)
    assert list(orc.state_dict().keys()) == list(eng.state_dict().keys())
    sd = om.deterministic_state_dict(orc, 4)
    orc.load_state_dict(sd)
    eng.load_state_dict(sd)
    orc.eval()
    eng = eng.to(cuda).eval()
    host = {k: torch.from_numpy(v) for k, v in scene.items()}
    with torch.no_grad():
        lo = orc(dict(host))
        le = eng(synthetic.to_torch(scene, cuda))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            la = eng(synthetic.to_torch(scene, cuda))
    agree32 = float((le.argmax(1).cpu() == lo.argmax(1)).float().mean())
    agree16 = float((la.float().argmax(1).cpu() == lo.argmax(1)).float().mean())
    lines = [f"SpUNet-v1m1 base, 100000 voxels: fp32 rel_max {_rel_max(le, lo):.3e} rel_fro {_rel_fro(le, lo):.3e} argmax {agree32:.4f}",
             f"                                 bf16 rel_max {_rel_max(la, lo):.3e} rel_fro {_rel_fro(la, lo):.3e} argmax {agree16:.4f}"]
    _report("fullsize_spunet_forward.txt", lines)
    assert le.shape == lo.shape == (_n(100000), 20)
    assert _rel_max(le, lo) < 1e-3, lines[0]          # fp32 kernels end to end: summation order only
    assert _rel_max(la, lo) < 8e-2 and _rel_fro(la, lo) < 3e-2 and agree16 > 0.97, lines[1]


def test_ptv3_outdoor_full_scene_forward(cuda):
    from pointcept_amd import synthetic

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    scene = synthetic.collate([synthetic.outdoor_scene(3, _n(120000))])
    n = scene["grid_coord"].shape[0]
    assert n >= _n(100000) and int(scene["grid_coord"].max()) >= 2048          # depth 12 keys
    cfg = dict(BASE, in_channels=4)                                        # nuscenes/semseg-pt-v3m1-0-base.py:16
    orc, eng = _pair(cfg, seed=5)
    orc.eval()
    eng = eng.to(cuda).eval()
    with torch.no_grad():
        torch.manual_seed(3)
        po = orc({k: torch.from_numpy(v) for k, v in scene.items()})
        torch.manual_seed(3)
        pe = eng(synthetic.to_torch(scene, cuda))
    assert pe.serialized_depth == po.serialized_depth >= 12
    for key in ("serialized_code", "serialized_order", "serialized_inverse"):
        assert torch.equal(pe[key].cpu(), po[key]), key
    _report("fullsize_ptv3_outdoor.txt", [f"outdoor n = {n} depth {pe.serialized_depth}: feat rel_max {_rel_max(pe.feat, po.feat):.3e} "
                                          f"rel_fro {_rel_fro(pe.feat, po.feat):.3e}"])
    assert _rel_max(pe.feat, po.feat) < 2e-2


def test_ptv3_fp16_gradscaler_step(cuda):
    """engines/train.py:203-231: fp16 autocast forward, scaler.scale(loss).backward(), scaler.step(optimizer) with its
    unscale + inf check, scaler.update().  The step must be taken (finite gradients at the default initial scale after
    the scaler's own back-off), move the weights, and the loss must agree with the bf16-autocast and fp32 forwards."""
    from pointcept_amd import synthetic
    from pointcept_amd.point_transformer_v3 import PointTransformerV3
    from pointcept_amd.segmentor import DefaultSegmentorV2

    cfg = dict(BASE, enc_depths=(1, 1, 1, 2, 1), dec_depths=(1, 1, 1, 1))
    torch.manual_seed(0)
    model = DefaultSegmentorV2(20, 64, PointTransformerV3(**cfg), criteria=("ce", "lovasz")).to(cuda).train()
    batch = synthetic.to_torch(synthetic.collate([synthetic.indoor_scene(41, _n(30000)), synthetic.indoor_scene(42, _n(9000))]), cuda)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05, fused=True)
    scaler = torch.amp.GradScaler("cuda")
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    with torch.no_grad():
        torch.manual_seed(2)
        l32 = float(model(dict(batch))["loss"])
    steps_taken, losses = 0, []
    for it in range(6):
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(2)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = model(dict(batch))["loss"]
        assert torch.isfinite(loss), f"fp16 loss not finite at iteration {it}"
        scaler.scale(loss).backward()
        scale_before = scaler.get_scale()
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss))
        if scaler.get_scale() >= scale_before:      # no overflow found: the optimizer step ran
            steps_taken += 1
    assert abs(losses[0] - l32) < 3e-2 * abs(l32), (losses[0], l32)
    assert steps_taken >= 3, f"GradScaler skipped too many steps: scale {scaler.get_scale()}, losses {losses}"
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters())
    assert moved == len(before), f"only {moved}/{len(before)} parameters moved"
    assert losses[-1] < losses[0], losses                   # same batch, 3+ AdamW steps: the loss goes down
    for p in model.parameters():
        assert torch.isfinite(p).all()


def _spunet_stage_of(name):
    parts = name.split(".")
    return ".".join(parts[:2]) if parts[0] in ("down", "enc", "up", "dec") else parts[0]


def _stage_distances(ga, gb, stage_of):
    """{stage: relative Frobenius distance of gradient set ga from gb}, worst parameters"""
    per_stage, per_param = {}, []
    for n, g in gb.items():
        d, r = float((ga[n].double() - g.double()).norm()), float(g.double().norm())
        st = per_stage.setdefault(stage_of(n), [0.0, 0.0])
        st[0] += d * d
        st[1] += r * r
        per_param.append((d / max(r, 1e-30), r, n))
    return {k: (v[0] ** 0.5) / max(v[1] ** 0.5, 1e-30) for k, v in per_stage.items()}, sorted(per_param, reverse=True)[:6]


class _RoundSTE(torch.autograd.Function):
    """x -> x rounded to `dtype` (kept in fp32); the gradient is rounded the same way: what a 16-bit tensor between two modules does"""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).float(), None


def _autocast_rounding_twin(model, dtype):
    """fp32 copy of an engine model that rounds where 16-bit autocast rounds (spconv + BatchNorm + ReLU under AMP: every conv,
    norm and activation output is a 16-bit tensor, conv weights are cast): the fp32 kernels -- pinned to the oracle -- plus the
    ROUNDINGS of the 16-bit path and nothing else.  Hooked activations are never fused (PNN.fused_act), so the twin runs the
    reference's module sequence."""
    import copy

    from pointcept_amd import spconv_api

    twin = copy.deepcopy(model)
    with torch.no_grad():
        for m in twin.modules():
            if isinstance(m, spconv_api._SparseConvolution):
                m.weight.copy_(m.weight.to(dtype).float())

    def hook(m, inp, out):
        if torch.is_tensor(out):
            return _RoundSTE.apply(out, dtype) if out.is_floating_point() else out
        if hasattr(out, "features") and out.features is not None:
            return out.replace_feature(_RoundSTE.apply(out.features, dtype))
        return out

    for m in twin.modules():
        if not list(m.children()) and not isinstance(m, torch.nn.Identity):
            m.register_forward_hook(hook)
    return twin


def test_spunet_base_two_full_scenes_train_step_vs_oracle(cuda):
    """BASELINE configs[1] end to end, forward AND backward, at 2 x 100000 voxels: SpUNet-v1m1 base
    (scannet/semseg-spunet-v1m1-0-base.py:16-17) in train mode (batch-statistics BatchNorm) + CE, every parameter gradient grouped by
    stage (conv_input, down.s, enc.s, up.s, dec.s, final) against the fp32 CPU oracle -- the sliced wgrad7 instances (96 / 128 / 224 / 192
    channels), the 128-column conv3 instances and the strided / inverse tables on the engine side.  Three engine runs on the same weights:
      fp32           algorithm parity (summation order only): loss 1e-5, logits 2e-3, every stage's gradient 1e-2;
      bf16 autocast  (the bench secondary) and fp16 autocast with a fixed loss scale of 1024 (configs[1]'s AMP dtype; GradScaler's
                     unscale is the division below): loss 2e-3, logits 8e-2 max / 3e-2 Frobenius, arg-max agreement 0.97.
    44 convolutions deep with batch-statistics BatchNorm, a 16-bit rounding of every intermediate tensor moves the GRADIENTS of this
    network by tens of per cent at the deep stages whoever does the rounding (measured 28-53 % for bf16 at enc / down, on the MI355X
    kernels and on plain torch CPU autocast alike, profiles/r05_b_fullsize_spunet_step.txt).  The 16-bit gradients are therefore
    judged against an envelope: the SAME engine in fp32 with each module output / gradient rounded to the autocast dtype
    (_autocast_rounding_twin) deviates from the fp32 oracle by e(stage); the 16-bit kernels must stay inside 1.1 e(stage) + 5e-3 (round 6,
    VERDICT r5 weak 2: the two columns agreed to 2-3 % of themselves in every session of round 5; the old 1.5 e + 2e-2 would have let an
    80 % gradient error pass at the deep stages).  Is the 28-53 % the DATA (labels U{0..19}: every class gradient a sum of cancelling
    terms)?  No: with structured labels -- segment = f(0.5 m cell of the voxel), PTC_TEST_STRUCTURED_LABELS=1 -- the same run gives
    23-57 % for bf16 and 9-20 % for fp16 at the same stages, kernels and envelope again within a few per cent of each other
    (profiles/r06_b_fullsize_spunet_step_structured_labels.txt; the fp32 leg there sits at 1.2e-2 instead of 5e-3).  It is the network:
    44 convolutions with batch-statistics BatchNorm amplify every rounding of an intermediate tensor, whoever rounds."""
    from oracle import ptv3_model as om
    from oracle import spunet_model as osp
    from pointcept_amd import functional as PF
    from pointcept_amd import synthetic
    from pointcept_amd.sparse_unet import SpUNetBase

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    on_gpu = torch.device(cuda).type == "cuda"
    batch = synthetic.collate([synthetic.indoor_scene(61, _n(100000)), synthetic.indoor_scene(62, _n(100000))])
    if os.environ.get("PTC_TEST_STRUCTURED_LABELS") == "1":             # the one-off run VERDICT r5 weak 2 asked for (see the docstring)
        cell = batch["grid_coord"] // 25                               # 0.5 m cells at the 0.02 m grid
        batch["segment"] = np.where(batch["segment"] >= 0, (cell[:, 0] + 3 * cell[:, 1] + 7 * cell[:, 2]) % 20, -1).astype(np.int64)
    kw = dict(channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2))
    torch.manual_seed(0)
    orc, eng = osp.SpUNetBase(6, 20, **kw), SpUNetBase(6, 20, **kw)
    sd = om.deterministic_state_dict(orc, 6)
    orc.load_state_dict(sd)
    eng.load_state_dict(sd)
    orc.train()
    eng = eng.to(cuda).train()
    import time
    t0 = time.time()
    out_o = osp.Segmentor(orc)({k: torch.from_numpy(v) for k, v in batch.items()})
    out_o["loss"].backward()
    t_orc = time.time() - t0
    go = {n: p.grad.detach() for n, p in orc.named_parameters()}
    lo, logits_o = float(out_o["loss"].detach()), out_o["seg_logits"].detach()
    dev = synthetic.to_torch(batch, cuda)
    lines = [f"SpUNet-v1m1 base, 2 x {_n(100000)} voxels, train mode, CE; fp32 CPU oracle fwd+bwd {t_orc:.1f} s on {torch.get_num_threads()} threads"]

    def step(model, dtype, scale):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None):
            logits = model(dict(dev))
            loss = PF.cross_entropy(logits, dev["segment"], -1)
        (loss * scale).backward()
        g = {n: p.grad.detach().float().cpu() / scale for n, p in model.named_parameters()}
        assert all(torch.isfinite(v).all() for v in g.values())
        return float(loss.detach()), logits.detach().float().cpu(), g

    failures = []
    modes = (("fp32", None), ("bf16", torch.bfloat16), ("fp16", torch.float16)) if on_gpu else (("fp32", None),)
    for mode, dtype in modes:      # (the CPU stand-ins have no 16-bit kernels: fp32 only in the dry run)
        eng.load_state_dict(sd)                                    # running statistics back to the start
        loss, logits, ge = step(eng, dtype, 1024.0 if mode == "fp16" else 1.0)
        stages, worst = _stage_distances(ge, go, _spunet_stage_of)
        l_rel = abs(loss - lo) / abs(lo)
        agree = float((logits.argmax(1) == logits_o.argmax(1)).float().mean())
        lines.append(f"{mode}: loss engine {loss:.6f} oracle {lo:.6f} rel {l_rel:.2e}; logits rel_max {_rel_max(logits, logits_o):.3e} "
                     f"rel_fro {_rel_fro(logits, logits_o):.3e} argmax {agree:.4f}")
        if dtype is None:
            lines += [f"   {k:12s} {v:.3e}" for k, v in stages.items()]
            ok = l_rel < 1e-5 and _rel_max(logits, logits_o) < 2e-3 and max(stages.values()) < 1e-2
        else:
            eng.load_state_dict(sd)
            _, _, gt = step(_autocast_rounding_twin(eng, dtype), None, 1024.0 if mode == "fp16" else 1.0)      # the same loss scale: fp16 gradients underflow alike
            env, _ = _stage_distances(gt, go, _spunet_stage_of)
            lines.append("   stage        16-bit kernels vs fp32 oracle   fp32 kernels + autocast roundings vs fp32 oracle")
            lines += [f"   {k:12s} {v:.3e}                       {env[k]:.3e}" for k, v in stages.items()]
            ok = (l_rel < 2e-3 and _rel_max(logits, logits_o) < 8e-2 and _rel_fro(logits, logits_o) < 3e-2 and agree > 0.97
                  and all(stages[k] <= 1.1 * env[k] + 5e-3 for k in stages))
        lines += [f"   worst {r:.3e} (|g| {n_:.3e}) {name}" for r, n_, name in worst[:3]]
        if not ok:
            failures.append(lines[-(len(stages) + 5):])
    _report("fullsize_spunet_step.txt", lines)
    assert not failures, failures


def test_ptv3_base_b8_equals_the_sum_of_its_scenes(cuda):
    """the bench batch (BASELINE configs[2], 8 x 102400 voxels, PT-v3m1 base depths, bf16 autocast) forward AND
    backward without 627 s of oracle.  Scenes are independent units of the path (batch-prefixed keys, windows never cross
    scenes, per-batch rulebooks) once BatchNorm uses its running statistics, so in eval mode with the CE criterion
        logits_B8[scene i] = logits_B1(scene i)          N_valid * loss_B8 = sum_i n_valid_i * loss_B1(i)
        N_valid * grad_B8  = sum_i n_valid_i * grad_B1(i)                      (up to fp32 summation order / bf16 re-rounding)
    and scene 0's B = 1 run is pinned here against the fp32 CPU oracle (loss, logits, per-stage gradients: the bars of
    test_ptv3_base_train_step_gradients_vs_oracle).  Not covered by this identity: train-mode BatchNorm statistics and the Lovasz term
    (both couple the scenes; tools/fullsize_parity.py b8 -> profiles/r04_v_fullsize_b8_parity.txt holds that run against the oracle)."""
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic
    from pointcept_amd.segmentor import DefaultSegmentorV2

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    on_gpu = torch.device(cuda).type == "cuda"
    # the serialization depth is a fact of the whole batch (bit_length of the largest coordinate, structure.py:74) and the Hilbert order
    # depends on it: the identity needs scenes whose own depth equals the batch's -- the first eight of the bench seeds 1000.. that share
    # the most common depth (at full size: depth 8, seed 1003 reaches 9 and is skipped)
    cand = [synthetic.indoor_scene(1000 + i, _n(102400)) for i in range(16)]
    depths = [int(sc["grid_coord"].max()).bit_length() for sc in cand]
    common = max(set(depths), key=depths.count)
    scenes = [sc for sc, d in zip(cand, depths) if d == common][:8]
    assert len(scenes) == 8, depths
    del cand
    _batch_equals_the_sum_of_its_scenes(cuda, scenes, BASE, 20, f"PT-v3m1 base, 8 x {_n(102400)} voxels", "fullsize_ptv3_b8_linearity.txt")


def _batch_equals_the_sum_of_its_scenes(cuda, scenes, cfg, n_cls, title, report):
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic
    from pointcept_amd.segmentor import DefaultSegmentorV2

    on_gpu = torch.device(cuda).type == "cuda"
    orc_b, eng_b = _pair(cfg, seed=7)
    torch.manual_seed(1)
    orc = om.SegmentorV2(n_cls, 64, orc_b).eval()
    eng = DefaultSegmentorV2(n_cls, 64, eng_b)
    eng.seg_head.load_state_dict(orc.seg_head.state_dict())
    eng = eng.to(cuda).eval()

    def run(batch):
        eng.zero_grad(set_to_none=True)
        dev = synthetic.to_torch(batch, cuda)
        torch.manual_seed(3)
        with torch.autocast("cuda" if on_gpu else "cpu", dtype=torch.bfloat16, enabled=on_gpu):
            out = eng(dev)
        out["loss"].backward()
        n_valid = int((batch["segment"] >= 0).sum())
        return (float(out["loss"].detach()), out["seg_logits"].detach().float().cpu(), n_valid,
                {n: p.grad.detach().double().cpu() for n, p in eng.named_parameters()})

    l8, logits8, nv8, g8 = run(synthetic.collate(scenes))
    acc_l, acc_g, logits1, first = 0.0, None, [], None
    for sc in scenes:
        l1, lg1, nv1, g1 = run(synthetic.collate([sc]))
        if first is None:
            first = (l1, lg1, {n: g.clone() for n, g in g1.items()})
        acc_l += l1 * nv1
        acc_g = {n: g * nv1 for n, g in g1.items()} if acc_g is None else {n: acc_g[n] + g1[n] * nv1 for n in g1}
        logits1.append(lg1)
    logits1 = torch.cat(logits1)
    sum_g = {n: g / nv8 for n, g in acc_g.items()}
    stages, worst = _stage_distances(g8, sum_g, _stage_of)
    same_rows = float((logits8 == logits1).all(1).float().mean())
    l_rel = abs(l8 - acc_l / nv8) / abs(l8)
    lines = [f"{title}, eval-mode BatchNorm, CE, bf16 autocast: B = {len(scenes)} vs the n_valid-weighted sum of its B = 1 runs",
             f"loss B8 {l8:.7f}  sum {acc_l / nv8:.7f}  rel {l_rel:.2e}; logits rel_max {_rel_max(logits8, logits1):.3e}, rows bit-identical {same_rows:.4f}"]
    lines += [f"   {k:10s} {v:.3e}" for k, v in stages.items()]
    lines += [f"   worst {r:.3e} (|g| {n_:.3e}) {name}" for r, n_, name in worst[:3]]
    # scene 0 alone against the fp32 oracle
    host = {k: torch.from_numpy(v) for k, v in synthetic.collate([scenes[0]]).items()}
    torch.manual_seed(3)
    out_o = orc(dict(host))
    out_o["loss"].backward()
    go = {n: p.grad.detach().double() for n, p in orc.named_parameters()}
    o_stages, o_worst = _stage_distances(first[2], go, _stage_of)
    lo = float(out_o["loss"].detach())
    agree = float((first[1].argmax(1) == out_o["seg_logits"].argmax(1)).float().mean())
    lines += [f"scene 0 (B = 1) vs fp32 CPU oracle: loss {first[0]:.6f} / {lo:.6f} rel {abs(first[0] - lo) / abs(lo):.2e}; logits rel_max "
              f"{_rel_max(first[1], out_o['seg_logits']):.3e} rel_fro {_rel_fro(first[1], out_o['seg_logits']):.3e} argmax {agree:.4f}"]
    lines += [f"   {k:10s} {v:.3e}" for k, v in o_stages.items()]
    _report(report, lines)
    assert l_rel < 1e-5, lines[1]
    assert _rel_max(logits8, logits1) < 1e-2, lines[1]          # measured: see profiles/r05_*_fullsize_ptv3_b8_linearity.txt
    assert max(stages.values()) < 1e-2, stages
    assert abs(first[0] - lo) < 5e-3 * abs(lo)
    assert _rel_fro(first[1], out_o["seg_logits"]) < 3e-2 and agree > 0.97
    assert max(o_stages.values()) < (6e-2 if on_gpu else 5e-3), o_stages


def test_ptv3_outdoor_batch_equals_the_sum_of_its_scenes(cuda):
    """BASELINE configs[4] (outdoor LiDAR, ~180 k voxels per scene, depth-12 grid: nuscenes/semseg-pt-v3m1-0-base.py:16,122
    -- in_channels 4, 16 classes) forward AND backward in the driver-run suite, by the linearity identity of the indoor test above: in eval
    mode with CE a batch of outdoor sweeps equals the n_valid-weighted sum of its scenes run alone (logits row by row, loss, every stage's
    gradient), and scene 0 alone is pinned to the fp32 CPU oracle -- loss, logits, per-stage gradients.  The scenes are the bench's own
    generator (`synthetic.outdoor_scene(seed, azimuth_steps=3300)`), picked to share one serialization depth."""
    from pointcept_amd import synthetic

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    full = SCALE >= 1.0
    cand = [synthetic.outdoor_scene(5000 + i, azimuth_steps=3300) if full else synthetic.outdoor_scene(5000 + i, _n(180000)) for i in range(5)]
    depths = [int(sc["grid_coord"].max()).bit_length() for sc in cand]
    common = max(set(depths), key=depths.count)
    scenes = [sc for sc, d in zip(cand, depths) if d == common][:2]
    assert len(scenes) == 2, depths
    n0 = scenes[0]["grid_coord"].shape[0]
    assert not full or (n0 >= 150000 and common >= 12), (n0, common)
    del cand
    _batch_equals_the_sum_of_its_scenes(cuda, scenes, dict(BASE, in_channels=4), 16, f"PT-v3m1 base outdoor (in_channels 4, 16 classes), 2 x ~{n0} voxels, depth {common}",
                                        "fullsize_ptv3_outdoor_linearity.txt")

