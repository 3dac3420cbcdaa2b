"""-m "not gpu": the engine's PYTHON layer on CPU stand-ins of the ops (tests/mock_backend.py) against the oracle.
What this tier checks is the host logic -- module wiring and state-dict names, PointSequential dispatch, autograd
wrappers (which map feeds which backward gather), the gather tables folded into the qkv / proj GEMMs, host-fact
prefetching of the pooled level sizes, duplicate-voxel bookkeeping, the enc_mode parent chain, PDNorm, criteria.
Kernels are NOT involved (the stand-ins are oracle code); they are tested by the -m gpu tier on the real library.
"""
import numpy as np
import pytest
import torch

import mock_backend

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")
TINY = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 1, 1), dec_depths=(1, 1, 1, 1), enc_patch_size=(128,) * 5,
            dec_patch_size=(128,) * 4, drop_path=0.0, shuffle_orders=True)


# every sys.modules entry pointcept_amd.compat.install() writes: B3 tests snapshot and restore them all, so that a later test that
# imports a reference file afresh binds to the oracle's third-party stand-ins again
COMPAT_NAMES = ["spconv", "spconv.pytorch", "spconv.pytorch.modules", "flash_attn", "torch_scatter", "pointops", "pointops2",
                "pointops2.pointops", "pointops2.functions", "pointops2.functions.pointops", "pointrope"]


def _rel(a, b):
    return float((a.detach().float() - b.detach().float()).abs().max() / b.detach().float().abs().max().clamp(min=1e-12))


def _grad_check(eng, orc, tol):
    go = dict(orc.named_parameters())
    nmax = max(float(p.grad.norm()) for p in go.values() if p.grad is not None)
    for name, p in eng.named_parameters():
        r = go[name].grad
        assert (p.grad is None) == (r is None), name
        if r is None:
            continue
        dn, rn = float((p.grad - r).norm()), float(r.norm())
        assert dn <= (tol * rn if rn > 1e-6 * nmax else 1e-5 * nmax), (name, dn, rn)


def _batch(sizes, seed0=200):
    from pointcept_amd import synthetic

    b = synthetic.collate([synthetic.indoor_scene(seed0 + i, n) for i, n in enumerate(sizes)])
    return {k: torch.from_numpy(v) for k, v in b.items()}


@pytest.mark.parametrize("variant", ["flash", "dense_rpe", "enc_mode", "no_prefetch"])
def test_ptv3_python_layer_matches_oracle(variant, monkeypatch):
    from oracle import ptv3_model as om
    from pointcept_amd import config
    from pointcept_amd.point_transformer_v3 import PointTransformerV3
    from pointcept_amd.segmentor import DefaultSegmentorV2

    cfg = dict(TINY)
    if variant == "dense_rpe":
        cfg.update(enable_flash=False, enable_rpe=True, upcast_attention=True, upcast_softmax=True)
    if variant == "enc_mode":
        cfg = {k: v for k, v in cfg.items() if not k.startswith("dec_")}
        cfg.update(enc_mode=True)
    if variant == "no_prefetch":
        monkeypatch.setattr(config, "PREFETCH_LEVELS", False)
    width = 992 if variant == "enc_mode" else 64
    batch = _batch([700, 180, 333])
    with mock_backend.cpu_ops():
        torch.manual_seed(0)
        orc_b, eng_b = om.PointTransformerV3(**cfg), PointTransformerV3(**cfg)
        assert list(orc_b.state_dict().keys()) == list(eng_b.state_dict().keys())
        sd = om.deterministic_state_dict(orc_b, 21)
        orc_b.load_state_dict(sd)
        eng_b.load_state_dict(sd)
        torch.manual_seed(1)
        orc, eng = om.SegmentorV2(20, width, orc_b), DefaultSegmentorV2(20, width, eng_b, criteria=("ce", "lovasz"))
        eng.seg_head.load_state_dict(orc.seg_head.state_dict())
        orc.train()
        eng.train()
        torch.manual_seed(9)
        oo = orc(dict(batch))
        torch.manual_seed(9)
        oe = eng(dict(batch), return_point=True)
        lov, _ = mock_backend.olosses.lovasz_softmax(oo["seg_logits"].detach().numpy(), batch["segment"].numpy(), -1)
        assert abs(float(oe["loss"]) - (float(oo["loss"]) + lov)) <= 1e-3 * abs(float(oo["loss"]) + lov)
        pt = oe["point"]
        assert pt.feat.shape[0] == batch["coord"].shape[0] and "pooling_parent" not in pt.keys()
        if variant != "enc_mode":
            assert torch.equal(pt.serialized_order, eng_b_order(orc_b, batch))     # same keys, same stable sort
        # gradients: CE part only on the oracle side, so compare against an oracle loss with the Lovasz term added
        lo = oo["loss"] + LovaszOnOracle.apply(oo["seg_logits"], batch["segment"])
        lo.backward()
        oe["loss"].backward()
        _grad_check(eng, orc, 3e-2)   # flash branch: gradients cross bf16 tensors (2^-8 rounding per element, run-to-run
                                      # CPU summation order flips those roundings); structural errors are >> 3 %


def eng_b_order(orc_backbone, batch):
    from oracle import ptv3_model as om

    p = om.Point({k: v for k, v in batch.items()})
    torch.manual_seed(9)
    p.serialization(order=ORDERS, shuffle_orders=True)
    return p.serialized_order


class LovaszOnOracle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        loss, d = mock_backend.olosses.lovasz_softmax(logits.detach().numpy(), target.numpy(), -1)
        ctx.save_for_backward(torch.from_numpy(d).float())
        return torch.tensor(loss, dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        return ctx.saved_tensors[0] * g, None


def test_spunet_python_layer_with_duplicate_voxels_matches_oracle():
    from oracle import ptv3_model as om
    from oracle import spunet_model as osp
    from pointcept_amd import functional as PF
    from pointcept_amd import synthetic
    from pointcept_amd.sparse_unet import SpUNetBase

    cfg = dict(base_channels=16, channels=(16, 32, 32, 48, 48, 32, 32, 16), layers=(1, 1, 1, 1, 1, 1, 1, 1))
    a, b = synthetic.indoor_scene(301, 900), synthetic.indoor_scene(302, 500)
    mixed = {k: np.concatenate([a[k], b[k]]) for k in a}
    assert len(mixed["grid_coord"]) > len(np.unique(mixed["grid_coord"], axis=0))
    batch = {k: torch.from_numpy(v) for k, v in synthetic.collate([mixed, synthetic.indoor_scene(303, 200)]).items()}
    with mock_backend.cpu_ops():
        orc, eng = osp.SpUNetBase(6, 20, **cfg), SpUNetBase(6, 20, **cfg)
        sd = om.deterministic_state_dict(orc, 22)
        orc.load_state_dict(sd)
        eng.load_state_dict(sd)
        orc.train()
        eng.train()
        lo_logits, le_logits = orc(dict(batch)), eng(dict(batch))
        assert _rel(le_logits, lo_logits) < 1e-4
        torch.nn.functional.cross_entropy(lo_logits, batch["segment"], ignore_index=-1).backward()
        PF.cross_entropy(le_logits, batch["segment"], -1).backward()
        _grad_check(eng, orc, 2e-3)      # exact adjoint of the many-to-one maps (merge / zero corrections), fp32


def test_b3_modules_refuse_cpu_again_after_the_context():
    from pointcept_amd import ops
    from pointcept_amd._lib import PtcoreError

    with mock_backend.cpu_ops():
        assert ops.gather_rows is mock_backend.gather_rows
    with pytest.raises(PtcoreError, match="no CPU fallback"):
        ops.gather_rows(torch.zeros(4, 8), torch.zeros(4, dtype=torch.long))


@pytest.mark.needs_reference
def test_pdnorm_model_matches_the_reference_model():
    """PPT configuration (pdnorm_bn + pdnorm_ln, decoupled, adaptive): the engine model on the CPU stand-ins against the
    REFERENCE model file itself (on oracle/shims.py), same weights, `condition` / `context` inputs -- forward, loss-free
    scalar objective, every gradient (incl. the per-condition norm layers that were NOT selected: no gradient on both sides)."""
    from oracle import ptv3_model as om
    from oracle import ref_import
    from pointcept_amd.point_transformer_v3 import PointTransformerV3

    R = ref_import.load()
    cfg = dict(TINY, pdnorm_bn=True, pdnorm_ln=True, pdnorm_decouple=True, pdnorm_adaptive=True,
               pdnorm_conditions=("ScanNet", "S3DIS", "Structured3D"))
    batch = _batch([600, 250], seed0=400)
    batch["condition"] = "S3DIS"
    batch["context"] = torch.randn(1, 256, generator=torch.Generator().manual_seed(5))
    with mock_backend.cpu_ops():
        torch.manual_seed(0)
        ref, eng = R["ptv3"].PointTransformerV3(**cfg), PointTransformerV3(**cfg)
        assert list(ref.state_dict().keys()) == list(eng.state_dict().keys())
        sd = om.deterministic_state_dict(ref, 23)
        ref.load_state_dict(sd)
        eng.load_state_dict(sd)
        outs = []
        for net in (ref, eng):
            net.train()
            torch.manual_seed(9)
            feat = net(dict(batch)).feat
            (feat * torch.linspace(-1, 1, feat.shape[1])).pow(2).mean().backward()
            outs.append(feat.detach())
        assert _rel(outs[1], outs[0]) < 1e-3
        _grad_check(eng, ref, 3e-2)
        unused = [k for k, p in eng.named_parameters() if ".norm.0." in k and "enc0.block0.norm1" in k]
        assert unused and all(dict(eng.named_parameters())[k].grad is None for k in unused)      # "ScanNet" layers were not used


@pytest.mark.needs_reference
def test_b3_reference_model_files_run_unmodified_on_the_engine_operators():
    """SURVEY 8(b) B3 end to end: the REFERENCE's own files (structure.py, modules.py, point_transformer_v3m1_base.py,
    spconv_unet_v1m1_base.py) are imported a second time with `spconv.pytorch`, `flash_attn` and `torch_scatter` bound to
    what pointcept_amd.compat.install() provides (ops on the CPU stand-ins), and compared with the same files running
    on the oracle's third-party stand-ins: same outputs, same gradients.  Nothing of the engine's own model code is
    involved -- only its operator-level API."""
    import importlib
    import sys

    import pointcept_amd.compat as compat
    from oracle import ptv3_model as om
    from oracle import ref_import

    R = ref_import.load()                                        # reference files on oracle/shims.py
    R_m2 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m2_sonata")
    R_m3 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m3_utonia")
    names = COMPAT_NAMES + [
             "pointcept.models.utils", "pointcept.models.utils.structure", "pointcept.models.utils.misc",
             "pointcept.models.utils.serialization", "pointcept.models.modules", "pointcept.models.builder",
             "pointcept.models.point_transformer_v3.point_transformer_v3m1_base",
             "pointcept.models.point_transformer_v3.point_transformer_v3m2_sonata",
             "pointcept.models.point_transformer_v3.point_transformer_v3m3_utonia",
             "pointcept.models.sparse_unet.spconv_unet_v1m1_base"]
    names += [k for k in list(sys.modules) if k.startswith("pointcept.models.utils.serialization.")]
    saved = {k: sys.modules.pop(k, None) for k in names}
    try:
        compat.install(force=True)                               # engine operator API under the third-party names
        E = dict(ptv3=importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m1_base"),
                 spunet=importlib.import_module("pointcept.models.sparse_unet.spconv_unet_v1m1_base"))
        E_m2 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m2_sonata")
        E_m3 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m3_utonia")
        assert E["ptv3"] is not R["ptv3"] and E["ptv3"].spconv.__name__ == "pointcept_amd.spconv_api"
        assert E_m2 is not R_m2 and E_m2.spconv.__name__ == "pointcept_amd.spconv_api"
        with mock_backend.cpu_ops():
            # PT-v3m2 (Sonata; GridPooling / GridUnpooling, LayerScale): a SURVEY 8(f).2 model family the engine has no
            # module-level port of -- its reference file runs through the operator-level API as is
            m2cfg = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 1, 1), dec_depths=(1, 1, 1, 1),
                         enc_patch_size=(128,) * 5, dec_patch_size=(128,) * 4, drop_path=0.0, shuffle_orders=False,
                         layer_scale=0.5)
            mb = _batch([450, 200], seed0=520)
            mb["grid_size"] = 0.02
            torch.manual_seed(0)
            m_a, m_b = R_m2.PointTransformerV3(**m2cfg), E_m2.PointTransformerV3(**m2cfg)
            sd = om.deterministic_state_dict(m_a, 26)
            m_a.load_state_dict(sd)
            m_b.load_state_dict(sd)
            fm = []
            for net in (m_a, m_b):
                net.train()
                torch.manual_seed(9)
                f = net({k: v for k, v in mb.items()}).feat
                (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean().backward()
                fm.append(f.detach())
            assert _rel(fm[1], fm[0]) < 1e-3
            _grad_check(m_b, m_a, 3e-2)
            # PT-v3m3 (Utonia; 3-D RoPE inside attention needs head_dim % 3 == 0): flash_attn_varlen_qkvpacked_func then
            # takes the head_dim 17..64 kernels (here: their CPU stand-in) instead of the head_dim-16 ones
            m3cfg = dict(m2cfg, enc_channels=(48, 96, 96, 192, 192), enc_num_head=(2, 4, 4, 8, 8), dec_channels=(48, 96, 96, 192),
                         dec_num_head=(2, 4, 4, 8))
            m3cfg.pop("layer_scale")
            torch.manual_seed(0)
            u_a, u_b = R_m3.PointTransformerV3(**m3cfg), E_m3.PointTransformerV3(**m3cfg)
            sd = om.deterministic_state_dict(u_a, 27)
            u_a.load_state_dict(sd)
            u_b.load_state_dict(sd)
            fu = []
            for net in (u_a, u_b):
                net.train()
                torch.manual_seed(9)
                f = net({k: v for k, v in mb.items()}).feat
                (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean().backward()
                fu.append(f.detach())
            assert _rel(fu[1], fu[0]) < 2e-2             # bf16 attention vs the fp32-math stand-in
            _grad_check(u_b, u_a, 6e-2)
            cfg = dict(TINY, enable_flash=True)
            batch = _batch([500, 220], seed0=500)
            torch.manual_seed(0)
            a, b = R["ptv3"].PointTransformerV3(**cfg), E["ptv3"].PointTransformerV3(**cfg)
            sd = om.deterministic_state_dict(a, 24)
            a.load_state_dict(sd)
            b.load_state_dict(sd)
            feats = []
            for net in (a, b):
                net.train()
                torch.manual_seed(9)
                f = net({k: v for k, v in batch.items()}).feat
                (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean().backward()
                feats.append(f.detach())
            assert _rel(feats[1], feats[0]) < 1e-3
            _grad_check(b, a, 3e-2)
            scfg = dict(base_channels=16, channels=(16, 32, 32, 48, 48, 32, 32, 16), layers=(1, 1, 1, 1, 1, 1, 1, 1))
            c, d = R["spunet"].SpUNetBase(6, 20, **scfg), E["spunet"].SpUNetBase(6, 20, **scfg)
            sd = om.deterministic_state_dict(c, 25)
            c.load_state_dict(sd)
            d.load_state_dict(sd)
            outs = []
            for net in (c, d):
                net.train()
                o = net({k: v for k, v in batch.items()})
                torch.nn.functional.cross_entropy(o, batch["segment"], ignore_index=-1).backward()
                outs.append(o.detach())
            assert _rel(outs[1], outs[0]) < 1e-4
            _grad_check(d, c, 2e-3)
            # the no-skip variant of the same reference file (spconv_unet_v1m1_base.py:283-463)
            c, d = R["spunet"].SpUNetNoSkipBase(6, 20, **scfg), E["spunet"].SpUNetNoSkipBase(6, 20, **scfg)
            assert list(c.state_dict().keys()) == list(d.state_dict().keys())
            sd = om.deterministic_state_dict(c, 26)
            c.load_state_dict(sd)
            d.load_state_dict(sd)
            outs = []
            for net in (c, d):
                net.train()
                o = net({k: v for k, v in batch.items()})
                torch.nn.functional.cross_entropy(o, batch["segment"], ignore_index=-1).backward()
                outs.append(o.detach())
            assert _rel(outs[1], outs[0]) < 1e-4
            _grad_check(d, c, 2e-3)
    finally:
        for k in names:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


@pytest.mark.needs_reference
def test_b3_reference_litept_file_runs_unmodified_on_the_engine_operators(monkeypatch):
    """B3 for LitePT: the REFERENCE's own litept_v1.py imported with `spconv.pytorch`, `flash_attn`, `torch_scatter` AND `pointrope`
    bound to pointcept_amd.compat.install() (its PointROPE_func then calls the engine's `pointrope.pointrope`, its attention hands
    fp16 rows to the engine's flash_attn mirror), against the same file on the oracle's stand-ins + libs/pointrope's own CPU kernel."""
    import importlib
    import sys
    import types

    import pointcept_amd.compat as compat
    from oracle import ptv3_model as om

    R = _import_reference_litept(monkeypatch)                    # reference file on oracle/shims.py + pointrope_cpu
    names = COMPAT_NAMES + ["pointcept.models.utils", "pointcept.models.utils.structure", "pointcept.models.utils.misc",
             "pointcept.models.utils.serialization", "pointcept.models.modules", "pointcept.models.builder",
             "pointcept.models.litept", "pointcept.models.litept.litept_v1"]
    names += [k for k in list(sys.modules) if k.startswith("pointcept.models.utils.serialization.")]
    saved = {k: sys.modules.pop(k, None) for k in names}
    try:
        compat.install(force=True)
        pkg = types.ModuleType("pointcept.models.litept")
        pkg.__path__ = [saved["pointcept.models.litept"].__path__[0]]
        sys.modules["pointcept.models.litept"] = pkg
        E = importlib.import_module("pointcept.models.litept.litept_v1")
        assert E is not R and E.spconv.__name__ == "pointcept_amd.spconv_api" and hasattr(E, "PointROPE_func")
        assert E._kernels.__name__ == "pointcept_amd.pointrope_api"
        cfg = dict(LITEPT_TINY)
        torch.manual_seed(0)
        a, b = R.LitePT(**cfg), E.LitePT(**cfg)
        sd = om.deterministic_state_dict(a, 44)
        a.load_state_dict(sd)
        b.load_state_dict(sd)
        mb = _batch([600, 240], seed0=640)
        mb["grid_size"] = 0.02
        feats = []
        with mock_backend.cpu_ops():
            for net in (a, b):
                net.train()
                torch.manual_seed(9)
                f = net({k: v for k, v in mb.items()}).feat
                (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean().backward()
                feats.append(f.detach())
        assert _rel(feats[1], feats[0]) < 1e-2            # fp16 rows re-rounded to bf16 inside the engine's attention mirror
        _grad_check(b, a, 6e-2)
    finally:
        for k in names:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


@pytest.mark.needs_reference
def test_b3_reference_spunet_variants_run_unmodified_on_the_engine_operators():
    """B3 for the other SparseUNet files of the reference -- spconv_unet_v1m2_bn_momentum.py ("SpUNet-v1m2") and
    spconv_unet_v1m3_pdnorm.py ("SpUNet-v1m3": prompt-driven BatchNorm, modules called with [tensor, condition, context] lists) --
    imported with `spconv.pytorch` bound to pointcept_amd.compat.install(), against the same files on the oracle's stand-ins."""
    import importlib
    import sys

    import pointcept_amd.compat as compat
    from oracle import ptv3_model as om
    from oracle import ref_import

    ref_import.load()
    mods = ["pointcept.models.sparse_unet.spconv_unet_v1m2_bn_momentum", "pointcept.models.sparse_unet.spconv_unet_v1m3_pdnorm"]
    for m in mods:
        sys.modules.pop(m, None)
    builder = sys.modules["pointcept.models.builder"]
    for name in ("SpUNet-v1m2", "SpUNet-v1m3"):
        builder.MODELS._module_dict.pop(name, None)
    R = [importlib.import_module(m) for m in mods]
    names = COMPAT_NAMES + ["pointcept.models.builder"] + mods
    saved = {k: sys.modules.pop(k, None) for k in names}
    try:
        compat.install(force=True)
        E = [importlib.import_module(m) for m in mods]
        assert all(e is not r and e.spconv.__name__ == "pointcept_amd.spconv_api" for e, r in zip(E, R))
        batch = _batch([500, 220], seed0=650)
        scfg = dict(base_channels=16, channels=(16, 32, 32, 48, 48, 32, 32, 16), layers=(1, 1, 1, 1, 1, 1, 1, 1))
        cases = [(0, dict(in_channels=6, num_classes=13, bn_momentum=0.02, **scfg), {}),
                 (1, dict(in_channels=6, num_classes=13, context_channels=32, zero_init=False, **scfg),
                  dict(condition="S3DIS", context=torch.randn(1, 32, generator=torch.Generator().manual_seed(2))))]
        with mock_backend.cpu_ops():
            for mi, cfg, extra in cases:
                torch.manual_seed(0)
                a, b = R[mi].SpUNetBase(**cfg), E[mi].SpUNetBase(**cfg)
                assert list(a.state_dict().keys()) == list(b.state_dict().keys())
                sd = om.deterministic_state_dict(a, 45 + mi)
                a.load_state_dict(sd)
                b.load_state_dict(sd)
                outs = []
                for net in (a, b):
                    net.train()
                    o = net({**{k: v for k, v in batch.items()}, **extra})
                    torch.nn.functional.cross_entropy(o, batch["segment"] % 13, ignore_index=-1).backward()
                    outs.append(o.detach())
                assert outs[0].shape == (720, 13) and _rel(outs[1], outs[0]) < 1e-4, mods[mi]
                _grad_check(b, a, 2e-3)
    finally:
        for k in names:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


@pytest.mark.needs_reference
def test_b3_reference_ptv1_ptv2_files_run_unmodified_on_the_pointops_mirror(monkeypatch):
    """B3 for libs/pointops' callers: the REFERENCE's own point_transformer_seg.py ("PointTransformer-Seg26": farthest point sampling,
    kNN grouping with relative coordinates, inverse-distance interpolation) and point_transformer_v2m2_base.py ("PT-v2m2": kNN,
    grouping, torch_scatter pooling) imported with `pointops` / `torch_scatter` bound to pointcept_amd.compat.install(), against the
    same files on the reference's own pointops python package (oracle/pointops_c.py stand-ins for its compiled kernels) and the
    oracle's torch_scatter stand-in: same logits, same gradients."""
    import importlib
    import sys
    import types

    import pointcept_amd.compat as compat
    from oracle import pointops_c, ref_import
    from oracle import ptv3_model as om

    ref_import.load()
    P = pointops_c.load_reference_package()
    # point_transformer_seg.py:100 builds its offsets with torch.cuda.IntTensor: on this GPU-less container the constructor is pointed
    # at the CPU equivalent (the file itself stays untouched)
    monkeypatch.setattr(torch.cuda, "IntTensor", lambda v: torch.tensor(v, dtype=torch.int32), raising=False)
    mods = ["pointcept.models.point_transformer.point_transformer_seg", "pointcept.models.point_transformer_v2.point_transformer_v2m2_base"]
    pkgs = {"pointcept.models.point_transformer": "/pointcept/models/point_transformer",
            "pointcept.models.point_transformer_v2": "/pointcept/models/point_transformer_v2"}

    def fresh_import():
        for k in list(pkgs) + mods + ["pointcept.models.point_transformer.utils"]:
            sys.modules.pop(k, None)
        for k, rel in pkgs.items():
            pk = types.ModuleType(k)
            pk.__path__ = [ref_import.REF + rel]
            sys.modules[k] = pk
        reg = sys.modules["pointcept.models.builder"].MODELS._module_dict
        for name in ("PointTransformer-Seg26", "PointTransformer-Seg38", "PointTransformer-Seg50", "PT-v2m2"):
            reg.pop(name, None)
        return [importlib.import_module(m) for m in mods]

    names = COMPAT_NAMES + ["pointcept.models.builder"]
    saved = {k: sys.modules.get(k) for k in names + list(pkgs) + mods + ["pointcept.models.point_transformer.utils"]}
    try:
        sys.modules["pointops"] = P                                # reference side: its own python package
        R = fresh_import()
        sys.modules.pop("pointcept.models.builder", None)
        compat.install(force=True)                                 # engine side: the mirrors
        sys.modules["pointcept.models.builder"] = saved["pointcept.models.builder"]
        E = fresh_import()
        assert E[0].pointops.__name__ == "pointcept_amd.pointops_api" and R[0].pointops is P
        assert E[1].segment_csr.__module__ == "pointcept_amd.torch_scatter_api"
        g = torch.Generator().manual_seed(8)
        n_pts = 2600                                            # PTv1 keeps 1 / 256 of the points in its last stage
        coord = torch.rand(n_pts, 3, generator=g) * torch.tensor([2.0, 2.0, 1.0])
        data = dict(coord=coord, feat=torch.cat([coord, torch.rand(n_pts, 3, generator=g)], 1), offset=torch.tensor([1500, n_pts]),
                    segment=torch.randint(0, 13, (n_pts,), generator=g))
        v2cfg = dict(in_channels=6, num_classes=13, patch_embed_depth=1, patch_embed_channels=12, patch_embed_groups=3,
                     patch_embed_neighbours=8, enc_depths=(1, 1), enc_channels=(24, 48), enc_groups=(3, 6), enc_neighbours=(8, 8),
                     dec_depths=(1, 1), dec_channels=(12, 24), dec_groups=(3, 3), dec_neighbours=(8, 8), grid_sizes=(0.25, 0.5))
        builders = [(lambda m: m.PointTransformerSeg26(in_channels=6, num_classes=13)), (lambda m: m.PointTransformerV2(**v2cfg))]
        with mock_backend.cpu_ops():
            for mi, build in enumerate(builders):
                torch.manual_seed(0)
                a, b = build(R[mi]), build(E[mi])
                assert list(a.state_dict().keys()) == list(b.state_dict().keys())
                sd = om.deterministic_state_dict(a, 47 + mi)
                a.load_state_dict(sd)
                b.load_state_dict(sd)
                outs = []
                for net in (a, b):
                    net.train()
                    o = net({k: v.clone() for k, v in data.items()})
                    torch.nn.functional.cross_entropy(o, data["segment"]).backward()
                    outs.append(o.detach())
                assert outs[0].shape == (n_pts, 13) and _rel(outs[1], outs[0]) < 1e-4, mods[mi]
                ga = dict(a.named_parameters())
                gmax = max(float(p.grad.norm()) for p in ga.values())
                for name, p in b.named_parameters():        # (biases in front of a BatchNorm have a zero gradient: fp32 noise on both sides)
                    assert float((p.grad - ga[name].grad).norm()) <= 2e-3 * float(ga[name].grad.norm()) + 2e-5 * gmax, (mods[mi], name)
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


@pytest.mark.needs_reference
def test_b3_reference_oacnns_file_runs_unmodified_on_the_engine_operators():
    """B3 for another spconv caller of the reference: oacnns_v1m1_base.py ("OACNNs": three stem convolutions under one indice_key,
    strided down / inverse up convolutions, k = 1 head, torch_geometric voxel pooling around them) on the engine's spconv mirror."""
    import importlib
    import sys
    import types

    import pointcept_amd.compat as compat
    from oracle import ptv3_model as om
    from oracle import ref_import

    ref_import.load()
    mod = "pointcept.models.oacnns.oacnns_v1m1_base"

    def fresh_import():
        for k in ("pointcept.models.oacnns", mod):
            sys.modules.pop(k, None)
        pk = types.ModuleType("pointcept.models.oacnns")
        pk.__path__ = [ref_import.REF + "/pointcept/models/oacnns"]
        sys.modules["pointcept.models.oacnns"] = pk
        sys.modules["pointcept.models.builder"].MODELS._module_dict.pop("OACNNs", None)
        return importlib.import_module(mod)

    names = COMPAT_NAMES + ["pointcept.models.oacnns", mod]
    saved = {k: sys.modules.get(k) for k in names}
    try:
        R = fresh_import()
        compat.install(force=True)
        E = fresh_import()
        assert E is not R and E.spconv.__name__ == "pointcept_amd.spconv_api" and R.spconv.__name__ != E.spconv.__name__
        cfg = dict(in_channels=6, num_classes=13, embed_channels=16, enc_num_ref=[4, 4], enc_channels=[16, 32], groups=[2, 4], enc_depth=[1, 1],
                   down_ratio=[2, 2], dec_channels=[16, 32], point_grid_size=[[4, 8], [2, 4]], dec_depth=[1, 1])
        batch = _batch([500, 220], seed0=670)
        torch.manual_seed(0)
        a, b = R.OACNNs(**cfg), E.OACNNs(**cfg)
        assert list(a.state_dict().keys()) == list(b.state_dict().keys())
        sd = om.deterministic_state_dict(a, 50)
        a.load_state_dict(sd)
        b.load_state_dict(sd)
        outs = []
        with mock_backend.cpu_ops():
            for net in (a, b):
                net.train()
                o = net({k: v for k, v in batch.items()})
                torch.nn.functional.cross_entropy(o, batch["segment"] % 13, ignore_index=-1).backward()
                outs.append(o.detach())
        assert outs[0].shape == (720, 13) and _rel(outs[1], outs[0]) < 1e-4
        _grad_check(b, a, 2e-3)
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def test_physically_sorted_working_copy_and_restore():
    """Point.physically_sorted re-expresses every per-point tensor and all k serialization maps in the row order of the
    first curve; restore_order brings features back (and its backward routes gradients to the caller's rows)."""
    from pointcept_amd.structure import Point

    batch = _batch([400, 150], seed0=600)
    with mock_backend.cpu_ops():
        p = Point({k: v for k, v in batch.items()})
        p["feat"] = p.feat.clone().requires_grad_(True)
        torch.manual_seed(3)
        p.serialization(order=ORDERS, shuffle_orders=True)
        w = p.physically_sorted()
        pi = p.serialized_order[0]
        assert torch.equal(w.feat.detach(), p.feat.detach()[pi]) and torch.equal(w.grid_coord, p.grid_coord[pi])
        assert torch.equal(w.batch, p.batch[pi]) and torch.equal(w.offset, p.offset)
        assert torch.equal(w.serialized_order[0], torch.arange(pi.numel()))          # rows ARE the first curve now
        for k in range(len(ORDERS)):
            codes = w.serialized_code[k][w.serialized_order[k]]
            assert bool((codes[1:] >= codes[:-1]).all())
            assert torch.equal(w.serialized_inverse[k][w.serialized_order[k]], torch.arange(pi.numel()))
            assert torch.equal(w.serialized_code[k], p.serialized_code[k][pi])
        w["feat"] = w.feat * 2.0
        back = w.restore_order(p)
        assert torch.allclose(back.feat, p.feat * 2.0)
        probe = torch.randn_like(back.feat)
        (back.feat * probe).sum().backward()
        assert torch.allclose(p.feat.grad, 2.0 * probe)


@pytest.mark.needs_reference
def test_gridsample_python_layer_against_the_reference_transform():
    """pointcept_amd.transform.GridSample (dict protocol, index_valid_keys bookkeeping, sampled_index, inverse / grid_coord /
    min_coord / displacement, test mode) on the CPU stand-ins vs the reference transform (numpy) on the same cloud.  Which
    point represents a voxel is random on both sides, so pick-dependent outputs are compared through the voxel they belong to."""
    from oracle import ref_import
    from pointcept_amd.transform import GridSample

    tr = ref_import.load_transform()
    rng = np.random.default_rng(31)
    n = 3000
    coord = ((rng.random((n, 3)) - 0.4) * 1.5).astype(np.float32)
    normal = rng.standard_normal((n, 3)).astype(np.float32)
    segment = rng.integers(0, 20, n)
    sampled = np.sort(rng.choice(n, 40, replace=False))
    kw = dict(grid_size=0.05, hash_type="fnv", return_inverse=True, return_grid_coord=True, return_min_coord=True,
              return_displacement=True)    # (project_displacement with `normal` among the indexed keys raises in the reference itself)
    keys = ["coord", "normal", "segment"]
    np.random.seed(0)
    ref = tr.GridSample(mode="train", **kw)(dict(coord=coord.copy(), normal=normal.copy(), segment=segment.copy(),
                                                 sampled_index=sampled.copy(), index_valid_keys=list(keys)))
    with mock_backend.cpu_ops():
        torch.manual_seed(0)
        eng = GridSample(mode="train", **kw)(dict(coord=torch.from_numpy(coord), normal=torch.from_numpy(normal),
                                                  segment=torch.from_numpy(segment), sampled_index=torch.from_numpy(sampled),
                                                  index_valid_keys=list(keys)))
        parts = GridSample(mode="test", grid_size=0.05, return_grid_coord=True)(
            dict(coord=torch.from_numpy(coord), segment=torch.from_numpy(segment), index_valid_keys=["coord", "segment"]))
    ref_parts = tr.GridSample(mode="test", grid_size=0.05, hash_type="fnv", return_grid_coord=True)(
        dict(coord=coord.copy(), segment=segment.copy(), index_valid_keys=["coord", "segment"]))
    assert eng["index_valid_keys"] == ref["index_valid_keys"]
    assert np.array_equal(eng["inverse"].numpy(), ref["inverse"])
    assert np.allclose(eng["min_coord"].numpy(), ref["min_coord"])
    # every voxel is represented; the labelled points (sampled_index) are all kept and re-indexed (transform.py:883-891)
    n_vox = int(ref["inverse"].max()) + 1
    for out in (ref, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in eng.items()}):
        vox_of_pick = set(map(tuple, out["grid_coord"]))
        assert len(vox_of_pick) == n_vox
        assert len(out["sampled_index"]) == len(sampled)
        assert out["displacement"].shape == (out["coord"].shape[0], 3) and np.abs(out["displacement"]).max() <= 0.5
    eng_np = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in eng.items()}
    assert set(map(tuple, eng_np["coord"][eng_np["sampled_index"]])) == set(map(tuple, coord[sampled]))
    assert set(map(tuple, ref["coord"][ref["sampled_index"]])) == set(map(tuple, coord[sampled]))
    for out in (eng_np, ref):                                           # one pick per voxel + the labelled points not picked anyway
        assert n_vox <= out["coord"].shape[0] <= n_vox + len(sampled)
    # test mode: as many parts as the fullest voxel holds points; part i takes the (i mod count)-th point of every voxel
    assert len(parts) == len(ref_parts)
    for pe, pr in zip(parts, ref_parts):
        assert pe["grid_coord"].shape == pr["grid_coord"].shape
        assert set(map(tuple, pe["grid_coord"].numpy())) == set(map(tuple, pr["grid_coord"]))
    seen = np.zeros(n, bool)
    for pe in parts:
        seen[pe["index"].numpy()] = True
    assert seen.all()


def test_cast_cache_lifetime_and_invalidation():
    """functional._CastCache: shadows die with their parameter, follow the version counter, survive address
    reuse, and `.data` writes (which do not move the counter) are covered by invalidate_weight_casts()."""
    import gc

    from pointcept_amd import functional as PF

    c = PF._CastCache(cuda_only=False)
    p = torch.nn.Parameter(torch.randn(4, 27, 8))
    v = p.view(4, 27, 8)
    a = c.get(v, torch.bfloat16)
    assert a.dtype == torch.bfloat16 and torch.equal(a, p.detach().to(torch.bfloat16)) and len(c.entries) == 1
    assert c.get(p.view(4, 27, 8), torch.bfloat16).data_ptr() == a.data_ptr()      # a fresh view object hits the same entry
    with torch.no_grad():
        p.mul_(2.0)                                                                  # optimizer-style update: version moves
    assert torch.equal(c.get(p, torch.bfloat16), p.detach().to(torch.bfloat16))
    p.data.mul_(0.5)                                                                 # .data write: version does NOT move
    assert not torch.equal(c.get(p, torch.bfloat16), p.detach().to(torch.bfloat16))  # documented limitation ...
    c.invalidate()
    assert torch.equal(c.get(p, torch.bfloat16), p.detach().to(torch.bfloat16))      # ... and its remedy
    q = torch.nn.Parameter(torch.randn(16, 16))
    c.get(q, torch.bfloat16)
    assert len(c.entries) == 2
    del p, v, a
    gc.collect()
    assert len(c.entries) == 1                                                       # the entry died with its parameter
    assert c.get(q[:8], torch.bfloat16).shape == (8, 16) and len(c.entries) == 1     # partial views are never cached


M2_TINY = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 1, 1), enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32),
               dec_depths=(1, 1, 1, 1), dec_channels=(64, 64, 128, 256), dec_num_head=(4, 4, 8, 16), enc_patch_size=(128,) * 5,
               dec_patch_size=(128,) * 4, drop_path=0.0, shuffle_orders=False)


@pytest.mark.needs_reference
@pytest.mark.parametrize("layer_scale", [None, 0.5])
def test_ptv3m2_module_port_matches_the_reference_file(layer_scale):
    """the engine's module-level PT-v3m2 (pointcept_amd/point_transformer_v3m2.py: GridPooling /
    GridUnpooling / LayerScale / Linear stem) against the REFERENCE's own point_transformer_v3m2_sonata.py on the oracle's
    third-party stand-ins: same state-dict keys and shapes, same features, same gradients, ragged two-scene batch."""
    import importlib

    from oracle import ptv3_model as om
    from oracle import ref_import
    from pointcept_amd.point_transformer_v3m2 import PointTransformerV3 as EngM2

    ref_import.load()
    R_m2 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m2_sonata")
    cfg = dict(M2_TINY, layer_scale=layer_scale)
    torch.manual_seed(0)
    ref, eng = R_m2.PointTransformerV3(**cfg), EngM2(**cfg)
    assert list(ref.state_dict().keys()) == list(eng.state_dict().keys())
    for (k, a), (_, b) in zip(ref.state_dict().items(), eng.state_dict().items()):
        assert a.shape == b.shape, k
    sd = om.deterministic_state_dict(ref, 31)
    ref.load_state_dict(sd)
    eng.load_state_dict(sd)
    mb = _batch([700, 260], seed0=610)
    mb["grid_size"] = 0.02
    feats = []
    with mock_backend.cpu_ops():
        for net in (ref, eng):
            net.train()
            torch.manual_seed(9)
            f = net({k: v for k, v in mb.items()}).feat
            (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean().backward()
            feats.append(f.detach())
    assert feats[0].shape == feats[1].shape == (960, 64)
    assert _rel(feats[1], feats[0]) < 1e-3
    _grad_check(eng, ref, 3e-2)


M3_TINY = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 1, 1), enc_channels=(48, 96, 96, 192, 192), enc_num_head=(2, 4, 4, 8, 8),
               dec_depths=(1, 1, 1, 1), dec_channels=(48, 96, 96, 192), dec_num_head=(2, 4, 4, 8), enc_patch_size=(128,) * 5,
               dec_patch_size=(128,) * 4, drop_path=0.0, shuffle_orders=False)


@pytest.mark.needs_reference
@pytest.mark.parametrize("variant", ["eval_default", "train_aug", "dec_rope_off", "rope_off"])
def test_ptv3m3_module_port_matches_the_reference_file(variant):
    """the engine's module-level PT-v3m3 (pointcept_amd/point_transformer_v3m3.py: m2 + Point3DRoPE on q / k from the
    continuous coordinates, head_dim 24) against the REFERENCE's own point_transformer_v3m3_utonia.py on the oracle's third-party
    stand-ins: same state-dict keys and shapes (incl. the rope.inv_freq buffers), same features, same gradients; the training-time
    shift / jitter / rescale of the rope coordinates consumes the RNG exactly as the reference does (same seed -> same features)."""
    import importlib

    from oracle import ptv3_model as om
    from oracle import ref_import
    from pointcept_amd.point_transformer_v3m3 import PointTransformerV3 as EngM3

    ref_import.load()
    R_m3 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m3_utonia")
    cfg = dict(M3_TINY, rope_base=10)
    if variant == "train_aug":
        cfg.update(shift_coords=0.5, jitter_coords=1.1, rescale_coords=1.2, layer_scale=0.5)
    elif variant == "dec_rope_off":
        cfg.update(dec_rope_enable=False)
    elif variant == "rope_off":
        cfg.update(rope_base=None)
    torch.manual_seed(0)
    ref, eng = R_m3.PointTransformerV3(**cfg), EngM3(**cfg)
    assert list(ref.state_dict().keys()) == list(eng.state_dict().keys())
    assert any(k.endswith("rope.inv_freq") for k in eng.state_dict()) == (variant != "rope_off")
    for (k, a), (_, b) in zip(ref.state_dict().items(), eng.state_dict().items()):
        assert a.shape == b.shape, k
        if k.endswith("inv_freq"):
            assert torch.equal(a, b), k
    sd = om.deterministic_state_dict(ref, 31)
    ref.load_state_dict(sd)
    eng.load_state_dict(sd)
    mb = _batch([700, 260], seed0=620)
    mb["grid_size"] = 0.02
    feats = []
    with mock_backend.cpu_ops():
        for net in (ref, eng):
            net.train(variant != "eval_default")
            torch.manual_seed(9)
            f = net({k: v for k, v in mb.items()}).feat
            (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean().backward()
            feats.append(f.detach())
    assert feats[0].shape == feats[1].shape == (960, 48)
    assert _rel(feats[1], feats[0]) < 1e-3
    _grad_check(eng, ref, 3e-2)


LITEPT_TINY = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 2, 1), enc_channels=(36, 72, 72, 144, 144), enc_num_head=(2, 4, 4, 8, 8),
                   enc_patch_size=(128,) * 5, dec_channels=(36, 72, 72, 144), dec_num_head=(2, 4, 4, 8), dec_patch_size=(128,) * 4,
                   drop_path=0.0, shuffle_orders=False)


def _import_reference_litept(monkeypatch):
    """the reference's litept_v1.py with `pointrope` = the reference's OWN pointrope_cpu (oracle/_ref, compiled from
    libs/pointrope/pointrope.cpp): its PointROPE_func / PointROPE classes (:27-59), not the torch fallback of :60-126"""
    import importlib
    import sys

    from oracle import build_ref, ref_import

    ref_import.load()
    if "pointcept.models.litept.litept_v1" in sys.modules:        # imported once per process: the file registers "LitePT-v1" at import
        mod = sys.modules["pointcept.models.litept.litept_v1"]
    else:
        import types
        monkeypatch.setitem(sys.modules, "pointrope", build_ref.load_pointrope())
        pkg = types.ModuleType("pointcept.models.litept")
        pkg.__path__ = [ref_import.REF + "/pointcept/models/litept"]
        sys.modules["pointcept.models.litept"] = pkg
        mod = importlib.import_module("pointcept.models.litept.litept_v1")
    assert hasattr(mod, "PointROPE_func"), "the reference fell back to its torch PointROPE: the pointrope module did not import"
    return mod


@pytest.mark.needs_reference
@pytest.mark.parametrize("variant", ["default_layout", "attn_everywhere_with_decoder"])
def test_litept_module_port_matches_the_reference_file(variant, monkeypatch):
    """the engine's module-level LitePT-v1 (pointcept_amd/litept.py) against the REFERENCE's own litept_v1.py on the
    oracle's third-party stand-ins, with libs/pointrope's own pointrope_cpu as the reference's rotary kernel: same state-dict keys
    and shapes, same features, same gradients.  default_layout = convolution stages then attention stages, un-pooling decoder;
    the second variant puts attention (and serialization) in every stage and blocks in the decoder."""
    from oracle import ptv3_model as om
    from pointcept_amd.litept import LitePT as EngLitePT

    R = _import_reference_litept(monkeypatch)
    cfg = dict(LITEPT_TINY)
    if variant == "attn_everywhere_with_decoder":
        cfg.update(enc_conv=(True, False, True, False, True), enc_attn=(True,) * 5, dec_depths=(1, 1, 1, 1), dec_conv=(True, False, True, False),
                   dec_attn=(True, True, False, True), enc_rope_freq=(100.0, 50.0, 100.0, 10.0, 100.0))
    torch.manual_seed(0)
    ref, eng = R.LitePT(**cfg), EngLitePT(**cfg)
    assert list(ref.state_dict().keys()) == list(eng.state_dict().keys())
    for (k, a), (_, b) in zip(ref.state_dict().items(), eng.state_dict().items()):
        assert a.shape == b.shape, k
    sd = om.deterministic_state_dict(ref, 41)
    ref.load_state_dict(sd)
    eng.load_state_dict(sd)
    mb = _batch([700, 260], seed0=630)
    mb["grid_size"] = 0.02
    mb["mask"] = torch.rand(960, generator=torch.Generator().manual_seed(1)) > 0.5
    feats = []
    with mock_backend.cpu_ops():
        for net in (ref, eng):
            net.train()
            torch.manual_seed(9)
            f = net({k: v for k, v in mb.items()}).feat
            (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean().backward()
            feats.append(f.detach())
    assert feats[0].shape == feats[1].shape == (960, 36)
    # bf16 operands in the engine's attention, fp16 in the reference's (litept_v1.py:235): 3 mantissa bits, up to 10 attention blocks
    assert _rel(feats[1], feats[0]) < 1.5e-2
    _grad_check(eng, ref, 1e-1)
    # ... and that this IS the only difference: with the reference's fp16 roundings emulated in the two functional entry points
    # (operands rounded to fp16 before and after the rotation, fp16 attention output), the engine's module reproduces the file
    from oracle import ops as oops
    from pointcept_amd import functional as PF

    def rope_fp16(qkv, xyz, inv_freq, out_dtype=None):
        h = qkv.half()
        rot = PF.rope_xyz_torch.__globals__["torch"].cat((_rot_fp32(h[:, :2].float(), xyz, inv_freq).half(), h[:, 2:]), dim=1)
        return rot

    def _rot_fp32(t, xyz, inv_freq):
        n, _, H, D = t.shape
        Q = D // 6
        emb = xyz[:, :, None] * inv_freq[None, None, :]
        cos, sin = emb.cos()[:, None, None, :, None, :], emb.sin()[:, None, None, :, None, :]
        t = t.reshape(n, 2, H, 3, 2, Q)
        u, v = t[..., 0:1, :], t[..., 1:2, :]
        return torch.cat((u * cos - v * sin, v * cos + u * sin), dim=-2).reshape(n, 2, H, D)

    monkeypatch.setattr(PF, "rope_xyz_qkvpacked", rope_fp16)
    monkeypatch.setattr(PF, "attn_varlen_qkvpacked",
                        lambda qkv, cu, k, scale, *a: oops.attention_varlen(qkv.float(), cu.tolist(), scale).to(qkv.dtype))
    eng.zero_grad(set_to_none=True)
    with mock_backend.cpu_ops():
        torch.manual_seed(9)
        f = eng({k: v for k, v in mb.items()}).feat
        (f * torch.linspace(-1, 1, f.shape[1])).pow(2).mean().backward()
    assert _rel(f.detach(), feats[0]) < 1e-3
    # gradients: 3e-2 where no max-pooling arg-max flips under the remaining 4e-4 forward difference; the second layout has 13 such
    # flips among the 3168 cells of its 68-point stage (measured: 26 cells of the pooled gradient move to a neighbouring row)
    _grad_check(eng, ref, 3e-2 if variant == "default_layout" else 1e-1)


def test_flash_attn_mirror_accepts_fp16_like_liteptS_call_site():
    """litept_v1.py:235-260 hands flash_attn fp16 rows (head_dim 18): the mirror runs them on f16-operand instances of the window-attention
    kernels and returns fp16, with gradients in the dtype of the caller's tensor."""
    from oracle import ops as oops
    from pointcept_amd import flash_attn_api

    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(300, 3, 2, 18, generator=g).half().requires_grad_(True)
    cu = torch.tensor([0, 128, 256, 300], dtype=torch.int32)
    with mock_backend.cpu_ops():
        out = flash_attn_api.flash_attn_varlen_qkvpacked_func(qkv, cu, max_seqlen=128, softmax_scale=0.2)
        out.float().pow(2).sum().backward()
    assert out.dtype == torch.float16 and out.shape == (300, 2, 18) and qkv.grad.dtype == torch.float16
    want = oops.attention_varlen(qkv.detach().float(), cu.tolist(), 0.2)
    assert _rel(out.detach().float(), want) < 1e-3 and float(qkv.grad.float().abs().max()) > 0


def test_flash_attn_mirror_pads_small_heads_to_one_mfma_step():
    """head_dim 1..15 (flash-attn serves them; no reference model uses them): zero channels up to 16, the caller's head_dim in the default
    softmax scale, the result cut back -- values and gradients are those of the unpadded attention."""
    from oracle import ops as oops
    from pointcept_amd import flash_attn_api

    g = torch.Generator().manual_seed(12)
    cu = torch.tensor([0, 100, 130, 131], dtype=torch.int32)
    for d in (8, 12, 1):
        qkv = torch.randn(131, 3, 3, d, generator=g).bfloat16().requires_grad_(True)
        w = torch.randn(131, 3, d, generator=g)
        with mock_backend.cpu_ops():
            out = flash_attn_api.flash_attn_varlen_qkvpacked_func(qkv, cu, max_seqlen=100)
            (out.float() * w).sum().backward()
        assert out.shape == (131, 3, d) and out.dtype == torch.bfloat16 and qkv.grad.shape == qkv.shape
        q32 = qkv.detach().float().requires_grad_(True)
        want = oops.attention_varlen(q32, cu.tolist(), d ** -0.5)
        (want * w).sum().backward()
        assert _rel(out.detach().float(), want.detach()) < 1e-2 and _rel(qkv.grad.float(), q32.grad) < 2e-2, d


@pytest.mark.needs_reference
def test_register_models_in_the_reference_registry():
    """B1 in one call: compat.register_models puts the engine's six module-level ports into the reference's MODELS registry
    (pointcept/utils/registry.py) under the names its configs use; MODELS.build then constructs the engine classes from a
    reference-style config dict, and the originals come back when the test restores them."""
    import sys

    import pointcept_amd.compat as compat
    from oracle import ref_import

    ref_import.load()
    MODELS = sys.modules["pointcept.models.builder"].MODELS
    saved = dict(MODELS._module_dict)
    try:
        names = compat.register_models(MODELS)
        assert set(names) == {"PT-v3m1", "PT-v3m2", "PT-v3m3", "LitePT-v1", "SpUNet-v1m1", "SpUNetNoSkipBase"}
        for n in names:
            assert MODELS.get(n).__module__.startswith("pointcept_amd."), n
        net = MODELS.build(dict(type="PT-v3m3", in_channels=6, order=ORDERS, enc_depths=(1, 1), enc_channels=(36, 72), enc_num_head=(2, 4),
                                stride=(2,), enc_patch_size=(64, 64), dec_depths=(1,), dec_channels=(36,), dec_num_head=(2,),
                                dec_patch_size=(64,), rope_base=10))
        assert type(net).__module__ == "pointcept_amd.point_transformer_v3m3" and any(k.endswith("rope.inv_freq") for k in net.state_dict())
        sp = MODELS.build(dict(type="SpUNet-v1m1", in_channels=6, num_classes=20, base_channels=16, channels=(16, 32, 32, 48, 48, 32, 32, 16),
                               layers=(1,) * 8))
        assert type(sp).__module__ == "pointcept_amd.sparse_unet"
        ns = MODELS.build(dict(type="SpUNetNoSkipBase", in_channels=6, out_channels=20, base_channels=16, channels=(16, 32, 32, 48, 48, 32, 32, 16),
                               layers=(1,) * 8))
        assert type(ns).__name__ == "SpUNetNoSkipBase" and type(ns).__module__ == "pointcept_amd.sparse_unet"
    finally:
        MODELS._module_dict.clear()
        MODELS._module_dict.update(saved)


def test_duplicate_row_merge_is_one_segmented_sum():
    """functional._merge_duplicate_rows (one segmented sum over a CSR of the representatives) against a per-row loop, on representatives
    with 1 .. 4 copies; the CSR is built once per representative tensor and dropped with it."""
    import gc

    from pointcept_amd import functional as PF

    g = torch.Generator().manual_seed(6)
    n = 500
    rep = torch.arange(n)
    src = torch.randperm(n, generator=g)[:120]
    for j, r in enumerate(src.tolist()):              # row r becomes a copy of a LOWER row (the representative is the lowest row)
        if r > 0:
            rep[r] = int(torch.randint(0, r, (1,), generator=g))
    rep = torch.where(rep[rep] != rep, rep[rep], rep)             # chains collapse onto the true lowest row ...
    rep = torch.where(rep[rep] != rep, rep[rep], rep)
    rep = torch.where(rep[rep] != rep, rep[rep], rep)
    assert bool((rep[rep] == rep).all()) and int((rep != torch.arange(n)).sum()) > 50
    grad = torch.randn(n, 24, generator=g)
    with mock_backend.cpu_ops():
        want = grad.clone()
        for r in range(n):                                     # ascending rows into the representative
            if int(rep[r]) != r:
                want[int(rep[r])] += grad[r]
        got = PF._merge_duplicate_rows(grad, rep)
        again = PF._merge_duplicate_rows(grad * 2, rep)
    assert torch.allclose(got, want, rtol=0, atol=1e-6) and torch.allclose(again, want * 2, rtol=0, atol=2e-6)
    assert torch.equal(got[rep != torch.arange(n)], grad[rep != torch.arange(n)])       # copies keep their own rows
    assert id(rep) in PF._dup_csr
    k = id(rep)
    del rep
    gc.collect()
    assert k not in PF._dup_csr


def test_bench_fp16_recipe_child_process_plumbing(monkeypatch):
    """bench.py measures the reference's fp16 + GradScaler recipe in a CHILD process (crash isolation for the headline numbers): the
    child's command line parses to the intended switches, its JSON line is picked out of a noisy stdout, failures become an
    {"error": ...} object instead of an exception."""
    import json
    import subprocess
    import sys
    import types

    monkeypatch.setattr(sys, "argv", ["bench.py", "--batch", "4", "--points", "2048"])
    import bench

    a = bench.parse()

    def fake_run(cmd, **kw):
        assert kw.get("timeout") and cmd[0] == sys.executable and cmd[1].endswith("bench.py")
        assert not any(k in kw["env"] for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
        monkeypatch.setattr(sys, "argv", cmd[1:])
        b = bench.parse()
        assert b.amp == "fp16" and b.no_secondary and b.no_cpu_baseline and b.no_fp16_recipe and (b.batch, b.points, b.gpus) == (4, 2048, 1)
        line = json.dumps({"value": 150.0, "unit": "scenes/s", "ms_per_step": 53.3, "steps": 5, "warmup": 2,
                           "config": {"amp": "fp16 autocast + GradScaler", "final_loss": 4.4}})
        return types.SimpleNamespace(returncode=0, stdout=("NCCL version banner\n" + line + "\n").encode())

    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    r = bench.fp16_recipe_in_a_child(a)
    assert r["value"] == 150.0 and r["amp"].startswith("fp16") and "error" not in r
    monkeypatch.setattr(bench.subprocess, "run", lambda *x, **k: types.SimpleNamespace(returncode=-11, stdout=b""))
    assert "error" in bench.fp16_recipe_in_a_child(a)

    def boom(*x, **k):
        raise subprocess.TimeoutExpired("bench", 1)

    monkeypatch.setattr(bench.subprocess, "run", boom)
    assert "TimeoutExpired" in bench.fp16_recipe_in_a_child(a)["error"]


def test_cast_twin_registry_identity_version_and_lifetime():
    """functional.register_cast_twin / cast_twin: the bf16 copy a residual joint wrote is handed out only for THE tensor it was
    registered for, only while that tensor is unmodified, only in the registered dtype / shape -- and the entry dies with it."""
    import gc

    from pointcept_amd import functional as PF

    x = torch.randn(10, 4)
    tw = x.to(torch.bfloat16)
    PF.register_cast_twin(x, tw)
    assert PF.cast_twin(x, torch.bfloat16) is tw
    assert PF.cast_twin(x, torch.float16) is None                      # another dtype
    assert PF.cast_twin(x.clone(), torch.bfloat16) is None              # equal values, different tensor
    assert PF.cast_twin(x[:5], torch.bfloat16) is None                  # a view is a different tensor object
    x.add_(1.0)                                                          # in-place write moves the version counter
    assert PF.cast_twin(x, torch.bfloat16) is None
    y = torch.randn(3, 2)
    PF.register_cast_twin(y, y.to(torch.bfloat16))
    n = len(PF._act_twins)
    del y
    gc.collect()
    assert len(PF._act_twins) == n - 1


def test_pointops2_offsets_to_pair_index():
    """pointops2_api._index_from_offsets: CSR offsets of the pairs by query -> query index of every pair (the inverse of the
    `index_0_offsets` the Stratified Transformer files build from a sorted `index_0`), empty segments included."""
    from oracle import pointops2 as orc
    from pointcept_amd import pointops2_api as p2

    i0 = torch.tensor([0, 0, 0, 2, 2, 5, 5, 5, 5], dtype=torch.int64)
    off = orc.offsets_of(i0, 7)
    assert off.tolist() == [0, 3, 3, 5, 5, 5, 9, 9]
    assert torch.equal(p2._index_from_offsets(off, i0.numel()).long(), i0)



def test_batched_bn_counters_context(monkeypatch):
    """nn.batched_bn_counters: counters appended inside the context get +1 each when it closes (one multi-tensor launch), a counter that
    was appended twice gets +2, nested contexts keep their own lists, the switch PTC_BATCH_BN_COUNTERS=0 turns collecting off (modules then
    increment by themselves), and the list is per thread."""
    import threading

    from pointcept_amd import config
    from pointcept_amd import nn as PNN

    a, b, c = (torch.zeros((), dtype=torch.int64) for _ in range(3))
    assert getattr(PNN._bn_tls, "pending", None) is None
    with PNN.batched_bn_counters():
        PNN._bn_tls.pending += [a, b, a]
        with PNN.batched_bn_counters():
            PNN._bn_tls.pending.append(c)
        assert int(c) == 1 and int(a) == 0          # the inner context closed, the outer one is still collecting
        seen = []
        t = threading.Thread(target=lambda: seen.append(getattr(PNN._bn_tls, "pending", None)))
        t.start()
        t.join()
        assert seen == [None]                       # another thread's forward does not see this list
    assert (int(a), int(b), int(c)) == (2, 1, 1) and PNN._bn_tls.pending is None
    with pytest.raises(RuntimeError):               # a forward that raises flushes nothing and still restores the outer list
        with PNN.batched_bn_counters():
            PNN._bn_tls.pending.append(b)
            raise RuntimeError("forward failed")
    assert int(b) == 1 and PNN._bn_tls.pending is None
    monkeypatch.setattr(config, "BATCH_BN_COUNTERS", False)
    with PNN.batched_bn_counters():
        assert PNN._bn_tls.pending is None
