"""GPU tests of the PT-v3m3 / LitePT-v1 module ports and of ptc_rope3d_xyz against the goldens of the reference's own model
files (tests/golden/make_golden_m3.py, make_golden_litept.py).  First hardware pass: round 3 (profiles/r03_a_tests_summary.txt).
Their bodies also run on the CPU stand-ins in every CPU run (tests/test_gpu_tests_dry_run_cpu.py).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")
M3_CFG = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 2, 1), enc_channels=(36, 72, 72, 144, 144), enc_num_head=(2, 4, 4, 8, 8),
              dec_depths=(1, 1, 1, 1), dec_channels=(36, 72, 72, 144), dec_num_head=(2, 4, 4, 8), enc_patch_size=(128,) * 5,
              dec_patch_size=(128,) * 4, drop_path=0.0, shuffle_orders=False, layer_scale=0.5, rope_base=10)   # = make_golden_m3.py


def test_rope3d_xyz_kernel_matches_reference_golden_and_inverts(cuda):
    """ptc_rope3d_xyz against the outputs of the reference's Point3DRoPE class (tests/golden/ptv3m3_tiny.npz, rope_*): fp32 in / out
    to 2e-6 (sin / cos of the device library against torch's), bf16 in -> bf16 out equal to the rounded fp32 result up to one ulp,
    the value slab converted and otherwise untouched, in place == out of place, and sign = -1 undoes sign = +1 (orthogonality),
    also at 819200 rows."""
    from pointcept_amd import ops

    g = np.load(os.path.join(GOLD, "ptv3m3_tiny.npz"))
    for ci in range(int(g["n_rope_cases"])):
        q, k = torch.from_numpy(g[f"rope_q_{ci}"]), torch.from_numpy(g[f"rope_k_{ci}"])
        xyz, inv_freq = torch.from_numpy(g[f"rope_xyz_{ci}"]).to(cuda), torch.from_numpy(g[f"rope_inv_freq_{ci}"]).to(cuda)
        v = torch.randn(q.shape, generator=torch.Generator().manual_seed(ci))
        qkv = torch.stack((q, k, v), dim=1).contiguous().to(cuda)                       # [n, 3, H, D]
        want = torch.stack((torch.from_numpy(g[f"rope_q_out_{ci}"]), torch.from_numpy(g[f"rope_k_out_{ci}"]), v), dim=1).to(cuda)
        out = ops.rope3d_xyz(qkv, xyz, inv_freq, 2, 1.0)
        assert out.dtype == torch.float32 and (out - want).abs().max() <= 2e-6 * max(1.0, float(want.abs().max())), ci
        assert torch.equal(out[:, 2], qkv[:, 2])
        back = ops.rope3d_xyz(out, xyz, inv_freq, 2, -1.0)
        assert (back - qkv).abs().max() <= 4e-6 * float(qkv.abs().max())
        # 16-bit operand in, bf16 out: the fp32 result of the ROUNDED operand, rounded once
        qb = qkv.to(torch.bfloat16)
        out_b = ops.rope3d_xyz(qb, xyz, inv_freq, 2, 1.0, torch.bfloat16)
        ref_b = ops.rope3d_xyz(qb.float(), xyz, inv_freq, 2, 1.0).to(torch.bfloat16)
        assert out_b.dtype == torch.bfloat16 and torch.equal(out_b[:, 2], qb[:, 2])
        diff = (out_b.float() - ref_b.float()).abs()
        assert float(diff.max()) <= 2 ** -7 * float(ref_b.float().abs().max()) and float((diff > 0).float().mean()) < 0.01
        # the library call in place
        from pointcept_amd._lib import lib
        from pointcept_amd.ops import check, dtype_code, ptr, stream_ptr
        buf = qb.clone()
        n, S, H, D = buf.shape
        check(lib().ptc_rope3d_xyz(ptr(buf), dtype_code(buf), ptr(buf), dtype_code(buf), ptr(xyz), ptr(inv_freq), n, S, 2, H, D, 1.0,
                                   stream_ptr()), "ptc_rope3d_xyz")
        assert torch.equal(buf, out_b)
    # full size: 819200 padded rows, 3 heads of 18 (Utonia stage 0), round trip
    gen = torch.Generator(device=cuda).manual_seed(3)
    n, H, D = 819200, 3, 18
    qkv = torch.randn(n, 3, H, D, device=cuda, generator=gen).to(torch.bfloat16)
    xyz = torch.rand(n, 3, device=cuda, generator=gen) * 8.0
    inv_freq = (1.0 / (10.0 ** (torch.arange(0, D // 3, 2).float() / (D // 3)))).to(cuda)
    out = ops.rope3d_xyz(qkv, xyz, inv_freq, 2, 1.0, torch.float32)
    assert torch.isfinite(out).all()
    nq, no = qkv[:, :2].float().reshape(n, 2 * H, 3, 2, D // 6).pow(2).sum(3), out[:, :2].reshape(n, 2 * H, 3, 2, D // 6).pow(2).sum(3)
    assert (nq - no).abs().max() <= 1e-4 * float(nq.max())                              # every rotated pair keeps its length
    back = ops.rope3d_xyz(out, xyz, inv_freq, 2, -1.0, torch.float32)
    assert (back - qkv.float()).abs().max() <= 1e-5 * float(qkv.float().abs().max())


def test_ptv3m3_matches_reference_golden(cuda):
    """the engine's module-level PT-v3m3 (m2 + Point3DRoPE; head_dim 18, 36 / 72 / 144 channels as in the reference's
    Utonia configs) against tests/golden/ptv3m3_tiny.npz = the REFERENCE's own point_transformer_v3m3_utonia.py: state-dict keys,
    eval features, train-mode loss and every gradient norm (the rotation on ptc_rope3d_xyz)."""
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic
    from pointcept_amd.point_transformer_v3m3 import PointTransformerV3 as M3

    g = np.load(os.path.join(GOLD, "ptv3m3_tiny.npz"))
    torch.manual_seed(0)
    eng = M3(**M3_CFG)
    assert list(eng.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    sd = om.deterministic_state_dict(eng, 37)
    assert abs(float(sum(float(v.double().abs().sum()) for v in sd.values())) - float(g["weight_checksum"])) < 1e-6 * float(g["weight_checksum"])
    eng.load_state_dict(sd)
    eng = eng.to(cuda)
    batch = synthetic.collate([synthetic.indoor_scene(int(s), int(n)) for s, n in zip(g["scene_seeds"], g["n_points"])])
    assert batch["grid_coord"].sum() == g["input_checksum"][0]
    inp = synthetic.to_torch(batch, cuda)
    inp["grid_size"] = 0.02
    eng.eval()
    torch.manual_seed(5)
    with torch.no_grad():
        out = eng(dict(inp)).feat.float().cpu().numpy()
    assert np.isfinite(out).all()
    err = np.abs(out[::8] - g["feat_rows"]).max() / float(g["feat_absmax"])
    assert err < 2e-2, f"engine PT-v3m3 vs reference golden: rel err {err:.3e}"
    eng.train()
    torch.manual_seed(5)
    f = eng(dict(inp)).feat
    loss = (f * torch.linspace(-1, 1, f.shape[1], device=f.device)).pow(2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-2 * abs(float(g["loss"])), (float(loss.detach()), float(g["loss"]))
    ref = dict(zip([str(k) for k in g["grad_names"]], g["grad_norms"]))
    gmax = max(ref.values())
    bad = []
    for name, p in eng.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        gn, rn = float(p.grad.norm()), ref[name]
        if not abs(gn - rn) <= 0.1 * rn + 1e-4 * gmax:
            bad.append((name, round(gn, 5), round(float(rn), 5)))
    if bad:   # the whole list for the next reader (pytest truncates the assertion message)
        os.makedirs(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out"), exist_ok=True)
        with open(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out", "grad_norm_mismatch.txt"), "a") as fh:
            fh.write(f"{type(eng).__name__}: parameter, engine |grad|, reference |grad|\n" + "\n".join(f"  {n} {a} {b}" for n, a, b in bad) + "\n")
    assert not bad, bad


LITEPT_CFG = dict(in_channels=6, order=ORDERS, enc_depths=(1, 1, 1, 2, 1), enc_channels=(36, 72, 72, 144, 144), enc_num_head=(2, 4, 4, 8, 8),
                  enc_patch_size=(128,) * 5, dec_channels=(36, 72, 72, 144), dec_num_head=(2, 4, 4, 8), dec_patch_size=(128,) * 4,
                  drop_path=0.0, shuffle_orders=False)                                             # = make_golden_litept.py


def test_litept_matches_reference_golden(cuda):
    """the engine's module-level LitePT-v1 (pointcept_amd/litept.py: convolution stages, PointROPE attention stages with
    head_dim 18, grid pooling with and without re-serialization, un-pooling decoder) against tests/golden/litept_tiny.npz = the
    REFERENCE's own litept_v1.py running libs/pointrope's own pointrope_cpu: state-dict keys, eval features, train-mode loss and
    every gradient norm.  (The reference's attention operands are fp16, the engine's bf16: inside the 2e-2 bar.)"""
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic
    from pointcept_amd.litept import LitePT

    g = np.load(os.path.join(GOLD, "litept_tiny.npz"))
    torch.manual_seed(0)
    eng = LitePT(**LITEPT_CFG)
    assert list(eng.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    sd = om.deterministic_state_dict(eng, 43)
    assert abs(float(sum(float(v.double().abs().sum()) for v in sd.values())) - float(g["weight_checksum"])) < 1e-6 * float(g["weight_checksum"])
    eng.load_state_dict(sd)
    eng = eng.to(cuda)
    batch = synthetic.collate([synthetic.indoor_scene(int(s), int(n)) for s, n in zip(g["scene_seeds"], g["n_points"])])
    assert batch["grid_coord"].sum() == g["input_checksum"][0]
    inp = synthetic.to_torch(batch, cuda)
    inp["grid_size"] = 0.02
    eng.eval()
    torch.manual_seed(5)
    with torch.no_grad():
        out = eng(dict(inp)).feat.float().cpu().numpy()
    assert np.isfinite(out).all()
    err = np.abs(out[::8] - g["feat_rows"]).max() / float(g["feat_absmax"])
    assert err < 2e-2, f"engine LitePT vs reference golden: rel err {err:.3e}"
    eng.train()
    torch.manual_seed(5)
    f = eng(dict(inp)).feat
    loss = (f * torch.linspace(-1, 1, f.shape[1], device=f.device)).pow(2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-2 * abs(float(g["loss"])), (float(loss.detach()), float(g["loss"]))
    ref = dict(zip([str(k) for k in g["grad_names"]], g["grad_norms"]))
    gmax = max(ref.values())
    bad = []
    for name, p in eng.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        gn, rn = float(p.grad.norm()), ref[name]
        if not abs(gn - rn) <= 0.1 * rn + 1e-4 * gmax:
            bad.append((name, round(gn, 5), round(float(rn), 5)))
    if bad:   # the whole list for the next reader (pytest truncates the assertion message)
        os.makedirs(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out"), exist_ok=True)
        with open(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out", "grad_norm_mismatch.txt"), "a") as fh:
            fh.write(f"{type(eng).__name__}: parameter, engine |grad|, reference |grad|\n" + "\n".join(f"  {n} {a} {b}" for n, a, b in bad) + "\n")
    assert not bad, bad


# the widths the reference actually ships: configs/sonata (PT-v3m2), configs/utonia (PT-v3m3), litept_v1.py:601
SHIPPED_WIDTHS = dict(
    m2=dict(enc_channels=(48, 96, 192, 384, 512), enc_num_head=(3, 6, 12, 24, 32), dec_channels=(48, 96, 192, 384), dec_num_head=(3, 6, 12, 24)),
    m3=dict(enc_channels=(54, 108, 216, 432, 576), enc_num_head=(3, 6, 12, 24, 32), dec_channels=(54, 108, 216, 432), dec_num_head=(3, 6, 12, 24)),
    litept=dict(enc_channels=(36, 72, 144, 252, 504), enc_num_head=(2, 4, 8, 14, 28), dec_channels=(72, 72, 144, 252), dec_num_head=(4, 4, 8, 14)))


@pytest.mark.parametrize("family", ["m2", "m3", "litept"])
def test_f2_models_at_shipped_widths_run_on_the_engine_only(cuda, family, monkeypatch):
    """PT-v3m2 / PT-v3m3 / LitePT at the channel widths of the reference's configs, bf16 autocast, forward + backward: no library GEMM
    and no library LayerNorm anywhere -- `F.linear`, `F.layer_norm`, `torch.matmul` / `@` raise while the model runs, and (where the
    profiler is available) no kernel of the step is a Tensile GEMM (`Cijk_*`) or an ATen layer-norm kernel.  The same step in fp32 must
    agree with the bf16 one (loss within 3 %): the generic LayerNorm and the padded GEMMs against their own fp32 forms."""
    from pointcept_amd import synthetic

    depths = dict(enc_depths=(1, 1, 1, 1, 1), enc_patch_size=(128,) * 5, dec_patch_size=(128,) * 4, drop_path=0.0, shuffle_orders=False)
    if family == "m2":
        from pointcept_amd.point_transformer_v3m2 import PointTransformerV3 as Net
        cfg = dict(in_channels=6, order=ORDERS, dec_depths=(1, 1, 1, 1), **depths, **SHIPPED_WIDTHS["m2"])
    elif family == "m3":
        from pointcept_amd.point_transformer_v3m3 import PointTransformerV3 as Net
        cfg = dict(in_channels=6, order=ORDERS, dec_depths=(1, 1, 1, 1), layer_scale=0.5, rope_base=10, **depths, **SHIPPED_WIDTHS["m3"])
    else:
        from pointcept_amd.litept import LitePT as Net
        cfg = dict(in_channels=6, order=ORDERS, **depths, **SHIPPED_WIDTHS["litept"])
    torch.manual_seed(0)
    net = Net(**cfg).to(cuda).train()
    batch = synthetic.collate([synthetic.indoor_scene(71, 6000), synthetic.indoor_scene(72, 2500)])

    def step(dtype):
        net.zero_grad(set_to_none=True)
        inp = synthetic.to_torch(batch, cuda)
        inp["grid_size"] = 0.02
        torch.manual_seed(5)
        with torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None and torch.device(cuda).type == "cuda"):
            f = net(inp).feat
        loss = (f.float() * torch.linspace(-1, 1, f.shape[1], device=f.device)).pow(2).mean()
        loss.backward()
        assert all(p.grad is None or torch.isfinite(p.grad).all() for p in net.parameters())
        return float(loss.detach())

    l32 = step(None)
    if family == "m2" and torch.device(cuda).type == "cuda":
        # the residual joints of these widths run fused (generic add_norm kernels, round 5): same loss and gradients as the unfused Blocks
        from pointcept_amd import config
        g_fused = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        monkeypatch.setattr(config, "FUSE_BLOCK", False)
        l_unfused = step(None)
        monkeypatch.setattr(config, "FUSE_BLOCK", True)
        assert abs(l_unfused - l32) < 1e-4 * abs(l32), (l_unfused, l32)
        num = sum(float((p.grad - g_fused[k]).double().pow(2).sum()) for k, p in net.named_parameters())
        den = sum(float(g_fused[k].double().pow(2).sum()) for k in g_fused)
        assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5     # bf16 attention operands on both sides: rounding flips, not a different function

    def refuse(name):
        def fn(*a, **k):
            raise AssertionError(f"{name} reached from the engine's model code")
        return fn

    on_gpu = torch.device(cuda).type == "cuda"
    kernels = None
    with monkeypatch.context() as mp:
        if on_gpu:          # (the CPU stand-ins ARE these torch calls)
            mp.setattr(torch.nn.functional, "linear", refuse("F.linear"))
            mp.setattr(torch.nn.functional, "layer_norm", refuse("F.layer_norm"))
            mp.setattr(torch.nn.functional, "batch_norm", refuse("F.batch_norm"))     # round 6: bn.hip has 8- / 4-byte lanes (C = 36, 54, 108, 252)
            mp.setattr(torch, "matmul", refuse("torch.matmul"))
            mp.setattr(torch.Tensor, "__matmul__", refuse("Tensor @"))
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA] if on_gpu else [ProfilerActivity.CPU]) as prof:
                l16 = step(torch.bfloat16)
                if on_gpu:
                    torch.cuda.synchronize()
            kernels = [e.key for e in prof.key_averages()] if on_gpu else None
        except (ImportError, RuntimeError):
            l16 = step(torch.bfloat16)
    assert abs(l16 - l32) < 3e-2 * abs(l32), (l16, l32)
    if kernels:
        bad = [k for k in kernels if k.startswith("Cijk_") or "layer_norm" in k.lower() and "ptc" not in k and "layer_norm_fwd" not in k and "layer_norm_bwd" not in k]
        bad += [k for k in kernels if "batch_norm" in k.lower() and "at::" in k]          # ATen's BatchNorm kernels (at::native::batch_norm_*)
        assert not bad, bad
        if family == "litept":      # (the m2 / m3 configs normalise with LayerNorm throughout)
            assert any("bn_apply_kernel" in k for k in kernels), "no engine BatchNorm kernel in the step"
        # the engine's own normalisation kernels ran: the 8-channels-per-lane instances with a run-time width where C % 8 == 0 (m2: 48 ..
        # 512), the pair-per-lane generic form elsewhere (m3: 54, 108; LitePT: 36, 252)
        assert any(("layer_norm_fwd" in k or "add_norm_fwd" in k) and "at::" not in k for k in kernels), "no engine LayerNorm kernel in the step"
        if family != "m2":
            assert any("generic_kernel" in k for k in kernels), "the generic (pair-per-lane) LayerNorm did not run"
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", f"f2_{family}_kernels.txt"), "w") as fh:
            fh.write("\n".join(sorted(kernels)) + "\n")
