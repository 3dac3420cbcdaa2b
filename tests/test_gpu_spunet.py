"""-m gpu: the engine's SpUNet-v1m1 (BASELINE configs[1]; drop-in levels B1/B2 and B3) against
  (a) tests/golden/spunet_tiny.npz -- logits / loss / gradient norms produced by the REFERENCE file
      spconv_unet_v1m1_base.py run on the CPU stand-ins (tests/golden/make_golden.py),
  (b) the standalone CPU oracle (oracle/spunet_model.py) run live: every parameter gradient, running stats.

Tolerances: fp32 end to end (the conv kernels use the exact-f32 MFMA), so logits agree to 1e-3 of their
range (accumulation order only); gradients 2 % in Frobenius norm per parameter.
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(GOLD[: -len("/golden")])
TINY = dict(base_channels=16, channels=(16, 32, 48, 64, 64, 48, 32, 32), layers=(1, 2, 1, 1, 1, 1, 2, 1))


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def _pair(cfg, seed=1, **kw):
    from oracle import ptv3_model as om
    from oracle import spunet_model as osp
    from pointcept_amd.sparse_unet import SpUNetBase

    orc = osp.SpUNetBase(6, 20, **cfg, **kw)
    eng = SpUNetBase(6, 20, **cfg, **kw)
    assert list(orc.state_dict().keys()) == list(eng.state_dict().keys())
    for (k, a), (_, b) in zip(orc.state_dict().items(), eng.state_dict().items()):
        assert a.shape == b.shape, k
    sd = om.deterministic_state_dict(orc, seed)
    orc.load_state_dict(sd)
    eng.load_state_dict(sd)
    return orc, eng


def _golden_batch(g):
    from pointcept_amd import synthetic

    batch = synthetic.collate([synthetic.indoor_scene(int(s), int(n)) for s, n in zip(g["scene_seeds"], g["n_points"])])
    assert batch["grid_coord"].sum() == g["input_checksum"][0]
    return batch


def _fp64_twin(orc, batch):
    """the same oracle in float64 after one forward + backward on `batch`.  Why: the tiny configurations normalise over a
    dozen rows at their deepest level (train-mode BatchNorm) and are ill-conditioned enough that the fp32 CPU oracle ITSELF
    moves by 1-2 % on the worst BatchNorm parameters when only its thread count (= summation order) changes -- measured:
    8 threads 1.2e-2, 1 thread 1.8e-2, 4 threads 1.6e-5 away from the fp64 result, which is stable to 1e-13 under row
    permutation.  Gradients are therefore judged against the fp64 oracle, with a bar of twice the fp32 oracle's own spread."""
    import copy

    from oracle import spunet_model as osp

    o64 = copy.deepcopy(orc).double().train()
    o64.zero_grad(set_to_none=True)
    b = {k: torch.from_numpy(v) for k, v in batch.items()}
    b = {k: (v.double() if v.is_floating_point() else v) for k, v in b.items()}
    osp.Segmentor(o64)(b)["loss"].backward()
    return o64


def _compare_grads(eng, orc, tag, bar=0.02):
    go = dict(orc.named_parameters())
    rows, bad = [], []
    for name, p in eng.named_parameters():
        assert p.grad is not None, f"no gradient for {name}"
        assert torch.isfinite(p.grad).all(), name
        r = go[name].grad
        rows.append((name, float((p.grad.cpu().to(r.dtype) - r).norm()), float(r.norm()), float(r.abs().max())))
    gmax = max(r[3] for r in rows)
    for name, dn, rn, rmax in rows:
        rel = dn / max(rn, 1e-30)
        if rmax < 1e-5 * gmax:   # true gradient is zero (rounding noise on both sides): absolute comparison
            if dn > 1e-4 * gmax * max(1.0, float(go[name].numel()) ** 0.5):
                bad.append((name, rel, rn))
        elif rel > bar:
            bad.append((name, rel, rn))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/grad_report_{tag}.txt", "w") as f:
        f.write("\n".join(f"{dn / max(rn, 1e-30):10.3e} {rn:10.3e} {name}" for name, dn, rn, _ in rows) + "\n")
    assert not bad, f"gradient mismatch: {bad[:8]}"


def test_spunet_tiny_matches_reference_golden_and_oracle(cuda):
    from oracle import spunet_model as osp
    from pointcept_amd import functional as PF
    from pointcept_amd import synthetic

    g = np.load(os.path.join(GOLD, "spunet_tiny.npz"))
    orc, eng = _pair(TINY)
    assert [k for k, _ in eng.named_parameters()] == list(g["param_names"])
    eng = eng.to(cuda)
    batch = _golden_batch(g)
    dev = synthetic.to_torch(batch, cuda)
    amax = float(g["logits_absmax"])
    # eval mode (running statistics)
    eng.eval()
    with torch.no_grad():
        out = eng(dict(dev)).float().cpu().numpy()
    assert out.shape == (int(g["n_points"].sum()), 20) and np.isfinite(out).all()
    assert np.abs(out[::4] - g["logits_eval"]).max() <= 1e-3 * amax
    # train mode (batch statistics): logits, loss, gradients -- golden first, then the live oracle for every tensor
    eng.train()
    logits = eng(dict(dev))
    loss = PF.cross_entropy(logits, dev["segment"], -1)
    loss.backward()
    assert np.abs(logits.detach().float().cpu().numpy()[::4] - g["logits_train"]).max() <= 1e-3 * amax
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    norms = np.asarray([float(p.grad.double().norm()) for _, p in eng.named_parameters()])
    big = g["grad_norms"] > 1e-4 * g["grad_norms"].max()
    assert np.allclose(norms[big], g["grad_norms"][big], rtol=2e-2), np.abs(norms[big] / g["grad_norms"][big] - 1).max()
    assert _rel(eng.final.weight.grad, torch.from_numpy(g["grad_final"])) < 2e-3
    assert _rel(eng.conv_input[0].weight.grad, torch.from_numpy(g["grad_conv_input"])) < 2e-2
    orc.train()
    out_o = osp.Segmentor(orc)({k: torch.from_numpy(v) for k, v in batch.items()})
    out_o["loss"].backward()
    _compare_grads(eng, _fp64_twin(orc, batch), "spunet_tiny", bar=0.04)
    for (k, a), (_, b) in zip(eng.state_dict().items(), orc.state_dict().items()):
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert _rel(a, b) < 1e-3, k
        if k.endswith("num_batches_tracked"):
            assert int(a) == int(b) == 1


def test_spunet_base_channels_single_scene_and_duplicates(cuda):
    """the BASELINE channel plan (32..256, 96-wide decoder, 224-/192-channel concatenations) on one small scene
    whose voxel list carries duplicate coordinates (Mix3D merges scenes without re-voxelising):
    lowest row wins in every rulebook, on both sides."""
    from oracle import spunet_model as osp
    from pointcept_amd import functional as PF
    from pointcept_amd import synthetic

    cfg = dict(channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(1, 1, 1, 1, 1, 1, 1, 1))
    orc, eng = _pair(cfg, seed=4)
    eng = eng.to(cuda).train()
    orc.train()
    a, b = synthetic.indoor_scene(41, 1500), synthetic.indoor_scene(42, 900)
    mixed = {k: np.concatenate([a[k], b[k]]) for k in a}          # one batch item, overlapping coordinates
    batch = synthetic.collate([mixed, synthetic.indoor_scene(43, 600)])
    n_dup = len(mixed["grid_coord"]) - len(np.unique(mixed["grid_coord"], axis=0))
    assert n_dup > 0
    dev = synthetic.to_torch(batch, cuda)
    logits = eng(dict(dev))
    loss = PF.cross_entropy(logits, dev["segment"], -1)
    loss.backward()
    out_o = osp.Segmentor(orc)({k: torch.from_numpy(v) for k, v in batch.items()})
    out_o["loss"].backward()
    assert _rel(logits, out_o["seg_logits"]) < 2e-3
    assert abs(loss.item() - float(out_o["loss"])) <= 1e-4 * abs(float(out_o["loss"]))
    _compare_grads(eng, orc, "spunet_base")


def test_spunet_enc_mode(cuda):
    from pointcept_amd import synthetic

    orc, eng = _pair(TINY, seed=2, enc_mode=True)
    eng = eng.to(cuda).eval()
    orc.eval()
    batch = synthetic.collate([synthetic.indoor_scene(51, 2000), synthetic.indoor_scene(52, 500), synthetic.indoor_scene(53, 64)])
    with torch.no_grad():
        out = eng(synthetic.to_torch(batch, cuda))
        ref = orc({k: torch.from_numpy(v) for k, v in batch.items()})
    assert out.shape == ref.shape == (3, 20)
    assert _rel(out, ref) < 2e-3


def test_spunet_no_skip_variant(cuda):
    """SpUNetNoSkipBase (spconv_unet_v1m1_base.py:283-463: the decoder does not concatenate the encoder's features): same keys and
    shapes as the oracle restatement of that class, logits and gradients of a train-mode step within the fp32 bars of the v1m1 test"""
    from oracle import ptv3_model as om
    from oracle import spunet_model as osp
    from pointcept_amd import synthetic
    from pointcept_amd.sparse_unet import SpUNetNoSkipBase

    orc, eng = osp.SpUNetNoSkipBase(6, 20, **TINY), SpUNetNoSkipBase(6, 20, **TINY)
    assert list(orc.state_dict().keys()) == list(eng.state_dict().keys())
    assert all(a.shape == b.shape for a, b in zip(orc.state_dict().values(), eng.state_dict().values()))
    assert eng.dec[0].block0.conv1.in_channels == TINY["channels"][-1]          # no concatenated encoder channels
    sd = om.deterministic_state_dict(orc, 4)
    orc.load_state_dict(sd)
    eng.load_state_dict(sd)
    eng = eng.to(cuda).train()
    orc.train()
    batch = synthetic.collate([synthetic.indoor_scene(61, 3000), synthetic.indoor_scene(62, 700)])
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    ref = orc(tb)
    torch.nn.functional.cross_entropy(ref, tb["segment"], ignore_index=-1).backward()
    gb = synthetic.to_torch(batch, cuda)
    out = eng(gb)
    torch.nn.functional.cross_entropy(out, gb["segment"], ignore_index=-1).backward()
    assert _rel(out, ref) < 2e-3
    worst = max(_rel(pe.grad, po.grad) for (_, pe), (_, po) in zip(eng.named_parameters(), orc.named_parameters()) if po.grad is not None)
    assert worst < 5e-2, worst


def test_spunet_autocast_and_determinism(cuda):
    """bf16 / fp16 autocast (the reference trains under fp16 AMP, train.py:202-208) stay close to fp32, and two runs
    are bit-identical (no atomics)."""
    from pointcept_amd import functional as PF
    from pointcept_amd import synthetic

    _, eng = _pair(TINY, seed=5)
    eng = eng.to(cuda).train()
    dev = synthetic.to_torch(synthetic.collate([synthetic.indoor_scene(61, 3000)]), cuda)
    with torch.no_grad():
        ref = eng(dict(dev)).float()
    for dt, tol in ((torch.bfloat16, 0.1), (torch.float16, 0.02)):
        runs = []
        for _ in range(2):
            eng.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=dt):
                logits = eng(dict(dev))
                loss = PF.cross_entropy(logits, dev["segment"], -1)
            loss.backward()
            runs.append((logits.detach().clone(), [p.grad.clone() for p in eng.parameters()]))
        assert torch.isfinite(runs[0][0]).all()
        assert _rel(runs[0][0], ref) < tol, (dt, _rel(runs[0][0], ref))
        assert torch.equal(runs[0][0], runs[1][0])
        for x, y in zip(runs[0][1], runs[1][1]):
            assert torch.equal(x, y)


def test_reference_style_model_file_runs_on_the_engine_through_compat(cuda):
    """B3: a model file written against the `spconv.pytorch` names (here: the oracle's SpUNet source, re-executed with
    its spconv import bound to what pointcept_amd.compat.install() puts into sys.modules) runs on libptcore.so and
    reproduces the CPU oracle -- SparseConvTensor / SparseSequential dispatch over plain nn modules, SubMConv3d k=1/3/5,
    SparseConv3d, SparseInverseConv3d, indice_key reuse, autograd."""
    import pointcept_amd.compat as compat
    from oracle import ptv3_model as om
    from oracle import spunet_model as osp
    from pointcept_amd import synthetic

    saved = {k: sys.modules.get(k) for k in ("spconv", "spconv.pytorch", "spconv.pytorch.modules", "flash_attn", "torch_scatter")}
    try:
        compat.install(force=True)
        import spconv.pytorch as spconv  # noqa: F401  (resolves to pointcept_amd.spconv_api)
        from spconv.pytorch.modules import is_spconv_module

        src = open(os.path.join(ROOT, "oracle", "spunet_model.py")).read()
        assert "from . import shims as sp" in src
        mod = types.ModuleType("spunet_on_engine")
        exec(compile(src.replace("from . import shims as sp", "import spconv.pytorch as sp"), "spunet_on_engine", "exec"),
             mod.__dict__)
        net = mod.SpUNetBase(6, 20, **TINY)
        assert is_spconv_module(net.enc[0]) and is_spconv_module(net.enc[0][0]) and not is_spconv_module(net.enc[0][0].bn1)
        orc = osp.SpUNetBase(6, 20, **TINY)
        sd = om.deterministic_state_dict(orc, 6)
        orc.load_state_dict(sd)
        net.load_state_dict(sd)
        net = net.to(cuda).train()
        orc.train()
        batch = synthetic.collate([synthetic.indoor_scene(71, 2500), synthetic.indoor_scene(72, 300)])
        out = mod.Segmentor(net)(synthetic.to_torch(batch, cuda))
        out["loss"].backward()
        ref = osp.Segmentor(orc)({k: torch.from_numpy(v) for k, v in batch.items()})
        ref["loss"].backward()
        assert _rel(out["seg_logits"], ref["seg_logits"]) < 2e-3
        _compare_grads(net, orc, "spunet_compat")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_b3_flash_attn_and_segment_csr_entry_points(cuda):
    """the two third-party functions PT-v3m1 calls (ptv3m1:208-214, :416-421) under their own names and argument
    conventions, forward and backward, against the oracle; unsupported arguments raise instead of silently differing."""
    from oracle import ops as oops
    from pointcept_amd._lib import PtcoreError
    from pointcept_amd.flash_attn_api import flash_attn_varlen_qkvpacked_func
    from pointcept_amd.torch_scatter_api import segment_csr

    g = torch.Generator().manual_seed(3)
    lens = [128, 128, 77]
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    qkv = torch.randn(sum(lens), 3, 4, 16, generator=g).to(torch.bfloat16)
    dout = torch.randn(sum(lens), 4, 16, generator=g).to(torch.bfloat16)
    q_o = qkv.float().requires_grad_(True)
    ref = oops.attention_varlen(q_o, cu.tolist(), 0.25)
    ref.backward(dout.float())
    q_e = qkv.to(cuda).requires_grad_(True)
    with torch.no_grad():
        out = flash_attn_varlen_qkvpacked_func(q_e, cu.to(cuda), max_seqlen=128, dropout_p=0.0, softmax_scale=0.25)
    assert out.dtype == torch.bfloat16 and out.shape == (sum(lens), 4, 16)
    assert torch.allclose(out.float().cpu(), ref.detach(), rtol=2 ** -6, atol=2 ** -9 * float(qkv.float().abs().max()))
    q_e2 = qkv.to(cuda).requires_grad_(True)
    flash_attn_varlen_qkvpacked_func(q_e2, cu.to(cuda), 128, 0.0, softmax_scale=0.25).backward(dout.to(cuda))
    assert _rel(q_e2.grad, q_o.grad) < 3e-2
    # dropout_p > 0 (round 4): flash-attn's attention dropout on the engine's own counter-based mask (seed from torch's CPU generator)
    torch.manual_seed(11)
    d1 = flash_attn_varlen_qkvpacked_func(q_e2.detach(), cu.to(cuda), 128, dropout_p=0.1, softmax_scale=0.25)
    torch.manual_seed(11)
    d2 = flash_attn_varlen_qkvpacked_func(q_e2.detach(), cu.to(cuda), 128, dropout_p=0.1, softmax_scale=0.25)
    assert torch.equal(d1, d2) and not torch.equal(d1, out) and _rel(d1.float().cpu(), ref.detach()) < 0.8
    with pytest.raises(PtcoreError):
        flash_attn_varlen_qkvpacked_func(q_e2, cu.to(cuda), 128, causal=True)

    counts = torch.randint(1, 9, (200,), generator=g)
    indptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    src = torch.randn(int(indptr[-1]), 48, generator=g)
    for reduce in ("max", "mean", "sum"):
        s_o = src.clone().requires_grad_(True)
        r = oops.segment_csr(s_o, indptr, reduce)
        r.backward(torch.ones_like(r))
        s_e = src.to(cuda).requires_grad_(True)
        o = segment_csr(s_e, indptr.to(cuda), reduce=reduce)
        o.backward(torch.ones_like(o))
        assert torch.allclose(o.cpu(), r.detach(), rtol=1e-5, atol=1e-5), reduce
        assert torch.allclose(s_e.grad.cpu(), s_o.grad, rtol=1e-5, atol=1e-6), reduce
    with pytest.raises(PtcoreError):
        segment_csr(src.to(cuda), indptr.to(cuda), out=torch.empty(1, device=cuda))
