"""-m "not gpu": ELEMENTWISE kernels of pointcept_amd/csrc compiled UNMODIFIED for the host (tests/host_emulation: a stand-in for
<hip/hip_runtime.h> that turns a 1-D launch into a loop over (blockIdx.x, threadIdx.x)) and executed on the CPU against the goldens.
This checks a kernel's index arithmetic, dtype dispatch and argument validation without a GPU -- what it cannot check is anything
the hardware adds (LDS, wave intrinsics, MFMA, memory ordering): kernels that use those are only tested with -m gpu.

Kernels covered: rope.hip -- ptc_rope3d (libs/pointrope; already validated on the GPU, here it validates the harness itself against
the reference's pointrope_cpu golden) and ptc_rope3d_xyz (PT-v3m3 Point3DRoPE on packed rows; written after round 2's GPU time was
spent, so this is its only execution so far)."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
F32, F16, BF16 = 0, 1, 2          # ptc_dtype tags of include/ptcore.h


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ under /opt/rocm")
    out = str(tmp_path_factory.mktemp("emu") / "libptc_host_emu.so")
    src = os.path.join(HERE, "host_emulation")
    r = subprocess.run([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", src, "-o", out, os.path.join(src, "emulate.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    L = ctypes.CDLL(out)
    vp, i64, ci, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    L.ptc_rope3d.argtypes = [vp, ci, vp, i64, ci, ci, cf, cf, vp]
    L.ptc_rope3d_xyz.argtypes = [vp, ci, vp, ci, vp, vp, i64, ci, ci, ci, ci, cf, vp]
    L.emu_last_error.restype = ctypes.c_char_p
    return L


def _tag(t):
    return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[t.dtype]


def test_header_constants_match(emu):
    import re

    hdr = open(os.path.join(os.path.dirname(HERE), "include", "ptcore.h")).read()
    for name, val in (("PTC_F32", F32), ("PTC_F16", F16), ("PTC_BF16", BF16)):
        m = re.search(name + r"\s*=\s*(\d+)", hdr)
        assert m and int(m.group(1)) == val, name


def test_rope3d_on_the_host_emulation_matches_the_reference_cpu_golden(emu):
    """harness check: the GPU-validated ptc_rope3d, run thread by thread on the CPU, reproduces tests/golden/pointrope.npz
    (= libs/pointrope's own pointrope_cpu)"""
    g = np.load(os.path.join(GOLD, "pointrope.npz"))
    for ci in range(int(g["n_cases"])):
        B, N, H, D = (int(v) for v in g[f"shape_{ci}"])
        base, fwd = (float(v) for v in g[f"params_{ci}"])
        tok = torch.from_numpy(g[f"tokens_{ci}"].copy()).contiguous()
        pos = torch.from_numpy(g[f"pos_{ci}"].copy()).contiguous()
        assert emu.ptc_rope3d(tok.data_ptr(), F32, pos.data_ptr(), B * N, H, D, base, fwd, None) == 0
        # 2e-4: the bar of the GPU test of the same kernel (integer positions up to 300: the angle pos * (F0 / base^(i/Q)) of the CUDA
        # kernel and F0 * pos / base^(i/Q) of pointrope_cpu differ by an ulp of a 300-radian angle); the oracle with the kernel's
        # order of operations is matched much closer
        from oracle import pointrope as orope
        assert np.abs(tok.numpy() - g[f"out_{ci}"]).max() <= 2e-4 * np.abs(g[f"out_{ci}"]).max(), ci
        assert np.abs(tok.numpy() - orope.pointrope(g[f"tokens_{ci}"], g[f"pos_{ci}"], base, fwd)).max() <= 2e-5 * np.abs(g[f"out_{ci}"]).max(), ci


def test_rope3d_xyz_on_the_host_emulation(emu):
    """ptc_rope3d_xyz, thread by thread on the CPU: the reference Point3DRoPE golden (fp32), every dtype pair against the fp32 result
    rounded once, the pass-through slab, in place == out of place, sign = -1 inverts, and the argument checks."""
    g = np.load(os.path.join(GOLD, "ptv3m3_tiny.npz"))
    for ci in range(int(g["n_rope_cases"])):
        q, k = torch.from_numpy(g[f"rope_q_{ci}"]), torch.from_numpy(g[f"rope_k_{ci}"])
        xyz, f = torch.from_numpy(g[f"rope_xyz_{ci}"]).contiguous(), torch.from_numpy(g[f"rope_inv_freq_{ci}"]).contiguous()
        v = torch.randn(q.shape, generator=torch.Generator().manual_seed(ci))
        qkv = torch.stack((q, k, v), dim=1).contiguous()
        n, S, H, D = qkv.shape
        want = torch.stack((torch.from_numpy(g[f"rope_q_out_{ci}"]), torch.from_numpy(g[f"rope_k_out_{ci}"]), v), dim=1)

        def run(src, out_dtype, sign=1.0, rot=2):
            dst = torch.full(src.shape, float("nan")).to(out_dtype)
            rc = emu.ptc_rope3d_xyz(src.data_ptr(), _tag(src), dst.data_ptr(), _tag(dst), xyz.data_ptr(), f.data_ptr(), n, S, rot, H, D, sign, None)
            assert rc == 0, emu.emu_last_error()
            return dst

        out = run(qkv, torch.float32)
        assert (out - want).abs().max() <= 3e-6 * max(1.0, float(want.abs().max())), ci
        assert torch.equal(out[:, 2], qkv[:, 2])
        back = run(out, torch.float32, sign=-1.0)
        assert (back - qkv).abs().max() <= 5e-6 * float(qkv.abs().max())
        only_q = run(qkv, torch.float32, rot=1)                         # rot_slabs = 1: k passes through
        assert torch.equal(only_q[:, 1], qkv[:, 1]) and torch.equal(only_q[:, 0], out[:, 0])
        for in_dt in (torch.float32, torch.bfloat16, torch.float16):
            src = qkv.to(in_dt)
            ref32 = run(src.float(), torch.float32)
            for out_dt in (torch.float32, torch.bfloat16, torch.float16):
                got = run(src, out_dt)
                assert torch.equal(got, ref32.to(out_dt)), (ci, in_dt, out_dt)             # fp32 arithmetic, one rounding
        buf = qkv.to(torch.bfloat16).clone()
        keep_v = buf[:, 2].clone()
        assert emu.ptc_rope3d_xyz(buf.data_ptr(), BF16, buf.data_ptr(), BF16, xyz.data_ptr(), f.data_ptr(), n, S, 2, H, D, 1.0, None) == 0
        assert torch.equal(buf, run(qkv.to(torch.bfloat16), torch.bfloat16)) and torch.equal(buf[:, 2], keep_v)
    # argument checks (same code path as the library)
    z = torch.zeros(4, 3, 2, 18)
    x3, fr = torch.zeros(4, 3), torch.zeros(3)
    assert emu.ptc_rope3d_xyz(z.data_ptr(), F32, z.data_ptr(), BF16, x3.data_ptr(), fr.data_ptr(), 4, 3, 2, 2, 18, 1.0, None) == -1     # in place, two dtypes
    assert emu.ptc_rope3d_xyz(z.data_ptr(), F32, z.data_ptr(), F32, x3.data_ptr(), fr.data_ptr(), 4, 3, 2, 2, 20, 1.0, None) == -2
    assert emu.ptc_rope3d_xyz(z.data_ptr(), 7, z.data_ptr(), F32, x3.data_ptr(), fr.data_ptr(), 4, 3, 2, 2, 18, 1.0, None) == -1        # bad dtype tag


def test_rope3d_xyz_host_emulation_agrees_with_the_engine_formulation(emu):
    """the kernel (emulated) against functional.rope_xyz_torch -- the formulation PT-v3m3 / LitePT run while the kernel is switched
    off -- on a random bf16 batch: bit-identical bf16 outputs, i.e. flipping config.ROPE_XYZ_KERNEL changes no number."""
    from pointcept_amd import functional as PF

    gen = torch.Generator().manual_seed(12)
    n, H, D = 777, 3, 18
    qkv = torch.randn(n, 3, H, D, generator=gen).to(torch.bfloat16)
    xyz = (torch.rand(n, 3, generator=gen) * 9.0 - 2.0).contiguous()
    f = (1.0 / (10.0 ** (torch.arange(0, D // 3, 2).float() / (D // 3)))).contiguous()
    dst = torch.empty_like(qkv)
    assert emu.ptc_rope3d_xyz(qkv.data_ptr(), BF16, dst.data_ptr(), BF16, xyz.data_ptr(), f.data_ptr(), n, 3, 2, H, D, 1.0, None) == 0
    ref = PF.rope_xyz_torch(qkv, xyz, f)
    diff = (dst.float() - ref.float()).abs()
    # sin / cos come from two math libraries (glibc here, the device library on the GPU, ATen in the torch path): a last-bit
    # difference in sin / cos may flip a bf16 rounding of the product
    assert float((diff > 0).float().mean()) < 2e-3 and float(diff.max()) <= 2 ** -7 * float(ref.float().abs().max())
