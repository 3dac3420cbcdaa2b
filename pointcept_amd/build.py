"""Build libptcore.so (HIP, gfx950) and libptcore_hostprobe.so (g++, CPU probes of the pure
host/device helper functions) IN-TREE next to this file.

hipcc cross-compiles for gfx950 without a GPU; the resulting .so files travel to the GPU box with
the repo snapshot.  Each .hip file is compiled to an object (parallel, cached by mtime), then
linked into one shared library whose only HIP dependency is libamdhip64.so.7 -- the SONAME of the
runtime bundled with PyTorch-ROCm, so that `import torch` followed by ctypes.CDLL shares ONE HIP
runtime (device pointers and streams are interchangeable).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libptcore.so")
PROBE_LIB = os.path.join(HERE, "libptcore_hostprobe.so")


def lib_path(variant: str = "") -> str:
    return os.path.join(HERE, f"libptcore_{variant}.so") if variant else LIB

HIP_SOURCES = [
    "serialize.hip",
    "scan_sort.hip",
    "maps.hip",
    "rows.hip",
    "rulebook.hip",
    "blocks.hip",
    "spconv.hip",
    "conv7.hip",
    "gemm3.hip",
    "wgrad3.hip",
    "wgrad7.hip",
    "norm.hip",
    "attention.hip",
    "loss.hip",
    "lovasz.hip",
    "voxelize.hip",
    "pointops.hip",
    "pointops_edges.hip",
    "pointops2.hip",
    "bn.hip",
    "rope.hip",
    "evalhist.hip",
    "block_exec.hip",
    "mlp.hip",
]
CXX_SOURCES = ["core.cpp"]
PROBE_SOURCES = ["host_probe.cpp"]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
HIP_FLAGS = [
    "-O3",
    "-std=c++17",
    f"--offload-arch={ARCH}",
    "-mcode-object-version=5",
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
]
# attention.hip: no SLP packing of the scalar f32 multiplies into v_pk_mul_f32 (beside MFMAs a packed f32 op costs more than
# the two scalar ones it replaces, MI355X_MICROARCH.md per-instruction constants): backward 1460 -> 1406 us at the bench shape.
# NOT global: the implicit-GEMM kernels lose 15-18 % without SLP at 96 / 128 channels (profiles/r02_o_slp_ab.txt).
# pointops2.hip: hardware fp32 atomic adds for the scatter gradients (the default expands atomicAdd(float) into a CAS loop)
# conv7.hip: MFMA accumulators in architectural VGPRs (one wave per SIMD: the default puts them in AGPRs and every block pays 128
# register moves; see the head of the file)
PER_FILE_FLAGS = {"attention.hip": ["-fno-slp-vectorize"], "pointops2.hip": ["-munsafe-fp-atomics"],
                  "conv7.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _newer(src: str, dst: str, extra: list[str]) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + extra if os.path.exists(p))


def _headers() -> list[str]:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hs.append(os.path.join(HERE, "..", "include", "ptcore.h"))
    return hs


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} {cmd[-1]}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


# Compiler-flag variants for A/B measurements in ONE GPU session: `python -m pointcept_amd.build --variant NAME` builds
# libptcore_NAME.so next to the default library (objects under csrc/build_NAME); PTC_LIB_VARIANT=NAME makes _lib.lib() load it.
VARIANTS = {
    "slp": [],                         # every file with the compiler's default SLP packing (no PER_FILE_FLAGS)
    "noslp": ["-fno-slp-vectorize"],   # no file with it
    "vgprform": ["-mllvm", "-amdgpu-mfma-vgpr-form"],   # every file with MFMA accumulators in architectural VGPRs where they fit (conv7.hip has it by default)
}
# "d_<MACRO>_<VALUE>": -D<MACRO>=<VALUE> (timing ablations of one kernel, e.g. d_C7_ABLATE_4); only the sources that name the macro
# (directly or through a header of csrc/) are recompiled, the other objects come from the default build


def _variant_flags(variant: str) -> list:
    if variant.startswith("d_"):
        macro, val = variant[2:].rsplit("_", 1)
        return [f"-D{macro}={val}"]
    return VARIANTS[variant]


def _names_macro(src_path: str, macro: str, hdrs: list) -> bool:
    text = open(src_path).read()
    return macro in text or any(macro in open(h).read() and f'"{os.path.basename(h)}"' in text for h in hdrs)


def build(force: bool = False, verbose: bool = False, variant: str = "") -> str:
    """Compile every HIP source for gfx950 and link libptcore.so.  Returns the library path."""
    LIB = lib_path(variant)
    BUILD = os.path.join(HERE, "csrc", "build" + ("_" + variant if variant else ""))
    HIP_FLAGS = globals()["HIP_FLAGS"] + (_variant_flags(variant) if variant else [])
    if not os.path.exists(HIPCC):
        if os.path.exists(LIB):
            return LIB  # GPU box without a compiler: use the prebuilt in-tree library
        raise RuntimeError(f"hipcc not found at {HIPCC} and no prebuilt {LIB}")
    os.makedirs(BUILD, exist_ok=True)
    # one builder at a time: the ranks of a multi-GPU job all pass through here (lib() -> build()); if the snapshot's
    # mtimes ever asked for a rebuild, N concurrent hipcc runs would write the same objects.  The others wait, then find
    # everything up to date.
    import fcntl

    with open(os.path.join(BUILD, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, variant, LIB, BUILD, HIP_FLAGS)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool, variant: str, LIB: str, BUILD: str, HIP_FLAGS: list) -> str:
    hdrs = _headers()
    jobs = []
    objs = []
    for src in HIP_SOURCES + CXX_SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise RuntimeError(f"missing source {sp}")
        op = os.path.join(BUILD, src + ".o")
        if variant.startswith("d_") and not _names_macro(sp, variant[2:].rsplit("_", 1)[0], hdrs):
            objs.append(os.path.join(HERE, "csrc", "build", src + ".o"))      # untouched by the macro: the default build's object
            continue
        objs.append(op)
        if force or _newer(sp, op, hdrs):
            if src.endswith(".hip"):
                cmd = [HIPCC] + HIP_FLAGS + (PER_FILE_FLAGS.get(src, []) if not variant or variant.startswith("d_") else []) + ["-c", sp, "-o", op]
            else:
                import zlib

                abi = f"{zlib.crc32(open(os.path.join(HERE, '..', 'include', 'ptcore.h'), 'rb').read()) & 0xffffffff:08x}"
                cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", f'-DPTC_ABI_HASH="{abi}"', "-x", "c++", "-c", sp, "-o", op]
            jobs.append(cmd)
    if jobs:
        if verbose:
            print(f"[ptcore.build] compiling {len(jobs)} file(s) for {ARCH}")
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or not os.path.exists(LIB):
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
    return LIB


def build_host_probe(force: bool = False) -> str:
    """g++ build of the pure helper functions shared with the kernels (CPU unit checks)."""
    gxx = shutil.which("g++")
    if gxx is None:
        if os.path.exists(PROBE_LIB):
            return PROBE_LIB
        raise RuntimeError("g++ not found")
    srcs = [os.path.join(CSRC, s) for s in PROBE_SOURCES]
    if force or any(_newer(s, PROBE_LIB, _headers()) for s in srcs):
        _run([gxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", PROBE_LIB] + srcs)
    return PROBE_LIB


if __name__ == "__main__":
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, verbose=True, variant=var))
    print(build_host_probe(force="--force" in sys.argv))
