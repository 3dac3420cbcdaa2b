"""TEST INFRASTRUCTURE.  `pointcept_amd.ops` on the HOST EMULATION of the kernel files that need neither LDS nor wave intrinsics nor
MFMA (tests/host_emulation): inside `emulated_ops()` the ctypes library behind ops.* is the emulation build (same extern "C" entry
points, same source files, compiled by the host clang++), "GPU only" guards are lifted and the stream is NULL, so the bodies of the
corresponding `-m gpu` tests run on CPU tensors against the same oracles.  Entry points of files that are not emulated raise."""
import contextlib
import ctypes
import os
import subprocess
import tempfile

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
HERE = os.path.dirname(os.path.abspath(__file__))
_lib_cache = {}


def available() -> bool:
    return os.path.exists(CLANG)


FILES = None      # None = every .hip of pointcept_amd/csrc

# The emulator's own translation unit: the fiber switch and the dynamic-LDS arrays.
_RUNTIME_CPP = "#define EMU_IMPLEMENTATION 1\n#include <hip/hip_runtime.h>\n"


def build(verbose: bool = False):
    """Every kernel file compiled as its own translation unit (as in the real build) against tests/host_emulation/hip/hip_runtime.h,
    after the token substitution `extern __shared__` -> `extern`, `__shared__` -> `static`; linked into one shared library."""
    if "lib" in _lib_cache:
        return _lib_cache["lib"]
    pre = os.environ.get("PTC_EMU_LIB")       # a library a parent process built (tests/test_dp_gloo.py hands it to its ranks)
    if pre and os.path.exists(pre):
        _lib_cache["lib"] = ctypes.CDLL(pre)
        _lib_cache["dir"] = os.path.dirname(pre)
        return _lib_cache["lib"]
    import glob
    import re
    import shutil

    root = os.path.dirname(HERE)
    d = tempfile.mkdtemp(prefix="ptc_emu_")
    csrc = os.path.join(d, "pointcept_amd", "csrc")
    os.makedirs(csrc)
    os.makedirs(os.path.join(d, "include"))
    shutil.copy(os.path.join(root, "include", "ptcore.h"), os.path.join(d, "include", "ptcore.h"))
    for f in glob.glob(os.path.join(root, "pointcept_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "pointcept_amd", "csrc", "*.inc")) + \
            glob.glob(os.path.join(root, "pointcept_amd", "csrc", "*.hip")) + \
            glob.glob(os.path.join(root, "pointcept_amd", "csrc", "*.cpp")):
        txt = open(f).read()
        txt = re.sub(r"extern\s+__shared__", "extern", txt)
        txt = re.sub(r"\b__shared__\b", "static", txt)
        open(os.path.join(csrc, os.path.basename(f)), "w").write(txt)
    shim = os.path.join(HERE, "host_emulation")
    open(os.path.join(csrc, "emu_runtime.cpp"), "w").write(_RUNTIME_CPP)
    units = sorted(glob.glob(os.path.join(csrc, "*.hip"))) + [os.path.join(csrc, "core.cpp"), os.path.join(csrc, "emu_runtime.cpp")]
    units = [u for u in units if os.path.basename(u) != "host_probe.cpp"]
    if FILES is not None:
        units = [u for u in units if os.path.basename(u) in FILES or u.endswith(".cpp")]
    objs, procs = [], []
    for u in units:
        o = u + ".o"
        objs.append(o)
        # spconv.hip instantiates ~200 implicit-GEMM kernels whose always-inline bodies take the optimiser 4.5 minutes: -O0 (9 s)
        opt = "-O0" if os.path.basename(u) in ("spconv.hip",) else "-O1"
        # PTC_EMU_ASAN=1: AddressSanitizer on the kernels' GLOBAL and HEAP accesses (not their stacks: lanes run on switched stacks), to
        # be used with  LD_PRELOAD=<clang's libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0  (tools/emu_asan.sh)
        extra = ["-fsanitize=address", "-mllvm", "-asan-stack=0", "-g"] if os.environ.get("PTC_EMU_ASAN") == "1" else []
        # PTC_EMU_DEFINES="LV_XCD=1,...": the emulation of a build variant (python -m pointcept_amd.build --variant d_<MACRO>_<VALUE>)
        extra += ["-D" + m for m in os.environ.get("PTC_EMU_DEFINES", "").split(",") if m]
        procs.append((u, subprocess.Popen([CLANG, "-x", "c++", "-std=c++17", opt, "-fPIC", "-w", "-I", shim] + extra + ["-c", u, "-o", o],
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    errs = []
    for u, pr in procs:
        _, err = pr.communicate()
        if pr.returncode != 0:
            errs.append(f"== {os.path.basename(u)}\n" + err[-2500:])
    if errs:
        raise RuntimeError("host emulation build failed:\n" + "\n".join(errs))
    out = os.path.join(d, "libptc_host_emu.so")
    link = ["-fsanitize=address", "-shared-libasan"] if os.environ.get("PTC_EMU_ASAN") == "1" else []
    r = subprocess.run([CLANG, "-shared", "-o", out] + link + objs + ["-lm"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host emulation link failed:\n" + r.stderr[-3000:])
    _lib_cache["lib"] = ctypes.CDLL(out)
    _lib_cache["dir"] = d
    _lib_cache["path"] = out
    return _lib_cache["lib"]


def library_path() -> str:
    """path of the built emulation library (builds it on first use)"""
    build()
    return _lib_cache.get("path") or os.environ["PTC_EMU_LIB"]


class _EmuLib:
    """the emulation library with the ctypes signatures of pointcept_amd._lib; anything it does not export is refused"""

    def __init__(self):
        from pointcept_amd import _lib
        from pointcept_amd._lib import PtcoreError

        self._cdll, self._sig, self._err = build(), _lib._SIGNATURES, PtcoreError
        self._fns = {}

    def __getattr__(self, name):
        fns = self.__dict__["_fns"]
        if name not in fns:
            try:
                fn = getattr(self._cdll, name)
            except AttributeError:
                raise self._err(f"{name}: not part of the host emulation (its kernel file uses LDS / wave intrinsics / MFMA)") from None
            fn.restype, fn.argtypes = self._sig[name]
            fns[name] = fn
        return fns[name]


@contextlib.contextmanager
def emulated_ops():
    from pointcept_amd import _lib, ops

    emu = _EmuLib()
    saved = [(ops, "lib", ops.lib), (_lib, "lib", _lib.lib), (ops, "require_cuda", ops.require_cuda), (ops, "stream_ptr", ops.stream_ptr)]
    try:
        ops.lib = _lib.lib = lambda: emu
        ops.require_cuda = lambda *a, **k: None
        ops.stream_ptr = lambda: None
        yield emu
    finally:
        for mod, name, val in saved:
            setattr(mod, name, val)


@contextlib.contextmanager
def hybrid(real_ops):
    """The model-level CPU backend (tests/mock_backend.py: oracle stand-ins behind the ops' contracts, GPU guards lifted) with the
    ops named in `real_ops` running their REAL kernels on the host emulation instead."""
    import mock_backend
    from pointcept_amd import ops

    orig = {n: getattr(ops, n) for n in real_ops}
    with mock_backend.cpu_ops(), emulated_ops():
        saved = {n: getattr(ops, n) for n in real_ops}
        try:
            for n, f in orig.items():
                setattr(ops, n, f)
            yield
        finally:
            for n, f in saved.items():
                setattr(ops, n, f)

