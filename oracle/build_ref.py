"""TEST INFRASTRUCTURE (oracle).  Builds the pieces of the REFERENCE that compile from their own few sources into
oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).  Sources are compiled where they lie under
/root/reference -- nothing is copied into the repo.

  pointrope_ref : /root/reference/libs/pointrope/pointrope.cpp (pointrope_cpu, :13-49; the PYBIND11 module of :69-71) +
                  oracle/pointrope_stub.cpp (a definition for the forward-declared pointrope_cuda of :11 that throws: the
                  .cu file needs nvcc).  Built as a torch C++ extension (g++ via torch.utils.cpp_extension; no GPU code).

    python -m oracle.build_ref
"""
from __future__ import annotations

import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PTC_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")


def have_reference() -> bool:
    return os.path.exists(os.path.join(REF, "libs", "pointrope", "pointrope.cpp"))


def build_pointrope(verbose: bool = False):
    """-> the imported extension module (pointrope_ref.pointrope(tokens, positions, base, fwd), CPU tensors)."""
    import torch.utils.cpp_extension as ext

    os.makedirs(OUT, exist_ok=True)
    return ext.load(name="pointrope_ref", sources=[os.path.join(REF, "libs", "pointrope", "pointrope.cpp"), os.path.join(HERE, "pointrope_stub.cpp")],
                    build_directory=OUT, extra_cflags=["-O2"], verbose=verbose, with_cuda=False)


def load_pointrope():
    """the prebuilt module from oracle/_ref (GPU box: no /root/reference, no rebuild), building it first where the reference exists"""
    import glob
    import importlib.util

    if have_reference():
        return build_pointrope()
    so = glob.glob(os.path.join(OUT, "pointrope_ref*.so"))
    if not so:
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)

    spec = importlib.util.spec_from_file_location("pointrope_ref", so[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    if have_reference():
        m = build_pointrope(verbose=True)
        print("built", m.__file__)
    else:
        print("no reference tree at", REF)
