"""TEST INFRASTRUCTURE (oracle).  Import the reference's OWN hot-path files, unmodified, from
/root/reference on top of oracle/shims.py (recipe of SURVEY Appendix E).

Only usable in the authoring container (the GPU box has no /root/reference): it is called by
tests/golden/make_golden.py to generate fixtures and by the `needs_reference` CPU tests, which skip
when the directory is absent.  Nothing in -m gpu tests, smoke() or bench.py imports this module.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF = os.environ.get("POINTCEPT_REFERENCE", "/root/reference")

# /root/reference is read-only by contract: importing its files must not drop __pycache__ directories next to them.  The flag is
# process-wide and set at import of this module, i.e. before any reference file is imported (the cost is re-parsing ~20 files).
sys.dont_write_bytecode = True


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "pointcept", "models"))


_loaded = {}


def load():
    """Returns dict(ptv3=<module>, spunet=<module>, serialization=<module>, structure=<module>)."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError(f"reference not found under {REF}")
    from . import shims

    shims.install_third_party()

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    # namespace stubs that bypass the heavy __init__.py files but keep the REAL source dirs
    pkg("pointcept", REF + "/pointcept")
    pkg("pointcept.models", REF + "/pointcept/models")
    pkg("pointcept.engines", REF + "/pointcept/engines")
    hooks = types.ModuleType("pointcept.engines.hooks")
    hooks.HookBase = type("HookBase", (), {})
    sys.modules[hooks.__name__] = hooks
    ppt = pkg("pointcept.models.point_prompt_training", REF + "/pointcept/models/point_prompt_training")
    ppt.PDNorm = importlib.import_module(
        "pointcept.models.point_prompt_training.prompt_driven_normalization").PDNorm
    pkg("pointcept.models.point_transformer_v3", REF + "/pointcept/models/point_transformer_v3")
    pkg("pointcept.models.sparse_unet", REF + "/pointcept/models/sparse_unet")
    _loaded["ptv3"] = importlib.import_module(
        "pointcept.models.point_transformer_v3.point_transformer_v3m1_base")
    _loaded["spunet"] = importlib.import_module("pointcept.models.sparse_unet.spconv_unet_v1m1_base")
    _loaded["serialization"] = importlib.import_module("pointcept.models.utils.serialization")
    _loaded["structure"] = importlib.import_module("pointcept.models.utils.structure")
    _loaded["misc"] = importlib.import_module("pointcept.models.utils.misc")
    return _loaded


def load_transform():
    """pointcept/datasets/transform.py (numpy / scipy only, apart from an unused torchvision import that is stubbed)."""
    load()
    if "transform" not in _loaded:
        pkg = types.ModuleType("pointcept.datasets")
        pkg.__path__ = [REF + "/pointcept/datasets"]
        sys.modules["pointcept.datasets"] = pkg
        if "torchvision" not in sys.modules:
            tv = types.ModuleType("torchvision")
            tv.transforms = types.ModuleType("torchvision.transforms")
            sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tv.transforms
        _loaded["transform"] = importlib.import_module("pointcept.datasets.transform")
    return _loaded["transform"]


def load_dataset_utils():
    """pointcept/datasets/utils.py (collate_fn / point_collate_fn with Mix3D), on the torch_scatter stand-in."""
    load()
    if "dataset_utils" not in _loaded:
        if "pointcept.datasets" not in sys.modules:
            pkg = types.ModuleType("pointcept.datasets")
            pkg.__path__ = [REF + "/pointcept/datasets"]
            sys.modules["pointcept.datasets"] = pkg
        mu = sys.modules["pointcept.models.utils"] if "pointcept.models.utils" in sys.modules else importlib.import_module("pointcept.models.utils")
        if not hasattr(mu, "offset2batch"):
            mu.offset2batch = _loaded["misc"].offset2batch
        _loaded["dataset_utils"] = importlib.import_module("pointcept.datasets.utils")
    return _loaded["dataset_utils"]
