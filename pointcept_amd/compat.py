"""Operator-level drop-in (SURVEY 8(b) level B3): make `import spconv.pytorch`, `import flash_attn`
`import torch_scatter`, `import pointops`, `import pointops2.pointops` and `import pointrope` resolve to the engine, so the reference's model files
(point_transformer_v3m1_base.py, spconv_unet_v1m1_base.py, structure.py, modules.py) run UNMODIFIED
on libptcore.so.  Call once before importing pointcept.models:

    import pointcept_amd.compat; pointcept_amd.compat.install()

Only names the hot-path files use are provided; real packages already imported are left alone
unless force=True.
"""
from __future__ import annotations

import sys
import types


def install(force: bool = False) -> None:
    from . import flash_attn_api, pointops2_api, pointops_api, pointrope_api, spconv_api, torch_scatter_api

    def put(name, module):
        if force or name not in sys.modules:
            sys.modules[name] = module

    sp = types.ModuleType("spconv")
    sp.pytorch = spconv_api
    put("spconv", sp)
    put("spconv.pytorch", spconv_api)
    put("spconv.pytorch.modules", spconv_api.modules)
    fa = types.ModuleType("flash_attn")
    fa.flash_attn_varlen_qkvpacked_func = flash_attn_api.flash_attn_varlen_qkvpacked_func
    put("flash_attn", fa)
    ts = types.ModuleType("torch_scatter")
    ts.segment_csr = torch_scatter_api.segment_csr
    put("torch_scatter", ts)
    put("pointops", pointops_api)
    # libs/pointops2: `import pointops2.pointops as pointops` (stratified_transformer_v1m1_origin.py:21, v1m2_refine.py:31)
    p2 = types.ModuleType("pointops2")
    p2.pointops = pointops2_api
    p2f = types.ModuleType("pointops2.functions")
    p2f.pointops = pointops2_api
    p2.functions = p2f
    put("pointops2", p2)
    put("pointops2.pointops", pointops2_api)
    put("pointops2.functions", p2f)
    put("pointops2.functions.pointops", pointops2_api)
    put("pointrope", pointrope_api)      # libs/pointrope: `import pointrope as _kernels` (litept_v1.py:26)


# ------------------------------------------------------------------------------------------------
# module level: the engine's backbones under the reference's registry names
# ------------------------------------------------------------------------------------------------
MODEL_CLASSES = {                                   # registry name (reference file:line) -> (engine module, class)
    "PT-v3m1": ("point_transformer_v3", "PointTransformerV3"),        # point_transformer_v3m1_base.py:518
    "PT-v3m2": ("point_transformer_v3m2", "PointTransformerV3"),      # point_transformer_v3m2_sonata.py:544
    "PT-v3m3": ("point_transformer_v3m3", "PointTransformerV3"),      # point_transformer_v3m3_utonia.py:686
    "LitePT-v1": ("litept", "LitePT"),                                # litept_v1.py:593
    "SpUNet-v1m1": ("sparse_unet", "SpUNetBase"),                     # spconv_unet_v1m1_base.py:88
    "SpUNetNoSkipBase": ("sparse_unet", "SpUNetNoSkipBase"),          # spconv_unet_v1m1_base.py:283 (registered under its class name)
}


def register_models(registry, names=None, force: bool = True) -> list:
    """Registers the engine's module-level ports in the reference's `MODELS` registry (pointcept/models/builder.py) under the
    names the reference's configs use, replacing the CUDA-library implementations (`force=True`), so that
    `MODELS.build(cfg.model.backbone)` constructs them.  Returns the names registered.

        from pointcept.models.builder import MODELS
        import pointcept_amd.compat; pointcept_amd.compat.register_models(MODELS)
    """
    import importlib

    done = []
    for name in (names or MODEL_CLASSES):
        mod, cls = MODEL_CLASSES[name]
        registry.register_module(name, force=force, module=getattr(importlib.import_module(f"{__package__}.{mod}"), cls))
        done.append(name)
    return done

