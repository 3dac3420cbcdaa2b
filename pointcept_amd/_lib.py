"""ctypes binding of libptcore.so (the C-ABI declared in include/ptcore.h).

`import torch` happens BEFORE the library is loaded so that libptcore.so's DT_NEEDED
libamdhip64.so.7 resolves to the HIP runtime bundled with PyTorch-ROCm (one runtime, shared device
pointers and streams).  There is no CPU fallback: `lib()` raises if the library cannot be loaded,
and every op wrapper in this package raises on non-CUDA tensors.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: provides the HIP runtime)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libptcore.so")
PROBE_PATH = os.path.join(HERE, "libptcore_hostprobe.so")

c_i64, c_int, c_f32, c_size, c_ptr = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/ptcore.h one to one
_SIGNATURES = {
    "ptc_version": (ctypes.c_char_p, []),
    "ptc_last_error": (ctypes.c_char_p, []),
    "ptc_serialize_encode": (c_int, [c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_ptr]),
    "ptc_sort_keys_workspace_bytes": (c_size, [c_i64, c_int]),
    "ptc_sort_keys": (c_int, [c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_exclusive_scan_workspace_bytes": (c_size, [c_i64]),
    "ptc_exclusive_scan_i32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_patch_pad_maps": (c_int, [c_ptr, c_int, c_int, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_attn_tables": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_pool_maps_workspace_bytes": (c_size, [c_i64]),
    "ptc_pool_maps_count": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_pool_maps_fill": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "ptc_pool_child_codes": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "ptc_pool_level_counts": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_int, c_ptr, c_ptr]),
    "ptc_gather_rows": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "ptc_gather_rows_add": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "ptc_segment_csr_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_segment_csr_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_ptr, c_ptr]),
    "ptc_hash_table_size": (c_i64, [c_i64]),
    "ptc_hash_table_bytes": (c_size, [c_i64]),
    "ptc_hash_build": (c_int, [c_ptr, c_i64, c_ptr, c_size, c_ptr]),
    "ptc_rulebook_subm": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_size, c_ptr, c_ptr]),
    "ptc_rulebook_down_workspace_bytes": (c_size, [c_i64]),
    "ptc_rulebook_down_count": (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_rulebook_down_fill": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_spconv_fwd": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "ptc_ptv3_block_abi": (c_int, []),
    "ptc_ptv3_block_mlp_fused": (c_int, [c_int, c_int]),
    "ptc_ptv3_block_workspace_bytes": (c_size, [c_i64, c_i64, c_int, c_int]),
    "ptc_ptv3_block_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_ptv3_block_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_rulebook_blocks_tab_bytes": (c_size, [c_i64]),
    "ptc_rulebook_blocks": (c_int, [c_ptr, c_int, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_spconv_fwd_blk": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_i64, c_int, c_int, c_int, c_int,
                                   c_ptr, c_ptr]),
    "ptc_spconv_wgrad_blk_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "ptc_spconv_wgrad_blk": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_i64, c_int, c_int, c_int, c_int,
                                     c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_linear_supported_ex": (c_int, [c_int, c_int, c_int]),
    "ptc_linear_joint_supported": (c_int, [c_int, c_int, c_int]),
    "ptc_linear_norm_joint_fwd": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_f32, c_ptr, c_int, c_ptr, c_ptr, c_f32, c_int,
                                          c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_linear_joint_fwd": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_int,
                                     c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_linear_fwd_ex": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_mlp_supported": (c_int, [c_int, c_int]),
    "ptc_mlp_fwd": (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_mlp_bwd_workspace_bytes": (c_size, [c_i64, c_int]),
    "ptc_mlp_bwd": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_spconv_wgrad_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "ptc_spconv_wgrad": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_size,
                                 c_ptr]),
    "ptc_seg_eval_hist": (c_int, [c_ptr, c_int, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_i64, c_ptr, c_ptr]),
    "ptc_rope3d": (c_int, [c_ptr, c_int, c_ptr, c_i64, c_int, c_int, c_f32, c_f32, c_ptr]),
    "ptc_rope3d_xyz": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_f32, c_ptr]),
    "ptc_layer_norm_supported": (c_int, [c_int]),
    "ptc_layer_norm_fwd": (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_f32, c_ptr, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_layer_norm_bwd_workspace_bytes": (c_size, [c_i64, c_int]),
    "ptc_layer_norm_bwd": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                                   c_size, c_ptr]),
    "ptc_add_norm_fwd": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_f32, c_int, c_ptr, c_ptr, c_f32, c_int,
                                 c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_add_norm_bwd_workspace_bytes": (c_size, [c_i64, c_int]),
    "ptc_add_norm_bwd": (c_int, [c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr,
                                 c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_attn_varlen_fwd": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_f32, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_attn_varlen_bwd_workspace_bytes": (c_size, [c_i64, c_int]),
    "ptc_attn_varlen_dropout_fwd": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_f32, c_int, c_f32, ctypes.c_uint64, c_ptr, c_ptr, c_ptr]),
    "ptc_attn_varlen_dropout_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_f32, c_int, c_f32, ctypes.c_uint64,
                                            c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_attn_varlen_hd_supported": (c_int, [c_int, c_int]),
    "ptc_attn_varlen_hd_rope_supported": (c_int, [c_int, c_int]),
    "ptc_attn_varlen_hd_rope_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_attn_varlen_hd_rope_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_int, c_ptr,
                                            c_ptr, c_size, c_ptr]),
    "ptc_weight_layouts": (c_int, [c_ptr, c_ptr, c_int, c_i64, c_ptr]),
    "ptc_cast_many": (c_int, [c_ptr, c_ptr, c_int, c_i64, c_int, c_ptr]),
    "ptc_pair_dot_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "ptc_pair_dot_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_i64, c_i64, c_i64, c_int,
                                 c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_pair_aggregate_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "ptc_pair_aggregate_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr,
                                       c_ptr, c_ptr]),
    "ptc_attn_rpe_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_i64, c_int, c_int, c_f32, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_attn_rpe_bwd_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "ptc_attn_rpe_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_i64, c_int, c_int, c_f32, c_int, c_ptr,
                                 c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_attn_varlen_hd_fwd": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_attn_varlen_hd_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_int, c_ptr,
                                       c_ptr, c_size, c_ptr]),
    "ptc_attn_varlen_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_f32, c_int, c_ptr,
                                    c_ptr, c_size, c_ptr]),
    "ptc_batch_norm_supported": (c_int, [c_int, c_int]),
    "ptc_batch_norm_workspace_bytes": (c_size, [c_i64, c_int]),
    "ptc_batch_norm_act_fwd": (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_f32, c_f32, c_int, c_ptr, c_ptr, c_int, c_ptr, c_int,
                                       c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_batch_norm_act_bwd": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr,
                                       c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_batch_norm_add_act_fwd": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_f32, c_f32, c_int, c_ptr, c_ptr, c_int, c_ptr, c_int,
                                           c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_batch_norm_add_act_bwd": (c_int, [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr,
                                           c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_column_sum": (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "ptc_coord_max": (c_int, [c_ptr, c_int, c_i64, c_ptr, c_ptr]),
    "ptc_cross_entropy_partials": (c_i64, [c_i64]),
    "ptc_cross_entropy_fwd": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_i64, c_ptr, c_ptr, c_ptr]),
    "ptc_cross_entropy_bwd": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_i64, c_ptr, c_i64, c_ptr]),
    "ptc_knn_query": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_ball_query": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_i64, c_int, c_f32, c_f32, c_ptr, c_ptr, c_ptr]),
    "ptc_edge_rows_fwd": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_i64, c_ptr, c_i64, c_int, c_ptr]),
    "ptc_edge_reduce_fwd": (c_int, [c_int, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_i64, c_ptr, c_ptr]),
    "ptc_edge_csr_keys": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "ptc_edge_csr_ptr": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "ptc_edge_scatter_bwd": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_int, c_int, c_int, c_i64, c_ptr, c_ptr]),
    "ptc_pair_dot_weighted": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "ptc_pair_segment_sum": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "ptc_aggregation_edge_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_i64, c_ptr, c_ptr, c_ptr]),
    "ptc_farthest_point_sampling": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_ptr, c_ptr, c_ptr]),
    "ptc_voxel_keys": (c_int, [c_ptr, c_i64, ctypes.c_double, c_ptr, c_ptr, c_ptr, c_ptr]),
    "ptc_lovasz_softmax_workspace_bytes": (c_size, [c_i64, c_int]),
    "ptc_lovasz_softmax": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_i64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
}

PTC_F32, PTC_F16, PTC_BF16 = 0, 1, 2
REDUCE_CODES = {"sum": 0, "mean": 1, "max": 2, "min": 3}
ORDER_CODES = {"z": 0, "z-trans": 1, "hilbert": 2, "hilbert-trans": 3}

_lib = None


class PtcoreError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load libptcore.so (building it first when hipcc is available and the .so is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # build() is mtime-cached (a no-op when nothing changed) and returns the prebuilt library where there is no
    # hipcc (the GPU box): a stale .so can never meet newer ctypes signatures silently
    from . import build as _build

    variant = os.environ.get("PTC_LIB_VARIANT", "")      # compiler-flag A/B builds, see build.VARIANTS
    LIB_PATH = _build.build(variant=variant)
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # fail loudly: no fallback path exists
        raise PtcoreError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # ptc_version carries a hash of include/ptcore.h as it was when the library was compiled: compare it with the
    # header next to this binding (present in the repo; absent in a stripped install -> nothing to compare)
    ver = L.ptc_version().decode()
    hdr = os.path.join(HERE, "..", "include", "ptcore.h")
    if os.path.exists(hdr) and "abi " in ver:
        import zlib

        want = f"{zlib.crc32(open(hdr, 'rb').read()) & 0xffffffff:08x}"
        got = ver.split("abi ")[1].split(")")[0].split()[0]
        if got != want:
            raise PtcoreError(f"{LIB_PATH} was built against another include/ptcore.h (abi {got}, header {want}): rebuild "
                              "(python -m pointcept_amd.build --force)")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ptc_last_error().decode("utf-8", "replace")
        raise PtcoreError(f"{what} failed with code {rc}: {msg}")


def stream_ptr() -> int:
    # the raw HIP stream of torch's current stream on the current device (torch.cuda.current_stream().cuda_stream builds two
    # python objects per call: ~10 us x ~130 calls per step in the round-3 host profile, profiles/r03_i_host_profile.txt)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return PTC_F32
    if t.dtype == torch.float16:
        return PTC_F16
    if t.dtype == torch.bfloat16:
        return PTC_BF16
    raise PtcoreError(f"unsupported feature dtype {t.dtype}")


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PtcoreError(
                "pointcept_amd ops run only on the GPU through libptcore.so (no CPU fallback); "
                f"got a tensor on {t.device}")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


_blk_enums = None


def block_enums():
    """name -> index of the PTC_BLK_* argument tables of ptc_ptv3_block_fwd / _bwd, read from include/ptcore.h (one definition: the
    header the library was compiled against -- lib() has checked its hash)."""
    global _blk_enums
    if _blk_enums is None:
        import re

        txt = open(os.path.join(HERE, "..", "include", "ptcore.h")).read()
        out = {}
        for body in re.findall(r"enum\s*\{([^}]*)\}", txt):
            names = [x.strip() for x in body.replace("\n", " ").split(",") if x.strip()]
            if names and names[0].startswith("PTC_BLK_"):
                for i, nm in enumerate(names):
                    out[nm[len("PTC_BLK_"):]] = i
        out["ABI"] = int(re.search(r"#define\s+PTC_BLK_ABI\s+(\d+)", txt).group(1))
        _blk_enums = out
    return _blk_enums


_probe = None


def host_probe():
    """g++-built copy of the kernels' pure helper functions (CPU unit checks of product code)."""
    global _probe
    if _probe is None:
        if not os.path.exists(PROBE_PATH):
            from . import build as _build

            _build.build_host_probe()
        _probe = ctypes.CDLL(PROBE_PATH)
        _probe.probe_serialize_encode.argtypes = [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr]
        _probe.probe_serialize_encode.restype = None
        _probe.probe_patch_pad_maps.argtypes = [c_ptr, c_int, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]
        _probe.probe_patch_pad_maps.restype = None
        _probe.probe_padded_len.argtypes = [c_i64, c_i64]
        _probe.probe_padded_len.restype = c_i64
        _probe.probe_num_seq.argtypes = [c_i64, c_i64]
        _probe.probe_num_seq.restype = c_i64
        _probe.probe_vox_pack.argtypes = [c_int, c_int, c_int, c_int]
        _probe.probe_vox_pack.restype = ctypes.c_uint64
        _probe.probe_vox_hash.argtypes = [ctypes.c_uint64]
        _probe.probe_vox_hash.restype = ctypes.c_uint64
    return _probe
