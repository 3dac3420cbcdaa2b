"""Engine feature switches (environment variables, read once at import).

Every switch selects between two forms of the same math that are BOTH on libptcore.so -- there is no CPU path and no
library (hipBLASLt / ATen / SDPA) backend to switch to (round 4: PTC_OWN_LINEAR / PTC_OWN_NORM are gone; the library
comparison lives in tools/linear_kernels.py).  They exist so one GPU session can A/B a fusion against
the unfused form of the same math (tests/test_gpu_model.py runs the oracle comparison under both
settings), and so a regression can be bisected without a rebuild.

  PTC_FUSE_GATHER=0  serialized attention gathers / un-gathers rows with ptc_gather_rows instead of
                     folding the permutation into the qkv / proj GEMMs (kv = 1 gather tables)
  PTC_SORT_POINTS=0  PT-v3m1 keeps the caller's (dataloader) row order at stage 0 instead of physically
                     sorting the points along the first serialization curve for L2 locality
  PTC_FUSE_MLP=0     the MLP runs fc1, GELU, fc2 as three kernels (+2 in the backward) instead of fusing GELU into
                     fc1's epilogue and GELU' into the epilogue of fc2's input gradient
  PTC_PREFETCH_LEVELS=0  every SerializedPooling fetches its own sizes (two host syncs per stage) instead of the one
                     up-front copy of all level sizes (ptc_pool_level_counts)
  PTC_RPE_KERNEL=0   the RPE attention branch (enable_flash=False, enable_rpe=True) keeps the dense [P,H,K,K] torch
                     formulation under bf16 autocast instead of the window-attention kernels of csrc/attention_rpe.h
  PTC_EXEC_BLOCK=0   a PT-v3m1 Block is enqueued by ~16 Python autograd Functions (the fused joints below) instead of one C call per
                     direction (csrc/block_exec.hip: same kernels, same operands, bit-identical; ~20 ms less host time per step)
  PTC_WGRAD_BLK=0    the weight gradient of the 32 / 64-channel submanifold convolutions runs on the global-gather kernel (wgrad2) instead
                     of the block-staged, accumulator-stationary one (csrc/wgrad7.h)
  PTC_FUSE_BN_TAIL=0 SpUNet's residual block runs bn2, the residual add and the ReLU as three passes (the reference's form) instead of in
                     the BatchNorm's apply pass (ptc_batch_norm_add_act_*), and every BatchNorm site increments its step counter itself
                     instead of one multi-tensor launch per forward
  PTC_CONV8=0        3^3 convolutions of 96 channels and more on the global-gather kernel conv3 instead of the block-staged conv8
                     (round 6; forward and input gradient; profiles/r06_t_conv8_himg.txt)
  PTC_BLK_MLP_FUSED=0  the MLP of a 32- / 64-channel Block runs on the split kernels (fc1 + GELU, fc2 + joint; GELU' input gradient,
                     fc1 input gradient, two weight gradients) instead of csrc/mlp.hip's one kernel per direction (round 6)
  PTC_FUSE_BLOCK=0   the three residual joints of a PTv3 Block run as separate LayerNorm / add / cast
                     kernels instead of the fused add_norm passes
"""
from __future__ import annotations

import os


def _flag(name: str, default: bool) -> bool:
    v = os.environ.get(name)
    if v is None:
        return default
    return v.strip().lower() not in ("0", "false", "off", "no", "")


FUSE_GATHER = _flag("PTC_FUSE_GATHER", True)
SORT_POINTS = _flag("PTC_SORT_POINTS", True)
FUSE_BLOCK = _flag("PTC_FUSE_BLOCK", True)
EXEC_BLOCK = _flag("PTC_EXEC_BLOCK", True)
FUSE_MLP = _flag("PTC_FUSE_MLP", True)
CONV8 = _flag("PTC_CONV8", True)                      # also read by the C side (ptc_spconv_fwd_blk)
MLP_ONE_KERNEL = _flag("PTC_BLK_MLP_FUSED", True)     # the same variable switches the block executor's C side (block_exec.hip)
PREFETCH_LEVELS = _flag("PTC_PREFETCH_LEVELS", True)
RPE_KERNEL = _flag("PTC_RPE_KERNEL", True)
WGRAD_BLK = _flag("PTC_WGRAD_BLK", True)
FUSE_BN_TAIL = _flag("PTC_FUSE_BN_TAIL", True)
BATCH_BN_COUNTERS = _flag("PTC_BATCH_BN_COUNTERS", True)
