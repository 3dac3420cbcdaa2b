"""Data-parallel plumbing of the hot path: scenes shard across ranks, the only
exchange step is the gradient all-reduce, done by DistributedDataParallel over RCCL ("nccl" IS RCCL
on ROCm) with buckets overlapped with backward -- the role of pointcept/engines/launch.py:106-136
and pointcept/engines/defaults.py:22-43 in the reference.  One process per GPU; rendezvous on
127.0.0.1.  The same code runs on the gloo backend with CPU tensors for the world_size-2 tests.
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.distributed as dist


# PTC_FORCE_DDP=1: initialise the process group and wrap in DistributedDataParallel even at world size 1 -- the only way to run
# RCCL, DDP's bucket hooks and the engine's autograd Functions together on a one-GPU box (tests/test_gpu_model.py)
_FORCE = os.environ.get("PTC_FORCE_DDP", "0") == "1"


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: str | None = None, device: torch.device | None = None) -> None:
    """init_process_group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    rank, _, world = env_rank()
    if (world <= 1 and not _FORCE) or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)


def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def _gpu_local_cores(index: int) -> List[int] | None:
    """cores of the NUMA node the GPU hangs off (sysfs `local_cpulist` of its PCI function), None when unknown"""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            return _parse_cpulist(f.read()) or None
    except Exception:
        return None


def rank_core_share(local_rank: int, local_world: int, allowed: List[int], numa_cores: List[List[int] | None] | None = None,
                    reserve: int = 0) -> List[int]:
    """The cores rank `local_rank` of `local_world` ranks on this node may run on: the cores of its GPU's NUMA node (`numa_cores[r]`,
    restricted to `allowed`) divided evenly among the ranks that share that node -- used only when EVERY rank's node is known and
    has at least one core per rank that shares it (the decision is the same on all ranks, so the NUMA-local shares and the fallback never
    mix) -- else an even slice of `allowed`.  Pure function of its arguments (tests/test_dp_gloo.py); shares of different ranks are
    disjoint whenever `allowed` has at least `local_world` cores."""
    allowed = sorted(allowed)
    aset = set(allowed)

    def numa_share(r):
        mine = [c for c in numa_cores[r] if c in aset]
        peers = [q for q in range(local_world) if numa_cores[q] == numa_cores[r]]
        if not mine or len(mine) < len(peers):
            return None
        per = len(mine) // len(peers)
        k = peers.index(r)
        return mine[k * per:(k + 1) * per]

    if numa_cores is not None and len(numa_cores) >= local_world and all(c for c in numa_cores[:local_world]):
        shares = [numa_share(r) for r in range(local_world)]
        if all(shares):
            return shares[local_rank]
    per = max(1, len(allowed) // max(1, local_world))
    lo = (local_rank * per) % len(allowed)
    return allowed[lo:lo + per] or allowed


def pin_rank_to_cores(local_rank: int, local_world: int, max_threads: int = 8) -> dict:
    """One Python process per GPU shares the host (pointcept/engines/train.py:185-246 under launch.py:106-136): give every rank its own
    NUMA-local cores and a bounded intra-op thread pool, so that eight enqueue threads (~25 ms of one core per step each) and their
    OpenMP / ATen pools do not migrate across sockets or oversubscribe each other.  Returns what was done (goes into bench.py's line).
    Call it BEFORE the first parallel torch op of the process: sched_setaffinity(0) moves the calling thread (threads created later
    inherit its mask; pool threads that already exist keep theirs), and OMP_NUM_THREADS only sizes pools that are not yet built --
    torch.set_num_threads below resizes the intra-op pool either way."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return {"pinned": False}
    allowed = sorted(os.sched_getaffinity(0))
    numa = None
    if torch.cuda.is_available() and torch.cuda.device_count() >= local_world:
        numa = [_gpu_local_cores(r) for r in range(local_world)]
    cores = rank_core_share(local_rank, local_world, allowed, numa)
    os.sched_setaffinity(0, cores)
    nt = max(1, min(max_threads, len(cores)))
    os.environ["OMP_NUM_THREADS"] = str(nt)
    torch.set_num_threads(nt)
    return {"pinned": True, "cores": len(cores), "first_core": cores[0], "numa_local": bool(numa and all(numa)), "threads": nt}


def wrap_ddp(model: torch.nn.Module, device: torch.device, bucket_cap_mb: int = 25,
             find_unused_parameters: bool = False) -> torch.nn.Module:
    """DDP wrapper (defaults.py:22-43 semantics: broadcast_buffers=False, so BatchNorm statistics
    stay per-rank as in the reference).  bucket_cap_mb: PTv3's 185 MB of fp32
    gradients go out in ~8 buckets as backward produces them; xGMI is point-to-point, so the
    per-bucket ring time (~0.3 ms) hides under the remaining backward."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return model
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(
        model, device_ids=ids, broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb,
        find_unused_parameters=find_unused_parameters, gradient_as_bucket_view=True)


def scene_seeds(rank: int, scenes_per_rank: int, base_seed: int = 0) -> List[int]:
    """disjoint synthetic-scene seeds per rank: s = 1000*rank + scene_index."""
    return [base_seed + 1000 * rank + i for i in range(scenes_per_rank)]


def barrier(device: torch.device) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        if device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
