#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on MI355X: scenes/sec of one training step (forward + backward
+ optimizer) of DefaultSegmentorV2(PT-v3m1) on ScanNet-shaped synthetic scenes.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2] / SURVEY 8(d) config 3): PT-v3m1 base (46,166,272 parameters,
4 serialization orders, patch 1024), batch 8 scenes x 102,400 voxels PER GPU (weak scaling),
bf16 autocast, CrossEntropy loss (ignore_index -1), AdamW; N > 1: one process per GPU,
DistributedDataParallel over RCCL (gradient all-reduce overlapped with backward).
Inputs are generated on the host, copied to HBM BEFORE the timed region and reused every step
(rulebooks, sort, pad maps are rebuilt every step -- nothing is cached across steps).

Prints ONE JSON line on rank 0 (contract of the driver) carrying
  roofline     : the dominant kernel (serialized attention forward at the dec0/enc0 shapes),
                 timed live with HIP events on the launch stream
  cpu_baseline : the CPU oracle (oracle/ptv3_model.py, port of the reference model) timed on this
                 box's host cores on a bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")
PTV3_BASE = dict(  # configs/scannet/semseg-pt-v3m1-0-base.py:15-47
    in_channels=6, order=ORDERS, stride=(2, 2, 2, 2), enc_depths=(2, 2, 2, 6, 2), enc_channels=(32, 64, 128, 256, 512),
    enc_num_head=(2, 4, 8, 16, 32), enc_patch_size=(1024,) * 5, dec_depths=(2, 2, 2, 2), dec_channels=(64, 64, 128, 256),
    dec_num_head=(4, 4, 8, 16), dec_patch_size=(1024,) * 4, mlp_ratio=4, qkv_bias=True, drop_path=0.3,
    shuffle_orders=True, pre_norm=True, enable_flash=True, upcast_attention=False, upcast_softmax=False)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=102400, help="voxels per scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lovasz", action="store_true",
                    help="criteria = CE + Lovasz-Softmax as in scannet/semseg-pt-v3m1-0-base.py:49-52 (default: CE only)")
    ap.add_argument("--model", default="ptv3", choices=["ptv3", "spunet"],
                    help="ptv3 = BASELINE.json metric (configs[2]); spunet = configs[1] (SpUNet-v1m1, 100000 voxels/scene), "
                         "reported with its own metric name")
    ap.add_argument("--cpu-sample-points", type=int, default=20480)
    return ap.parse_args()


def attention_roofline(device, scenes: int, points: int):
    """Time attn_fwd_kernel alone at the shape of the largest attention of the model (dec0:
    C=64 -> H=4, N' = scenes*points padded to patches of 1024) with HIP events on the launch
    stream.  Algorithmic flops per launch = 4 L^2 D per (sequence, head)."""
    from pointcept_amd import ops

    L, H, D = 1024, 4, 16
    n_seq = scenes * ((points + L - 1) // L)
    T = n_seq * L
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = torch.randn(T, 3, H, D, generator=g).to(torch.bfloat16).to(device)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=device)
    scale = D ** -0.5
    for _ in range(3):
        ops.attn_varlen_fwd(qkv, cu, L, scale)
    iters = 10
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        ops.attn_varlen_fwd(qkv, cu, L, scale)
    stop.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(stop) / iters
    flops = 4.0 * L * L * D * n_seq * H
    achieved = flops / (ms * 1e-3) / 1e12
    out = {"kernel": "attn_fwd_kernel", "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
           "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
           "launch_ms": round(ms, 4), "shape": {"n_seq": n_seq, "L": L, "H": H, "D": D},
           "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": T * H * D * 2 * 4}
    # HBM bytes per launch of this kernel at this shape, from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 on
    # gfx950 + WRITE_SIZE, separate --pmc runs: tools/gpu_session.sh roof); not re-measured inside bench.py
    try:
        import glob

        # session tags run r01_a .. r01_z, r01_aa ..: order by (length, name) to get the most recent one
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*attn_pmc.json")),
                    key=lambda q: (len(os.path.basename(q)), os.path.basename(q)))[-1]
        k = json.load(open(pm))["kernels"]["attn_fwd_kernel"]
        if (n_seq, H) == (800, 4):
            out["traffic"] = round(k["hbm_bytes"])
            out["traffic_source"] = os.path.relpath(pm, ROOT)
            out["mfma_busy_frac"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES_mean"] * 32 / (k["GRBM_GUI_ACTIVE_mean"] * 1024), 3)
    except Exception:
        pass
    return out


def cpu_baseline(sample_points: int, scene_points: int):
    """The oracle port of the reference model (fp32, flash-branch semantics in fp32 math) on the host
    cores: one forward+backward of the SAME PT-v3m1 base architecture on one scene of
    `sample_points` voxels, scaled by points to scenes of `scene_points`."""
    from oracle import ptv3_model as om
    from pointcept_amd import synthetic

    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = om.SegmentorV2(20, 64, om.PointTransformerV3(**{k: v for k, v in PTV3_BASE.items()
                                                          if k not in ("enable_flash", "upcast_attention", "upcast_softmax")}))
    net.train()
    warm = {k: torch.from_numpy(v) for k, v in synthetic.collate([synthetic.indoor_scene(1, 2048)]).items()}
    net(warm)["loss"].backward()
    batch = {k: torch.from_numpy(v) for k, v in synthetic.collate([synthetic.indoor_scene(0, sample_points)]).items()}
    t0 = time.perf_counter()
    net(batch)["loss"].backward()
    dt = time.perf_counter() - t0
    n = int(batch["offset"][-1])
    value = (n / scene_points) / dt
    return {"value": round(value, 5), "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"1 scene x {n} voxels, 1 fwd+bwd of PT-v3m1 base fp32 on the CPU oracle in {dt:.2f} s, "
                      f"scaled by voxels to {scene_points}-voxel scenes"}


def main_spunet(args, rank, local_rank, world, device):
    """BASELINE configs[1]: SpUNet-v1m1 (configs/scannet/semseg-spunet-v1m1-0-base.py:12-18), batch 8 x 100000 voxels,
    CE loss, SGD(momentum 0.9, nesterov) as at :36, bf16 autocast (the reference runs fp16 AMP)."""
    import torch.nn.functional as F

    from pointcept_amd import functional as PF
    from pointcept_amd import synthetic
    from pointcept_amd.sparse_unet import SpUNetBase

    points = 100000 if args.points == 102400 else args.points
    model = SpUNetBase(6, 20, channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2)).to(device).train()
    n_params = sum(p.numel() for p in model.parameters())
    step_model = model
    if world > 1:
        step_model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
    batch = synthetic.to_torch(synthetic.indoor_batch(args.batch, points, rank=rank), device)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = step_model(dict(batch))
            loss = PF.cross_entropy(logits, batch["segment"], -1)
        loss.backward()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "scenes/sec (fwd+bwd+optimizer) SpUNet-v1m1 ScanNet-semseg @ 100k voxels", "value": round(args.batch * world * args.steps / dt, 4),
            "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SpUNet-v1m1 (39.2M params) + CE, fwd+bwd+SGD, {args.batch} scenes x {points} voxels per GPU",
                       "global_batch": args.batch * world, "params": n_params, "parallelism": f"dp{world}",
                       "final_loss": round(float(loss.detach()), 4)}}), flush=True)
    if world > 1:
        torch.distributed.barrier(device_ids=[local_rank])
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)

    from pointcept_amd import synthetic
    from pointcept_amd.point_transformer_v3 import PointTransformerV3
    from pointcept_amd.segmentor import DefaultSegmentorV2

    torch.manual_seed(1234)  # identical initial weights on every rank
    if args.model == "spunet":
        return main_spunet(args, rank, local_rank, world, device)
    criteria = ("ce", "lovasz") if args.lovasz else ("ce",)
    model = DefaultSegmentorV2(20, 64, PointTransformerV3(**PTV3_BASE), criteria=criteria).to(device).train()
    n_params = sum(p.numel() for p in model.parameters())
    step_model = model
    if world > 1:
        step_model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False)
    # AdamW of scannet/semseg-pt-v3m1-0-base.py:56; fused=True = the same update in one multi-tensor kernel per group
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)

    batch = synthetic.to_torch(synthetic.indoor_batch(args.batch, args.points, rank=rank), device)
    n_points = int(batch["offset"][-1])
    torch.manual_seed(100 + rank)  # order shuffles / DropPath differ per rank, as in training

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = step_model(dict(batch))["loss"]
        loss.backward()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    last_loss = float(loss.detach())

    if rank == 0:
        scenes_total = args.batch * world * args.steps
        out = {
            "metric": "scenes/sec (fwd+bwd+optimizer) PTv3 ScanNet-semseg @ ~100k pts",
            "value": round(scenes_total / dt, 4),
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "PT-v3m1 base (46.2M params) + seg head + " + ("CE + Lovasz" if args.lovasz else "CE") + ", fwd+bwd+AdamW, "
                                   f"{args.batch} scenes x {args.points} voxels per GPU, patch 1024, 4 orders",
                       "global_batch": args.batch * world, "points_per_gpu": n_points, "parallelism": f"dp{world}",
                       "params": n_params, "loss": "CrossEntropy(ignore_index=-1)" + (" + LovaszSoftmax" if args.lovasz else ""), "final_loss": round(last_loss, 4)},
        }
        try:
            out["roofline"] = attention_roofline(device, args.batch, args.points)
        except Exception as e:  # never lose the headline number to a diagnostics failure
            out["roofline"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_points, args.points)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier(device_ids=[local_rank])
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
