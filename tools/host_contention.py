#!/usr/bin/env python
"""Does the host side of the step hold up when 8 ranks share one host?  (VERDICT r2 next 6 i.)  The one-GPU box cannot run 8 ranks on 8
GPUs, but the HOST work of a rank -- the Python that enqueues a step -- is the same whichever GPU it feeds: N processes run the real
step (the bench model, a small batch so that the one shared GPU is not what they wait for most of the time) concurrently and report, per
step, their own CPU time (user + system of the process with blocking -- sleeping -- GPU waits: what each rank asks of the host) and wall time.  N = 1 vs N = 8: if CPU time per
step stays put the cores do not contend; the wall time under N = 8 additionally contains the queueing on the single shared GPU.

    python tools/host_contention.py [--procs 8] [--scenes 2] [--points 20000] [--steps 12]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(a):
    import ctypes

    import torch

    # waits inside the step (the forward reads a few sizes back) must SLEEP, not spin: a spinning wait is booked as CPU time of the
    # process and, with N processes queueing on one GPU, would be mistaken for host work (hipDeviceScheduleBlockingSync = 4)
    try:
        ctypes.CDLL("libamdhip64.so.7").hipSetDeviceFlags(4)
    except OSError:
        pass
    sys.path.insert(0, ROOT)
    import bench

    if a.pin > 0:      # what bench.py's self-launcher does for every rank (round 4, dp.pin_rank_to_cores): NUMA-local, disjoint core sets
        from pointcept_amd import dp
        dp.pin_rank_to_cores(a.rank, a.procs)

    sys.argv = [sys.argv[0], "--batch", str(a.scenes), "--points", str(a.points)]
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model, opt, batch, loss_of = bench.build_ptv3(args, dev, a.rank)
    step = bench.make_step(model, opt, batch, args.amp, loss_of, dev)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    # rendezvous through the file system: start the timed region together
    open(os.path.join(a.dir, f"ready{a.rank}"), "w").close()
    while len([f for f in os.listdir(a.dir) if f.startswith("ready")]) < a.procs:
        time.sleep(0.01)
    c0, w0 = time.process_time(), time.perf_counter()
    for _ in range(a.steps):
        step()
    c1 = time.process_time()            # CPU time of the enqueue (the final wait below sleeps)
    torch.cuda.synchronize()
    w1 = time.perf_counter()
    print(json.dumps({"rank": a.rank, "cpu_ms_per_step": (c1 - c0) / a.steps * 1e3, "wall_ms_per_step": (w1 - w0) / a.steps * 1e3}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--scenes", type=int, default=2)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--dir", default="")
    ap.add_argument("--pin", type=int, default=-1, help="1 / 0: workers pin themselves like bench.py's ranks (dp.pin_rank_to_cores); default: both")
    a = ap.parse_args()
    if a.rank >= 0:
        return worker(a)
    import tempfile

    print(f"host: {os.cpu_count()} cores; workload per process: PT-v3m1 base step, {a.scenes} x {a.points} voxels, {a.steps} timed steps")
    for n, pin in ((1, 0), (a.procs, 0), (a.procs, 1)) if a.pin < 0 else ((1, a.pin), (a.procs, a.pin)):
        d = tempfile.mkdtemp(prefix="ptc_cont_")
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), "--procs", str(n), "--dir", d, "--scenes", str(a.scenes),
                                "--points", str(a.points), "--steps", str(a.steps), "--pin", str(pin)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                               text=True)
              for r in range(n)]
        rows = []
        for p in ps:
            out, _ = p.communicate(timeout=900)
            rows += [json.loads(l) for l in out.splitlines() if l.startswith("{")]
        cpu = [r["cpu_ms_per_step"] for r in rows]
        wall = [r["wall_ms_per_step"] for r in rows]
        print(f"{n} concurrent process(es) on ONE GPU, {'pinned to NUMA-local disjoint cores' if pin else 'unpinned'}: host CPU time per step mean {sum(cpu) / len(cpu):.1f} ms (max {max(cpu):.1f}); "
              f"wall per step mean {sum(wall) / len(wall):.1f} ms (max {max(wall):.1f})")
    print("reading: CPU time per step = the host work one rank needs; it must stay (about) the same under N = 8 for 8 ranks not to contend for the "
          "host's cores; on an 8-GPU node every rank has its own GPU, so its wall per step is the N = 1 wall (GPU-bound) as long as that CPU time fits under it")


if __name__ == "__main__":
    main()
