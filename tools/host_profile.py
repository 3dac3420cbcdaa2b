#!/usr/bin/env python
"""Where does the HOST time of a step go?  cProfile of `step()` (enqueue only, one synchronize per step outside the profile) on a small
batch (2 x 20000 voxels: the GPU work is short, the step is host-bound), top functions by own time and by cumulative time."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=2)
ap.add_argument("--points", type=int, default=20000)
ap.add_argument("--steps", type=int, default=10)
mine = ap.parse_args()
sys.argv = [sys.argv[0], "--batch", str(mine.scenes), "--points", str(mine.points)]
args = bench.parse()
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model, opt, batch, loss_of = bench.build_ptv3(args, dev, 0)
step = bench.make_step(model, opt, batch, args.amp, loss_of, dev)
for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(mine.steps):
    step()
torch.cuda.synchronize()
print(f"{mine.scenes} x {mine.points}: {(time.perf_counter() - t0) / mine.steps * 1e3:.1f} ms per step un-profiled")
pr = cProfile.Profile()
for _ in range(mine.steps):
    pr.enable()
    step()
    pr.disable()
    torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print(f"==== by {key} ({mine.steps} steps)")
    print("\n".join(l[:180] for l in s.getvalue().splitlines()[4:]))
