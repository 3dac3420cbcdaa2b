"""In-process A/B of the two forms of structure.offset2batch on the headline step (bench.py's model and batch): the form that takes
the row count from its caller (bucketize, no host sync) against repeat_interleave, alternating inside ONE process on one box.
    python tools/ab_offset2batch.py [--steps 8] [--rounds 2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from pointcept_amd import structure  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    args = bench.parse([])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(1234)
    model, opt, batch, loss_of = bench.build_ptv3(args, device, 0)
    step = bench.make_step(model, opt, batch, args.amp, loss_of, device)
    new = structure.offset2batch

    def old(offset, n=None):
        return new(offset, None)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for name, fn in (("repeat_interleave", old), ("bucketize(n)", new)):
            structure.offset2batch = fn
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            print(f"round {r} {name:18s} {(time.perf_counter() - t0) / a.steps * 1e3:7.2f} ms per step", flush=True)


if __name__ == "__main__":
    main()
