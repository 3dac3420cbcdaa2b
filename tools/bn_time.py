#!/usr/bin/env python
"""BatchNorm + activation passes (csrc/bn.hip) at SpUNet / PT-v3 sizes: forward (reduce + finish + apply) and backward
(reduce + finish + apply), bf16 features, ReLU, with and without the residual operand; algorithmic bytes / time.
Round 5 measured a 2- and 4-rows-in-flight form of the streaming loops with this tool: not faster (profiles/r05_c_bn_rows_in_flight.txt), reverted."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    from bench_ops import timeit
    from pointcept_amd import ops

    dev = torch.device("cuda:0")
    print(f"library variant: {os.environ.get('PTC_LIB_VARIANT', '(default)')}")
    for n, c in ((800000, 96), (800000, 128), (819200, 64), (819200, 32), (200000, 128), (50000, 256)):
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(n, c, device=dev, generator=g).to(torch.bfloat16)
        r = torch.randn(n, c, device=dev, generator=g).to(torch.bfloat16)
        dy = torch.randn(n, c, device=dev, generator=g).to(torch.bfloat16)
        w, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        for res in (None, r):
            y, mean, rstd = ops.batch_norm_act_fwd(x, w, b, None, None, True, 0.01, 1e-3, "relu", res=res)
            tf = timeit(lambda: ops.batch_norm_act_fwd(x, w, b, None, None, True, 0.01, 1e-3, "relu", res=res), iters=20)
            tb = timeit(lambda: ops.batch_norm_act_bwd(dy, x, w, b, mean, rstd, True, "relu", res=res), iters=20)
            e = n * c * 2
            bf = e * (3 + (1 if res is not None else 0))                     # x twice (statistics, apply) + y (+ res)
            bb = e * (5 + (3 if res is not None else 0))                     # x, dy twice + dx (+ res twice + dres)
            print(f"n={n:7d} c={c:3d} res={'y' if res is not None else 'n'} | fwd {tf * 1e6:7.1f} us {bf / tf / 1e12:5.2f} TB/s | "
                  f"bwd {tb * 1e6:7.1f} us {bb / tb / 1e12:5.2f} TB/s")


if __name__ == "__main__":
    main()
