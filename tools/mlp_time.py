#!/usr/bin/env python
"""Timing sweeps of csrc/mlp.hip at the stage-0 shapes: persistent-workgroup counts (PTC_MLP_FWD_WGS / PTC_MLP_BWD_WGS), build variants
(PTC_LIB_VARIANT).  python tools/mlp_time.py [c ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcept_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    cs = [int(v) for v in sys.argv[1:]] or [64, 32]
    n = int(os.environ.get("MLP_N", "819200"))
    dt = torch.bfloat16
    print(f"variant {os.environ.get('PTC_LIB_VARIANT', '-')} n {n}")
    for c in cs:
        hid = 4 * c
        x = torch.randn(n, c, device=DEV).to(dt)
        w1 = (torch.randn(hid, c, device=DEV) / c ** 0.5).to(dt)
        w2 = (torch.randn(c, hid, device=DEV) / hid ** 0.5).to(dt)
        b1, b2 = torch.randn(hid, device=DEV) * 0.3, torch.randn(c, device=DEV)
        a = torch.randn(n, c, device=DEV)
        dm = torch.randn(n, c, device=DEV).to(dt)
        w2t = w2.t().contiguous()
        for wgs in os.environ.get("MLP_FWD_WGS", "512").split(","):
            os.environ["PTC_MLP_FWD_WGS"] = wgs
            t_j = timeit(lambda: ops.mlp_fwd(x, w1, b1, w2, b2, a, None))
            t_p = timeit(lambda: ops.mlp_fwd(x, w1, b1, w2, b2))
            print(f"c={c} fwd wgs={wgs:5s} joint {t_j:7.1f} us  plain {t_p:7.1f} us")
        for wgs in os.environ.get("MLP_BWD_WGS", "256").split(","):
            os.environ["PTC_MLP_BWD_WGS"] = wgs
            t_b = timeit(lambda: ops.mlp_bwd(dm, x, w1, b1, w2t))
            print(f"c={c} bwd wgs={wgs:5s} {t_b:7.1f} us")


if __name__ == "__main__":
    main()
