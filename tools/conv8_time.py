#!/usr/bin/env python
"""Timing probes of csrc/conv8.h on a curve-ordered synthetic batch: python tools/conv8_time.py [c_in c_out [scenes points]]; PTC_C8_ABLATE
(bit mask, wrong results: 1 no products, 2 no weight traffic after the first tap, 4 no per-tap barrier, 8 no halo staging)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import maps as omaps  # noqa: E402
from oracle import sfc as osfc  # noqa: E402
from pointcept_amd import ops, synthetic  # noqa: E402

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
    c_in, c_out = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 96)
    scenes, points = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8, 102400)
    b = synthetic.indoor_batch(scenes, points)
    bt = omaps.offset2batch(b["offset"])
    gc = b["grid_coord"]
    code = osfc.encode_c(gc, bt, int(gc.max() + 1).bit_length(), ("hilbert",))[0]
    o = np.argsort(code, kind="stable")
    ind = torch.from_numpy(np.concatenate([bt[o, None], gc[o]], 1).astype(np.int32)).to(DEV)
    nbr = ops.rulebook_subm(ind, 3)
    blk = ops.BlockTables(nbr)
    n = nbr.shape[1]
    x = torch.randn(n, c_in, device=DEV).to(torch.bfloat16)
    w = (torch.randn(c_out, 27, c_in, device=DEV) / (27 * c_in) ** 0.5).to(torch.bfloat16)
    pairs = int((nbr >= 0).sum())
    t3 = timeit(lambda: ops.spconv_fwd(x, w, None, nbr))
    line = f"n={n} {c_in}->{c_out} conv3 {t3:7.1f} us"
    for abl in os.environ.get("C8_ABL", "0").split(","):
        os.environ["PTC_C8_ABLATE"] = abl
        t8 = timeit(lambda: ops.spconv_fwd(x, w, None, nbr, blk))
        line += f" | conv8[abl={abl}] {t8:7.1f} us ({2.0 * pairs * c_in * c_out / t8 / 1e6:.0f} TF/s)"
        if int(abl) & 64:
            y = ops.spconv_fwd(x, w, None, nbr, blk)
            nb = (n + 127) // 128
            ph = y.view(-1)[: nb * 128 * c_out].view(nb, 128 * c_out)[:, :32].contiguous().view(torch.int64).double().cpu()     # [block][phase]
            names = ("prologue", "chunk entry", "halo image", "last tap tiles", "epilogue", "tiles + fetch issue", "entry wait", "W wait")
            line += "\n    phases of wave 0, cycles per block (mean over blocks): " + ", ".join(f"{nm} {ph[:, i].mean():.0f}" for i, nm in enumerate(names)) \
                    + f"; sum {ph[:, :8].sum(1).mean():.0f}"
    print(line)


if __name__ == "__main__":
    main()
