#!/usr/bin/env python
"""HBM traffic of ONE steady-state training step from rocprofv3 PMC passes.

    python tools/step_traffic.py <fetch dir K1> <fetch dir K2> <write dir K1> <write dir K2> <K1> <K2> [out.txt]

Each directory is the output of `rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) `-- python bench.py --steps K ...`: FETCH_SIZE and
WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md, counter table), hence four runs.  As in tools/steady_state_stats.py the
per-step figure is the DIFFERENCE of a K1-step and a K2-step run divided by K2 - K1, so model construction, optimizer-state fills
and the first step's table uploads cancel.  Units and the gfx950 correction follow the guide's HBM section: both counters are
KiB, FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced reads, so read bytes = FETCH_SIZE x 1024 x 2 (an upper
bound for kernels whose reads are narrow: those are counted at face value by the hardware and doubled here all the same).
The per-family table names where the step's bytes go next to SURVEY 8(d)'s compulsory ~24 GB.
"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_top import short  # noqa: E402


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r.get("Kernel_Name", "")]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    for f in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
        db = sqlite3.connect(f)
        for name, cn, val in db.execute("select name, counter_name, counter_value from pmc_events"):
            if cn == counter:
                a = acc[name]
                a[0] += 1
                a[1] += float(val)
    return acc


FAMILIES = [("attention", ("attn_",)), ("linear (fwd / dgrad GEMMs, incl. the GEMMs with a residual joint in their epilogue)", ("linear2_kernel", "linear2_joint_kernel", "conv3_kernel", "conv2_kernel", "mlp_fwd_kernel", "mlp_bwd_kernel", "gemm3_kernel")),
            ("gather convolution fwd / dgrad", ("conv7_kernel", "conv5_kernel")),
            ("weight gradients + reductions", ("wgrad", )), ("residual joints + LayerNorm", ("add_norm", "layer_norm")),
            ("BatchNorm", ("batch_norm", "bn_")), ("keys / sorts / maps / rulebooks", ("serialize", "rs_", "radix", "rulebook", "hash_", "pad_maps", "pool_", "scan", "coord_max", "maps")),
            ("rows (gather / segment)", ("gather_rows", "segment_csr", "rows_")), ("losses", ("lovasz", "cross_entropy", "column_sum")),
            ("optimizer + casts", ("multi_tensor", "weight_layouts", "cast_many", "adamw", "Adam"))]


def family(name):
    for fam, pats in FAMILIES:
        if any(p in name for p in pats):
            return fam
    return "ATen remainder"


def main():
    f1, f2, w1, w2 = sys.argv[1:5]
    k1, k2 = int(sys.argv[5]), int(sys.argv[6])
    out = sys.argv[7] if len(sys.argv) > 7 else None
    dk = float(k2 - k1)
    per = defaultdict(lambda: [0.0, 0.0, 0.0])          # kernel -> [launches, read bytes, written bytes] per step
    for a1, a2, col, scale in ((load(f1, "FETCH_SIZE"), load(f2, "FETCH_SIZE"), 1, 2048.0), (load(w1, "WRITE_SIZE"), load(w2, "WRITE_SIZE"), 2, 1024.0)):
        for name in set(a1) | set(a2):
            c = (a2[name][0] if name in a2 else 0) - (a1[name][0] if name in a1 else 0)
            v = (a2[name][1] if name in a2 else 0.0) - (a1[name][1] if name in a1 else 0.0)
            if c > 0:
                per[name][0] = c / dk
                per[name][col] += v * scale / dk
    fam = defaultdict(lambda: [0.0, 0.0, 0.0])
    for name, (c, rd, wr) in per.items():
        f = fam[family(name)]
        f[0] += c
        f[1] += rd
        f[2] += wr
    tot_r = sum(v[1] for v in fam.values())
    tot_w = sum(v[2] for v in fam.values())
    lines = [f"# HBM traffic of one steady-state step (difference of a {k1}-step and a {k2}-step PMC run / {k2 - k1}); read = FETCH_SIZE x 1024 x 2, written = WRITE_SIZE x 1024",
             f"TOTAL  read {tot_r / 1e9:8.2f} GB  written {tot_w / 1e9:8.2f} GB  sum {(tot_r + tot_w) / 1e9:8.2f} GB   (SURVEY 8(d) compulsory: ~24 GB -> ratio {(tot_r + tot_w) / 24e9:.2f}; at 8 TB/s: {(tot_r + tot_w) / 8e12 * 1e3:.2f} ms)",
             "family,launches_per_step,read_GB,written_GB,sum_GB,pct"]
    for f, (c, rd, wr) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        lines.append(f"\"{f}\",{c:.0f},{rd / 1e9:.3f},{wr / 1e9:.3f},{(rd + wr) / 1e9:.3f},{100 * (rd + wr) / (tot_r + tot_w):.1f}")
    lines.append("kernel,launches_per_step,read_MB,written_MB,sum_MB")
    for name, (c, rd, wr) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:int(os.environ.get("TOP", "40"))]:
        lines.append(f"\"{short(name)}\",{c:.1f},{rd / 1e6:.1f},{wr / 1e6:.1f},{(rd + wr) / 1e6:.1f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
