#!/usr/bin/env python
"""For the small fill / copy kernels of a step: which kernels run right before and after them?  Reads the kernel trace of a
`rocprofv3 --kernel-trace` run and prints, per target kernel, the histogram of (previous kernel, next kernel, grid size).

    python tools/kernel_neighbours.py <rocprof out dir> [name substring ...]
"""
import collections
import csv
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
targets = sys.argv[2:] or ["FillFunctor<float>", "copyBuffer"]
rows = []
for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], str(r.get("Grid_Size_X", r.get("Grid_Size", "?")))))
for p in glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True):   # rocprofv3's default output: a rocpd sqlite file
    cur = sqlite3.connect(p).execute("select * from kernels")
    cols = [d[0].lower() for d in cur.description]
    i_name, i_start = cols.index("name"), cols.index("start")
    i_grid = next((cols.index(c) for c in ("grid_x", "grid_size_x", "grid_size") if c in cols), None)
    for r in cur:
        rows.append((int(r[i_start]), r[i_name], str(r[i_grid]) if i_grid is not None else "?"))
if not rows:
    sys.exit(f"no kernel trace under {root}")
rows.sort()


def short(n):
    n = n.split("(")[0]
    for pre in ("void at::native::", "void "):
        if n.startswith(pre):
            n = n[len(pre):]
    return n[:70]


for t in targets:
    hist = collections.Counter()
    total = 0
    for i, (_, name, grid) in enumerate(rows):
        if t in name:
            total += 1
            prev = short(rows[i - 1][1]) if i else "-"
            nxt = short(rows[i + 1][1]) if i + 1 < len(rows) else "-"
            hist[(prev, nxt, grid)] += 1
    print(f"== {t}: {total} launches in the trace")
    for (prev, nxt, grid), c in hist.most_common(25):
        print(f"{c:6d}  grid {grid:>9s}  after {prev:70s}  before {nxt}")
    big = [(k, c) for k, c in hist.items() if k[2].isdigit() and int(k[2]) >= 4_000_000]
    if big:
        print(f"   -- launches of {t} with a grid of >= 4M work-items (large tensors), whatever their rank:")
        for (prev, nxt, grid), c in sorted(big, key=lambda kc: -int(kc[0][2])):
            print(f"{c:6d}  grid {grid:>9s}  after {prev:70s}  before {nxt}")
