#!/usr/bin/env python
"""Per-kernel totals of a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_stats.csv) as a compact
CSV (name, calls, total_ms, avg_us, pct) -> the file committed under profiles/."""
import csv
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict


def load(d):
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        db = sqlite3.connect(f)
        for name, dur in db.execute("select name, duration from kernels"):
            a = agg[name]
            a[0] += 1
            a[1] += dur
            a[2] = min(a[2], dur)
            a[3] = max(a[3], dur)
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[r["Name"]]
            a[0] += int(r["Calls"])
            a[1] += float(r["TotalDurationNs"])
            a[2] = min(a[2], float(r["MinNs"]))
            a[3] = max(a[3], float(r["MaxNs"]))
    return agg


def short(n):
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::|c10::", "", n)
    n = re.sub(r"\(.*$", "", n) if len(n) > 110 else n
    return n[:110]


def main():
    d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    out = sys.argv[3] if len(sys.argv) > 3 else None
    agg = load(d)
    tot = sum(a[1] for a in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = ["name,calls_per_step,total_ms_per_step,avg_us,min_us,max_us,pct"]
    for n, a in rows:
        lines.append(f"\"{short(n)}\",{a[0] / steps:.1f},{a[1] / steps / 1e6:.3f},{a[1] / a[0] / 1e3:.1f},{a[2] / 1e3:.1f},{a[3] / 1e3:.1f},{100 * a[1] / tot:.2f}")
    lines.append(f"\"TOTAL\",{sum(a[0] for a in agg.values()) / steps:.1f},{tot / steps / 1e6:.3f},,,,100")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print("\n".join(lines[:int(os.environ.get("TOP", "45"))] + lines[-1:]))


if __name__ == "__main__":
    main()
