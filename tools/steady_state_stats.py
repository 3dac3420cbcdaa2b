#!/usr/bin/env python
"""Per-kernel totals of ONE steady-state step: the difference of two `rocprofv3 --kernel-trace` runs of the same command with
different --steps, divided by the step difference.  Model construction, parameter upload, optimizer-state zero fills and the
first step's table uploads (several hundred small fill / copy launches) cancel instead of being averaged into "per step".

    python tools/steady_state_stats.py <dir of the K1-step run> <K1> <dir of the K2-step run> <K2> [out.csv]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_top import load, short  # noqa: E402

d1, k1, d2, k2 = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
out = sys.argv[5] if len(sys.argv) > 5 else None
a1, a2 = load(d1), load(d2)
dk = float(k2 - k1)
rows = []
for name in set(a1) | set(a2):
    c = (a2[name][0] if name in a2 else 0) - (a1[name][0] if name in a1 else 0)
    t = (a2[name][1] if name in a2 else 0.0) - (a1[name][1] if name in a1 else 0.0)
    if c > 0:
        rows.append((short(name), c / dk, t / dk / 1e6, t / c / 1e3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
lines = ["name,calls_per_step,total_ms_per_step,avg_us,pct"]
for n, c, t, avg in rows:
    lines.append(f"\"{n}\",{c:.1f},{t:.3f},{avg:.1f},{100 * t / tot:.2f}")
lines.append(f"\"TOTAL\",{sum(r[1] for r in rows):.1f},{tot:.3f},,100")
if out:
    open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:int(os.environ.get("TOP", "40"))] + lines[-1:]))
