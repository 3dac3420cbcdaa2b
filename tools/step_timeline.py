#!/usr/bin/env python
"""Timeline of ONE steady-state training step from a `rocprofv3 --kernel-trace` run of bench.py: where the GPU idles between kernels
and how the step's time divides over the Blocks of the model.

    python tools/step_timeline.py <rocprof out dir> [marker substring, default lovasz_keys]

The step is the span between the last two launches of the marker kernel (one per step).  Prints
  * busy time (sum of kernel durations), idle time (sum of gaps between the end of one kernel and the start of the next on the
    merged timeline), the number of gaps above 2 / 5 / 20 / 100 us and the time in each class;
  * the 30 largest gaps with the kernels on either side (host syncs, allocator calls and launch-bound stretches show up here);
  * the idle time grouped by the kernel that FOLLOWS the gap (who was late);
  * the time between consecutive attention kernels (one per Block and direction): the Block-level breakdown of the step.
"""
import collections
import csv
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "lovasz_keys"
rows = []
for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for p in glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True):
    cur = sqlite3.connect(p).execute("select * from kernels")
    cols = [d[0].lower() for d in cur.description]
    i_name, i_start, i_end = cols.index("name"), cols.index("start"), cols.index("end")
    for r in cur:
        rows.append((int(r[i_start]), int(r[i_end]), r[i_name]))
if not rows:
    sys.exit(f"no kernel trace under {root}")
rows.sort()


def short(n):
    n = n.split("(")[0]
    for pre in ("void at::native::", "void "):
        if n.startswith(pre):
            n = n[len(pre):]
    return n[:64]


marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < 2:
    sys.exit(f"marker {marker!r} found {len(marks)} times: need two steps in the trace")
lo, hi = marks[-2], marks[-1]
step = rows[lo:hi]
span = (rows[hi][0] - rows[lo][0]) / 1e3
busy = sum(e - s for s, e, _ in step) / 1e3
gaps = []
end = step[0][1]
for i in range(1, len(step) + 1):
    s, e, n = step[i] if i < len(step) else rows[hi]
    g = (s - end) / 1e3
    gaps.append((g, short(step[i - 1][2]), short(n), i))
    end = max(end, e)
idle = sum(max(g[0], 0.0) for g in gaps)
overlap = -sum(min(g[0], 0.0) for g in gaps)
print(f"step span {span / 1e3:.3f} ms, {len(step)} launches, busy {busy / 1e3:.3f} ms, idle {idle / 1e3:.3f} ms, overlapped {overlap / 1e3:.3f} ms")
for thr in (2, 5, 20, 100):
    sel = [g[0] for g in gaps if g[0] > thr]
    print(f"  gaps > {thr:3d} us: {len(sel):5d}, {sum(sel) / 1e3:.3f} ms")
small = [g[0] for g in gaps if 0 <= g[0] <= 2]
print(f"  gaps <= 2 us: {len(small):5d}, {sum(small) / 1e3:.3f} ms (mean {sum(small) / max(len(small), 1):.2f} us)")
print("\n== the 30 largest gaps (us | after | before | position)")
for g, a, b, i in sorted(gaps, reverse=True)[:30]:
    print(f"{g:9.1f}  {a:64s}  {b:64s}  #{i}")
by_next = collections.Counter()
cnt_next = collections.Counter()
for g, a, b, i in gaps:
    if g > 0:
        by_next[b] += g
        cnt_next[b] += 1
print("\n== idle time by the kernel that follows the gap (ms | gaps | kernel)")
for n, t in by_next.most_common(25):
    print(f"{t / 1e3:8.3f}  {cnt_next[n]:5d}  {n}")
print("\n== time between consecutive attention kernels (ms from step start | segment ms | launches | kernel)")
t0 = step[0][0]
last_t, last_i = t0, 0
for i, (s, e, n) in enumerate(step):
    if "attn_fwd" in n or "attn_bwd" in n:
        print(f"{(s - t0) / 1e6:8.3f}  {(s - last_t) / 1e6:7.3f}  {i - last_i:4d}  {short(n)}  [{(e - s) / 1e3:.0f} us]")
        last_t, last_i = s, i
print(f"{(rows[hi][0] - t0) / 1e6:8.3f}  {(rows[hi][0] - last_t) / 1e6:7.3f}  {len(step) - last_i:4d}  (end of step)")
seq = os.environ.get("PTC_TIMELINE_SEQ")
if seq:                                  # the whole step, one line per launch: position | start (us from step start) | gap before | duration | kernel
    with open(seq, "w") as f:
        end = step[0][0]
        for i, (s, e, n) in enumerate(step):
            f.write(f"{i:5d} {(s - t0) / 1e3:10.1f} {(s - end) / 1e3:8.1f} {(e - s) / 1e3:8.1f}  {short(n) or '(unnamed)'}\n")
            end = max(end, e)
