#!/usr/bin/env python
"""Static scan of a translation unit's gfx950 assembly (hipcc -S --cuda-device-only, or the .s of --save-temps) for the code-generation
patterns that cost conv8 / linear2 their matrix pipe in round 6:
  * exec-masked loads (s_and_saveexec ... ds_read / buffer_load / global_load ... s_or exec) between MFMAs -- a per-lane guard around an
    operand read: every read is waited for on its own;
  * full waits (lgkmcnt(0) / vmcnt(0)) directly in front of an MFMA;
  * runs of v_mov between MFMAs (accumulator copies: they read an MFMA's result the cycle after it issued).
    python tools/isa_stalls.py file.s [name-substring]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = re.split(r"\n(?=_Z[\w]+:\s+; @)", text)
rows = []
for f in funcs:
    m = re.match(r"(_Z\w+):", f)
    if not m:
        continue
    lines = [l for l in f.split("\n") if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
    idx = [i for i, l in enumerate(lines) if "\tv_mfma" in l]
    if len(idx) < 2:
        continue
    reg = lines[idx[0]:idx[-1] + 1]
    masked = sum(1 for i, l in enumerate(reg) if "s_and_saveexec" in l and any(re.search(r"ds_read|buffer_load|global_load", x) for x in reg[i + 1:i + 4]))
    full = sum(1 for i, l in enumerate(reg) if "v_mfma" in l and i > 0 and re.search(r"s_waitcnt.*(lgkmcnt\(0\)|vmcnt\(0\))", reg[i - 1]))
    runs, cur = 0, 0
    for l in reg:
        if re.match(r"\tv_mov_b(32|64)|\tv_accvgpr", l):
            cur += 1
        else:
            runs += cur >= 4
            cur = 0
    rows.append((m.group(1), len(idx), masked, full, runs))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'mfma':>5s} {'masked':>6s} {'full':>5s} {'movs':>5s}  kernel (between its first and last MFMA: exec-masked loads, full waits in front of an MFMA, runs of >= 4 v_mov)")
for (r, n) in sorted(zip(rows, names), key=lambda x: -(x[0][2] + x[0][4])):
    if flt in n and (r[2] or r[4] or r[3] > r[1] // 2):
        print(f"{r[1]:5d} {r[2]:6d} {r[3]:5d} {r[4]:5d}  " + re.sub(r"\(.*", "", n)[:150])
