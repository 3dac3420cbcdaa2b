"""Static instruction mix of a kernel's hottest loop from `hipcc -S` output (no GPU needed).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 [-fno-slp-vectorize] -S --cuda-device-only -o /tmp/k.s pointcept_amd/csrc/attention.hip
    python tools/static_loop_mix.py /tmp/k.s attn_bwd_dkv_kernel [--dump]

For every basic block of every matching kernel: instruction counts by class; the block with the most MFMAs (the main loop body
after unrolling) is printed first.  Classes: mfma, trans (v_exp / v_log / v_rcp / v_rsq / v_sqrt), cvt, pk (packed f32 / 16-bit
VALU), valu (all other vector ALU), lds (ds_*), vmem (buffer_ / global_ / scratch_), salu, wait (s_waitcnt), branch, other.
At an issue-bound kernel (89-92 % issue busy in the attention backward) the count of VALU + trans + cvt per MFMA is the
number to drive down.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_barrier", "s_setprio", "s_sleep", "s_nop", "s_sched")):
        return "sync"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(path: str):
    cur, out = None, OrderedDict()
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            out[cur] = OrderedDict()
            blk = "entry"
            out[cur][blk] = []
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            blk = m.group(1)
            out[cur][blk] = []
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        out[cur][blk].append(t.split(";")[0].strip())
    return out


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    for name, blocks in kernels(path).items():
        if pat not in name:
            continue
        rows = []
        for b, ins in blocks.items():
            c = Counter(classify(i.split()[0]) for i in ins)
            rows.append((c["mfma"], len(ins), b, c, ins))
        rows.sort(key=lambda r: (-r[0], -r[1]))
        print(f"== {name}")
        print("   block        instrs | mfma trans cvt  pk valu | lds vmem | salu wait sync branch")
        for mf, n, b, c, ins in rows[:4]:
            print(f"   {b:12s} {n:6d} | {c['mfma']:4d} {c['trans']:5d} {c['cvt']:3d} {c['pk']:3d} {c['valu']:4d} | {c['lds']:3d} {c['vmem']:4d} |"
                  f" {c['salu']:4d} {c['wait']:4d} {c['sync']:4d} {c['branch']:4d}")
        if dump and rows:
            top = rows[0]
            ops = Counter(i.split()[0] for i in top[4])
            print("   -- opcode histogram of", top[2])
            for op, k in ops.most_common():
                print(f"      {k:4d} {op}")


if __name__ == "__main__":
    main()
