"""Static resource usage of every kernel instance of libptcore.so (no GPU needed): each translation unit compiled with the
library's own flags plus -Rpass-analysis=kernel-resource-usage; one line per kernel with VGPRs, AGPRs, scratch bytes per lane,
waves per SIMD and static LDS, then the totals (instances, instances with scratch, instances below 2 waves per SIMD).
    python tools/static_resources.py > profiles/rNN_static_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pointcept_amd import build as B  # noqa: E402

CSRC = os.path.join(os.path.dirname(os.path.abspath(B.__file__)), "csrc")
INC = os.path.join(os.path.dirname(os.path.abspath(B.__file__)), "..", "include")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*$", "", o) if "<" not in o else re.sub(r">\(.*$", ">", o) for o in out]


def unit(src):
    with tempfile.TemporaryDirectory() as d:
        flags = list(B.HIP_FLAGS) + B.PER_FILE_FLAGS.get(src, [])
        r = subprocess.run([B.HIPCC] + flags + ["-I", INC, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src),
                            "-o", os.path.join(d, "o.o")], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(src + "\n" + r.stderr[-2000:])
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0]] = int(v)
    return src, rows


def main():
    srcs = [s for s in B.HIP_SOURCES] if hasattr(B, "HIP_SOURCES") else sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(unit, srcs))
    print("# static resource usage of every kernel instance of libptcore.so (library flags + -Rpass-analysis=kernel-resource-usage, gfx950)")
    print("# file              VGPR AGPR scratch[B/lane] waves/SIMD staticLDS[B]  kernel   (dynamic LDS is set at launch and not shown)")
    tot = scratch = low = 0
    for src, rows in res:
        names = demangle([r["name"] for r in rows]) if rows else []
        for r, nm in zip(rows, names):
            tot += 1
            scratch += r.get("ScratchSize", 0) > 0
            low += r.get("Occupancy", 8) < 2
            print(f"{src:18s} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('ScratchSize', 0):7d} {r.get('Occupancy', 0):6d} {r.get('LDS', 0):8d}  {nm[:150]}")
    print(f"# {tot} kernel instances in {len(res)} translation units; {scratch} with scratch; {low} below two waves per SIMD")


if __name__ == "__main__":
    main()
