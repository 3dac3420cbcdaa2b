"""Fingerprint of every gfx950 kernel inside libptcore.so: name -> (sha1 of the kernel's machine code, VGPRs, SGPRs, LDS, scratch).

Why: a change to one .hip file must not move the code of kernels that were validated on hardware.  Run before and after an edit
(`python tools/kernel_fingerprint.py --save /tmp/a.json`, then `--diff /tmp/a.json`): anything listed as CHANGED has to go back
through the GPU tests; NEW / REMOVED are listed separately.  Works without a GPU (reads the ELF code objects of the fat binary).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path: str):
    """every gfx950 ELF embedded in the (uncompressed) offload bundles of `path`"""
    blob = open(path, "rb").read()
    pos, out = 0, []
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernels_of(elf: bytes):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(elf)
        name = f.name
    try:
        sym = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", "--demangle", name], capture_output=True, text=True, check=True).stdout
        sec = subprocess.run([f"{LLVM}/llvm-readelf", "-SW", name], capture_output=True, text=True, check=True).stdout
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", name], capture_output=True, text=True).stdout
    finally:
        os.unlink(name)
    text_addr = text_off = None
    for line in sec.splitlines():
        parts = line.replace("[", " ").replace("]", " ").split()
        if len(parts) > 5 and parts[1] == ".text":
            text_addr, text_off = int(parts[3], 16), int(parts[4], 16)
    res = {}
    for line in sym.splitlines():
        parts = line.split(None, 7)
        if len(parts) == 8 and parts[3] == "FUNC" and parts[6].isdigit():
            addr, size, nm = int(parts[1], 16), int(parts[2]), parts[7]
            if size and text_addr is not None:
                a = addr - text_addr + text_off
                res[nm] = {"sha1": hashlib.sha1(elf[a:a + size]).hexdigest()[:16], "bytes": size}
    # register / LDS use from the metadata note (msgpack rendered as YAML by readelf)
    cur = None
    meta = {}
    for line in notes.splitlines():
        t = line.strip()
        if t.startswith("- .agpr_count") or t.startswith("- .args"):
            cur = {}
            meta_list = meta.setdefault("_list", [])
            meta_list.append(cur)
        if cur is not None and t.lstrip("- ").startswith("."):
            k, _, v = t.lstrip("- ").partition(":")
            cur[k.strip()] = v.strip()
    for m in meta.get("_list", []):
        nm = m.get(".name", "").strip("'\"")
        for full, rec in res.items():
            pass
        m["_name"] = nm
    return res, meta.get("_list", [])


def fingerprint(path: str):
    out = {}
    for elf in code_objects(path):
        res, metas = kernels_of(elf)
        by_mangled = {}
        for m in metas:
            by_mangled[m.get("_name", "")] = m
        out.update(res)
        for m in metas:   # attach resource use under the mangled name (stable, greppable)
            nm = m.get("_name")
            if nm:
                out.setdefault("meta:" + nm, {}).update({k: m.get(k) for k in (".vgpr_count", ".sgpr_count", ".agpr_count", ".group_segment_fixed_size",
                                                                                ".private_segment_fixed_size") if k in m})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pointcept_amd", "libptcore.so"))
    ap.add_argument("--save")
    ap.add_argument("--diff")
    ap.add_argument("--spills", action="store_true", help="list the kernels that use scratch memory (register spills) with their register counts")
    a = ap.parse_args()
    fp = fingerprint(a.lib)
    if a.spills:
        rows = [(int(v.get(".private_segment_fixed_size") or 0), int(v.get(".vgpr_count") or 0), int(v.get(".group_segment_fixed_size") or 0), k[5:])
                for k, v in fp.items() if k.startswith("meta:")]
        rows = sorted((r for r in rows if r[0] > 0), reverse=True)
        print(f"{len(rows)} of {sum(1 for k in fp if k.startswith('meta:'))} kernels use scratch (bytes per lane | VGPRs | static LDS | kernel)")
        for sc, vg, lds, nm in rows:
            dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
            print(f"  {sc:5d} | {vg:3d} | {lds:6d} | {dem[:120]}")
        return
    if a.save:
        json.dump(fp, open(a.save, "w"), indent=0, sort_keys=True)
        print(f"{len([k for k in fp if not k.startswith('meta:')])} kernels / device functions -> {a.save}")
    if a.diff:
        old = json.load(open(a.diff))
        changed = sorted(k for k in fp if k in old and fp[k] != old[k])
        new = sorted(k for k in fp if k not in old)
        gone = sorted(k for k in old if k not in fp)
        for tag, lst in (("CHANGED", changed), ("NEW", new), ("REMOVED", gone)):
            print(f"{tag}: {len(lst)}")
            for k in lst:
                print("   ", k[:160])
        sys.exit(1 if changed or gone else 0)
    if not a.save and not a.diff:
        for k in sorted(fp):
            print(k[:150], fp[k])


if __name__ == "__main__":
    main()
