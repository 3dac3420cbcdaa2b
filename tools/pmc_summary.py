#!/usr/bin/env python
"""Summarise rocprofv3 output directories into small JSON/CSV files for profiles/.

    python tools/pmc_summary.py --stats DIR --pmc DIR [DIR ...] --kernels attn_fwd_kernel,... --out profiles/x.json

--stats DIR : a `rocprofv3 --kernel-trace --stats` output dir (reads *_kernel_stats.csv)
--pmc DIRs  : `rocprofv3 --pmc <counter>` output dirs (reads *_counter_collection.csv); per kernel and
              counter the mean value per dispatch is reported.  FETCH_SIZE / WRITE_SIZE are in KiB;
              on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x
              (MI355X_MICROARCH.md section HBM), so `hbm_read_bytes` = FETCH_SIZE * 1024 * 2.
"""
import argparse
import csv
import glob
import json
import os
import sqlite3
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--kernels", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    want = [k for k in a.kernels.split(",") if k]
    res = {"kernels": {}}

    def keep(name):
        return not want or any(w in name for w in want)

    def short(name):
        for w in want:
            if w in name:
                # keep the template arguments (distinct instances = distinct shapes), drop the parameter list
                head = name.split("(")[0].strip()
                return head.replace("void ", "") if "<" in head else w
        return name[:80]

    # rocprofv3 >= 7.x writes a rocpd SQLite database by default (views `top_kernels`, `kernels`,
    # `pmc_events`); CSV (--output-format csv) is read too.
    if a.stats:
        for f in find(a.stats, "*_results.db"):
            db = sqlite3.connect(f)
            for name, in db.execute("select distinct name from kernels"):
                if keep(name):
                    d = [r[0] for r in db.execute("select duration from kernels where name = ?", (name,))]
                    k = res["kernels"].setdefault(short(name), {})
                    k["calls"] = len(d)
                    k["avg_us"] = round(sum(d) / len(d) / 1e3, 2)
                    k["min_us"] = round(min(d) / 1e3, 2)
                    k["max_us"] = round(max(d) / 1e3, 2)
        for f in find(a.stats, "*kernel_stats.csv"):
            for r in csv.DictReader(open(f)):
                if keep(r["Name"]):
                    k = res["kernels"].setdefault(short(r["Name"]), {})
                    k["calls"] = int(r["Calls"])
                    k["avg_us"] = round(float(r["AverageNs"]) / 1e3, 2)
                    k["min_us"] = round(float(r["MinNs"]) / 1e3, 2)
                    k["max_us"] = round(float(r["MaxNs"]) / 1e3, 2)
    for d in a.pmc:
        acc = defaultdict(lambda: defaultdict(list))
        for f in find(d, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name", "")
                if keep(name):
                    acc[short(name)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in find(d, "*_results.db"):
            db = sqlite3.connect(f)
            for name, cn, val in db.execute("select name, counter_name, counter_value from pmc_events"):
                if keep(name):
                    acc[short(name)][cn].append(float(val))
        for kn, cs in acc.items():
            k = res["kernels"].setdefault(kn, {})
            for cn, vals in cs.items():
                k[cn + "_mean"] = sum(vals) / len(vals)
                k[cn + "_dispatches"] = len(vals)
    for kn, k in res["kernels"].items():
        if "FETCH_SIZE_mean" in k:
            k["hbm_read_bytes"] = k["FETCH_SIZE_mean"] * 1024 * 2  # gfx950 correction
        if "WRITE_SIZE_mean" in k:
            k["hbm_write_bytes"] = k["WRITE_SIZE_mean"] * 1024
        if "hbm_read_bytes" in k and "hbm_write_bytes" in k:
            k["hbm_bytes"] = k["hbm_read_bytes"] + k["hbm_write_bytes"]
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
