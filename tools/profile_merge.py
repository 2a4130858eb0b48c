#!/usr/bin/env python3
"""Merges one profile round (tools/profile_round.sh) into kernel_profile.json -- what bench.py reads
for `roofline.traffic`, the rocprofv3 average duration and the MFMA-busy counters:

  {"source_sha16": sha256 of csrc/*.hip, csrc/*.h, include/*.h at collection time,
   "kernels": {name: {"calls", "avg_us" (rocprofv3 --kernel-trace --stats), "hbm_bytes_per_launch"
                      ((2*FETCH_SIZE + WRITE_SIZE)*1024, MI355X_MICROARCH.md HBM section),
                      "mfma_busy_cycles", "sq_busy_cycles", "gui_active_cycles" (per launch),
                      "mfma_util" = mfma_busy / (gui_active * 256 CUs * 4 SIMDs)}}}

  python tools/profile_merge.py gpurun_out/prof_<tag>   ->  <folder>/kernel_profile.json + table
"""
import csv
import glob
import hashlib
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "tf-attend-infer-repeat_amd", "csrc", "*")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def counters(folder, sub):
    acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for path in glob.glob(folder + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    return {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}


def main(folder):
    out = {"source_sha16": source_sha16(), "kernels": {}}
    K = out["kernels"]
    dbs = glob.glob(folder + "/trace/**/*.db", recursive=True)
    if dbs:
        db = sqlite3.connect(dbs[0])
        for name, calls, avg in db.execute("select name, count(*), avg(end-start) from kernels group by name"):
            K.setdefault(short(name), {}).update(calls=calls, avg_us=round(avg / 1e3, 3))
    tr = os.path.join(folder, "pmc_traffic.json")
    if os.path.exists(tr):
        for k, v in json.load(open(tr)).items():
            K.setdefault(k, {})["hbm_bytes_per_launch"] = int(v["hbm_bytes_per_launch"])
    for k, c in counters(folder, "pmc_MFMA").items():
        d = K.setdefault(k, {})
        d["mfma_busy_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        d["sq_busy_cycles"] = c.get("SQ_BUSY_CYCLES")
        d["gui_active_cycles"] = c.get("GRBM_GUI_ACTIVE")
        if c.get("GRBM_GUI_ACTIVE"):
            d["mfma_util"] = round((c.get("SQ_VALU_MFMA_BUSY_CYCLES") or 0.0) / (c["GRBM_GUI_ACTIVE"] * 256 * 4), 5)
    json.dump(out, open(os.path.join(folder, "kernel_profile.json"), "w"), indent=1, sort_keys=True)
    print("# source_sha16 %s" % out["source_sha16"])
    print("%-64s %6s %9s %14s %14s %10s" % ("kernel", "calls", "avg_us", "hbm_MB/launch", "mfma_busy_cyc", "mfma_util"))
    for k, d in sorted(K.items(), key=lambda kv: -(kv[1].get("avg_us", 0) * kv[1].get("calls", 0))):
        print("%-64s %6s %9s %14s %14s %10s" % (k[:64], d.get("calls", ""), d.get("avg_us", ""),
                                                ("%.3f" % (d["hbm_bytes_per_launch"] / 1e6)) if "hbm_bytes_per_launch" in d else "",
                                                ("%.0f" % d["mfma_busy_cycles"]) if d.get("mfma_busy_cycles") is not None else "",
                                                d.get("mfma_util", "")))


if __name__ == "__main__":
    main(sys.argv[1])
