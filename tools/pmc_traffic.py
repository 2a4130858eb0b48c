#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).

Corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads, so the read side is
doubled: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Infinity-Cache hits are counted, so
this is memory-side L2 traffic, an upper bound on HBM bytes.

  python tools/pmc_traffic.py gpurun_out/prof_<tag>   ->   table + JSON (kernel -> bytes/launch)
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def collect(folder, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(folder + "/pmc_%s/**/*counter_collection.csv" % counter, recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}, cnt


def main(folder):
    fetch, nf = collect(folder, "FETCH_SIZE")
    write, nw = collect(folder, "WRITE_SIZE")
    out = {}
    print("%-72s %8s %14s %14s %14s" % ("kernel", "launches", "FETCH_KiB", "WRITE_KiB", "hbm_MB/launch"))
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
        b = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
        out[k] = {"fetch_KiB": fetch.get(k), "write_KiB": write.get(k), "hbm_bytes_per_launch": b, "launches": nf.get(k, nw.get(k))}
        print("%-72s %8d %14.1f %14.1f %14.3f" % (k[:72], nf.get(k, nw.get(k, 0)), fetch.get(k, 0), write.get(k, 0), b / 1e6))
    json.dump(out, open(folder + "/pmc_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
