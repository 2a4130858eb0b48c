#!/usr/bin/env python3
"""Per-kernel dynamic instruction mix from the SQ counter passes of tools/pmc_insts.sh: averages per
launch, and per wave (counter / SQ_WAVES)."""
import csv, glob, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main(folder):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(folder + "/p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_MFMA", "SQ_INSTS_BRANCH",
            "SQC_ICACHE_REQ", "SQC_ICACHE_MISSES"]
    print("%-44s %7s %7s | per wave: %s | %s" % ("kernel", "launch", "waves", " ".join("%6s" % c.replace("SQ_INSTS_", "").replace("SQC_ICACHE_", "IC_")[:6] for c in cols),
                                                 "busy_cyc  wait_inst/wave  valu_active/wave"))
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
        a, c = acc[k], cnt[k]
        avg = lambda name: a[name] / c[name] if c.get(name) else float("nan")
        waves = avg("SQ_WAVES")
        if not waves or waves != waves:
            continue
        print("%-44s %7d %7.0f |           %s | %8.0f  %8.0f  %8.0f" % (
            k[:44], c["SQ_WAVES"] // 2 or c["SQ_WAVES"], waves, " ".join("%6.0f" % (avg(n) / waves) for n in cols),
            avg("SQ_BUSY_CYCLES"), avg("SQ_WAIT_INST_ANY") / waves, avg("SQ_ACTIVE_INST_VALU") / waves))


if __name__ == "__main__":
    main(sys.argv[1])
