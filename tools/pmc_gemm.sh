#!/bin/bash
# SQ counters of one GEMM configuration (GPU box): tools/pmc_gemm.sh M N K tb prec tm tn
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_gemm; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o p -- python3 $root/tools/gemm_one.py "$@" > $out/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "gemm" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
