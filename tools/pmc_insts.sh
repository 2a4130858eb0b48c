#!/bin/bash
# Dynamic instruction mix per kernel (SQ counters, two passes) of the default bench step.
# usage: tools/pmc_insts.sh <tag> [bench args]   -> gpurun_out/insts_<tag>/insts.txt
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/insts_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/p1 -o pmc -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQC_ICACHE_REQ SQC_ICACHE_MISSES --kernel-trace --output-format csv -d $out/p2 -o pmc -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph "$@" > $out/p2.log 2>&1
python3 $root/tools/pmc_insts.py $out > $out/insts.txt 2>&1
rm -rf $out/p1/*/*.db $out/p2/*/*.db
