#!/bin/bash
# kernel-trace stats only (no PMC passes): tools/quick_stats.sh <tag> [bench args...]  -> gpurun_out/prof_<tag>/kernel_stats.txt
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-roofline "$@" > $out/trace.log 2>&1
python3 $root/tools/rocprof_summary.py $(find $out/trace -name '*.db' | head -1) > $out/kernel_stats.txt 2>&1
find $out -name '*.db' -size +20M -delete
head -12 $out/kernel_stats.txt
