#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench command, then three PMC
# passes (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots, they cost 3 + 2; the SQ /
# GRBM counters for MFMA-busy ride in a third).  Counters are collected with --kernel-trace only.
# usage: tools/profile_round.sh <tag> [bench args...]      (outputs under gpurun_out/prof_<tag>/)
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras "$@" > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o pmc -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph "$@" > $out/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_MFMA -o pmc -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph "$@" > $out/pmc_MFMA.log 2>&1
find $out -type f | head -80 > $out/files.txt
python3 $root/tools/rocprof_summary.py $(find $out/trace -name '*.db' | head -1) > $out/kernel_stats.txt 2>&1
python3 $root/tools/pmc_traffic.py $out > $out/pmc_traffic.txt 2>&1
python3 $root/tools/profile_merge.py $out > $out/kernel_profile.txt 2>&1
# keep only the small summaries (gpurun_out merges <= 64 MiB)
find $out -name '*.db' -size +20M -delete
