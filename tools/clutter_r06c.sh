#!/bin/bash
# the twin runs that two side-by-side attempts left unfinished, ONE AT A TIME (a twin iteration is 7 ms alone and 30-115 ms
# beside other processes: the GPU time-slices processes that never idle instead of overlapping them)
iters=${1:-40000}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r06_clutter_c; mkdir -p $out
cd $root
for spec in "blob1 0.05 2" "blob1 0.05 3" "pattern1 0.05 1" "pattern1 0.05 2" "pattern1 0.1 0" "pattern1 0.1 2" "pattern1 0.1 3"; do
  set -- $spec
  python tools/twin_train_gpu.py $3 $iters tests/golden/backgrounds.npz:$1 $2 --deadline-min 6 > $out/twin_$1_$2_$3.jsonl 2> $out/twin_$1_$2_$3.err
  echo "twin_$1_$2_$3: $(tail -n 1 $out/twin_$1_$2_$3.jsonl)"
done
