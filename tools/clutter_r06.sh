#!/bin/bash
# BASELINE configs[4] (clutter), round 6: closes the twin-vs-HIP comparison on blob1 / pattern1 at max intensity 0.05 / 0.1.
#   twin (tools/twin_train_gpu.py, one hipGraph replay per iteration): the seeds each cell still lacks for 4 x 40 k
#     (blob1: seeds 2, 3 -- seeds 0, 1 at 40 k are in profiles/r04_clutter_*; pattern1: seeds 0..3), all side by side;
#   HIP path (training.py defaults): 8 seeds per cell, 40 k iterations, 4 side by side (tools/gate_sweep.py).
# usage: tools/clutter_r06.sh [iterations] [twin deadline in minutes]
iters=${1:-40000}; dl=${2:-45}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r06_clutter; mkdir -p $out
cd $root
for I in 0.05 0.1; do
  for s in 2 3; do python tools/twin_train_gpu.py $s $iters tests/golden/backgrounds.npz:blob1 $I --deadline-min $dl > $out/twin_blob1_${I}_$s.jsonl 2> $out/twin_blob1_${I}_$s.err & done
  for s in 0 1 2 3; do python tools/twin_train_gpu.py $s $iters tests/golden/backgrounds.npz:pattern1 $I --deadline-min $dl > $out/twin_pattern1_${I}_$s.jsonl 2> $out/twin_pattern1_${I}_$s.err & done
done
for bg in blob1 pattern1; do for I in 0.05 0.1; do
  python tools/gate_sweep.py --out $out/hip_bf16.jsonl --tag clut --parallel 4 --arm bf16:${HIP_ARM:-reference:-:0}:$iters:0-7 -- --bg-path ../tests/golden/backgrounds.npz:$bg --bg-max-intensity $I
done; done
wait
for f in $out/twin_*.jsonl; do echo "$(basename $f): $(tail -n 1 $f)"; done
python tools/gate_report.py $out/hip_bf16.jsonl
