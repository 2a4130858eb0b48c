#!/bin/bash
# follow-up of tools/clutter_r06.sh: the twin runs the 40-minute deadline cut (blob1 @ 0.05 seeds 2, 3; pattern1 @ 0.05 seeds
# 1, 2), four side by side, and the HIP path in fp32 -- the twin's precision -- on blob1 at both intensities, 8 seeds each.
iters=${1:-40000}; dl=${2:-14}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r06_clutter_b; mkdir -p $out
cd $root
for s in 2 3; do python tools/twin_train_gpu.py $s $iters tests/golden/backgrounds.npz:blob1 0.05 --deadline-min $dl > $out/twin_blob1_0.05_$s.jsonl 2> $out/twin_blob1_0.05_$s.err & done
for s in 1 2; do python tools/twin_train_gpu.py $s $iters tests/golden/backgrounds.npz:pattern1 0.05 --deadline-min $dl > $out/twin_pattern1_0.05_$s.jsonl 2> $out/twin_pattern1_0.05_$s.err & done
for I in 0.1 0.05; do
  python tools/gate_sweep.py --out $out/hip_fp32.jsonl --tag clutb --parallel 2 --arm fp32:reference:-:0:$iters:0-7 -- --bg-path ../tests/golden/backgrounds.npz:blob1 --bg-max-intensity $I
done
wait
for f in $out/twin_*.jsonl; do echo "$(basename $f): $(tail -n 1 $f)"; done
python tools/gate_report.py $out/hip_fp32.jsonl
