#!/bin/bash
# the torch-autograd twin of the reference on the clutter settings where the HIP path learns (blob1 at 0.05 / 0.1), to the
# full 40 k iterations: four runs side by side (tools/clutter_sweep.sh ran 18 at once and got 5-35 k iterations in 50 minutes)
iters=${1:-40000}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04; mkdir -p $out
cd $root
for I in 0.05 0.1; do for s in 0 1; do
  python tools/twin_train_gpu.py $s $iters tests/golden/backgrounds.npz:blob1 $I > $out/clutter_twin40k_blob1_${I}_$s.jsonl 2> $out/clutter_twin40k_blob1_${I}_$s.err &
done; done
wait
tail -n 1 $out/clutter_twin40k_blob1_*.jsonl
