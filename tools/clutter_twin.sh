#!/bin/bash
# BASELINE configs[4] diagnosis: the torch-autograd twin of the reference trained on the SAME cluttered data
# (backgrounds pattern1 @ 0.3) as the HIP path -- three seeds side by side on one GPU (the twin is launch-bound).
mkdir -p gpurun_out
for s in 0 1 2; do
  python tools/twin_train_gpu.py $s ${1:-50000} tests/golden/backgrounds.npz:pattern1 > gpurun_out/r03_clutter_twin_seed$s.jsonl 2> gpurun_out/r03_clutter_twin_seed$s.err &
done
wait
tail -n 3 gpurun_out/r03_clutter_twin_seed*.jsonl
