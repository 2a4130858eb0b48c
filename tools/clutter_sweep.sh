#!/bin/bash
# BASELINE configs[4], clutter in a regime where something may learn: --bg-max-intensity {0.05, 0.1, 0.2} x
# {pattern1, gray1, blob1} x 2 seeds x <iterations>, for BOTH the torch-autograd twin of the reference
# (tools/twin_train_gpu.py; launch-bound, all 18 runs side by side on the one GPU) and the HIP product path
# (training.py, default bf16, one run after the other once the twins are done).  Output: gpurun_out/r04/clutter_*.jsonl
iters=${1:-40000}
mode=${2:-both}          # twin | hip | both
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04; mkdir -p $out
cd $root
if [ "$mode" != "hip" ]; then
for bg in pattern1 gray1 blob1; do for I in 0.05 0.1 0.2; do for s in 0 1; do
  python tools/twin_train_gpu.py $s $iters tests/golden/backgrounds.npz:$bg $I > $out/clutter_twin_${bg}_${I}_$s.jsonl 2> $out/clutter_twin_${bg}_${I}_$s.err &
done; done; done
wait
fi
[ "$mode" == "twin" ] && exit 0
cd $root/tf-attend-infer-repeat_amd
for bg in pattern1 gray1 blob1; do for I in 0.05 0.1 0.2; do for s in 0 1; do
  python training.py -r /tmp/clut_${bg}_${I}_$s -o 1 --iterations $iters --print-every 0 --precision bf16 --seed $s \
    --bg-path ../tests/golden/backgrounds.npz:$bg --bg-max-intensity $I > /tmp/clut.log 2>&1
  python - <<PY >> $out/clutter_hip.jsonl
import json
rows=[json.loads(l) for l in open("/tmp/clut_${bg}_${I}_$s/summary/scalars.jsonl")]
for r in rows:
    if r["step"] % 5000 == 0 or r is rows[-1]:
        print(json.dumps({"path": "hip bf16 backward=reference", "bg": "$bg", "intensity": $I, "seed": $s, "step": r["step"], "accuracy": round(r["accuracy"], 3),
                          "acc012": [None if r.get("digit_acc_%d_dig" % k) is None else round(r["digit_acc_%d_dig" % k], 2) for k in range(3)]}))
PY
done; done; done
for f in $out/clutter_twin_*.jsonl; do b=$(basename $f .jsonl); tail -n 1 $f | sed "s/^{/{\"run\": \"$b\", /"; done > $out/clutter_twin_final.jsonl
cat $out/clutter_twin_final.jsonl; grep -c . $out/clutter_hip.jsonl
