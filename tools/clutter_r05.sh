#!/bin/bash
# BASELINE configs[4], round 5: completes the twin-vs-HIP comparison where learning is possible at all -- blob1 at 0.05 / 0.1 to
# 4 seeds (seeds 2, 3 here; 0, 1 are in profiles/r04_clutter_*), pattern1 at 0.05 / 0.1 with 2 twin seeds -- 40 k iterations.
# Eight twin runs side by side (~45 min), then the HIP path on the same settings (seconds per run).
iters=${1:-40000}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05; mkdir -p $out
cd $root
for I in 0.05 0.1; do
  for s in 2 3; do python tools/twin_train_gpu.py $s $iters tests/golden/backgrounds.npz:blob1 $I > $out/clutter_twin_blob1_${I}_$s.jsonl 2> $out/clutter_twin_blob1_${I}_$s.err & done
  for s in 0 1; do python tools/twin_train_gpu.py $s $iters tests/golden/backgrounds.npz:pattern1 $I > $out/clutter_twin_pattern1_${I}_$s.jsonl 2> $out/clutter_twin_pattern1_${I}_$s.err & done
done
cd $root/tf-attend-infer-repeat_amd
for bg in blob1 pattern1; do for I in 0.05 0.1; do for s in 0 1 2 3; do
  python training.py -r /tmp/clut_${bg}_${I}_$s -o 1 --iterations $iters --print-every 0 --precision bf16 --seed $s --bg-path ../tests/golden/backgrounds.npz:$bg --bg-max-intensity $I > /tmp/clut.log 2>&1
  python - <<PY >> $out/clutter_hip_bf16.jsonl
import json
rows=[json.loads(l) for l in open("/tmp/clut_${bg}_${I}_$s/summary/scalars.jsonl")]
r=rows[-1]
print(json.dumps({"path": "hip bf16 backward=reference, shuffle_batch queue", "bg": "$bg", "intensity": $I, "seed": $s, "step": r["step"], "accuracy": round(r["accuracy"], 3),
                  "best": round(max(x["accuracy"] for x in rows), 3), "acc012": [round(r.get("digit_acc_%d_dig" % k, float("nan")), 2) for k in range(3)]}))
PY
done; done; done
wait
cd $root
tail -q -n 1 $out/clutter_twin_*.jsonl
cat $out/clutter_hip_bf16.jsonl
