#!/bin/bash
# BASELINE configs[4]: the HIP path on the same cluttered data as tools/clutter_twin.sh (3 seeds x <iterations>), per-count accuracy
cd "$(dirname "$0")/../tf-attend-infer-repeat_amd"
iters=${1:-50000}
for s in 0 1 2; do
python training.py -r /tmp/clut_$s -o 1 --iterations $iters --print-every 0 --precision fp32 --seed $s --bg-path ../tests/golden/backgrounds.npz:pattern1 --bg-max-intensity 0.3 > /tmp/clut_$s.log 2>&1
python - <<PY >> ../gpurun_out/r03_clutter_hip_50k.jsonl
import json
rows=[json.loads(l) for l in open("/tmp/clut_$s/summary/scalars.jsonl")]
for r in rows:
    if r["step"] % 5000 == 0 or r is rows[-1]:
        print(json.dumps({"path": "hip fp32 backward=reference", "seed": $s, "step": r["step"], "accuracy": round(r["accuracy"], 3),
                          "acc012": [round(r.get("digit_acc_%d" % k, float("nan")), 2) for k in range(3)]}))
PY
done
tail -n 3 ../gpurun_out/r03_clutter_hip_50k.jsonl
