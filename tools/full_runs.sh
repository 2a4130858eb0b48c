#!/bin/bash
# One full-length run (300 epochs = 276 k iterations, training.py defaults) per precision with its accuracy curve:
# tools/full_runs.sh <tag>   ->  gpurun_out/<tag>_training_accuracy_{fp32,bf16}.jsonl   (run on the GPU box)
tag=${1:-r03}
cd "$(dirname "$0")/../tf-attend-infer-repeat_amd"
for p in fp32 bf16; do
  python training.py -r /tmp/full_$p -o 1 --print-every 0 --precision $p --seed 0 > /tmp/full_$p.log 2>&1
  python - <<PY
import json
rows=[json.loads(l) for l in open("/tmp/full_$p/summary/scalars.jsonl")]
keep=[{"step":r["step"],"wall_s":r["wall_s"],"accuracy":r["accuracy"],"loss":r["loss"]} for r in rows if r["step"]%1000==0]
open("../gpurun_out/${tag}_training_accuracy_$p.jsonl","w").write("\n".join(json.dumps(k) for k in keep)+"\n")
print("$p", "final", keep[-1], "first >= 0.98:", next((k["step"] for k in keep if k["accuracy"]>=0.98), None))
PY
done
