#!/bin/bash
# Accuracy evidence (GPU box): 8 seeds x 120 k iterations per precision with the default backward, then one
# full-length run (300 epochs = 276 k iterations, training.py defaults) per precision with its accuracy curve.
cd "$(dirname "$0")/.."
tools/seed_sweep2.sh 120000 gpurun_out/r02_sweep_120k.jsonl "fp32 bf16" "reference" 0 1 2 3 4 5 6 7
cd tf-attend-infer-repeat_amd
for p in fp32 bf16; do
  python training.py -r /tmp/full_$p -o 1 --print-every 0 --precision $p --seed 0 > /tmp/full_$p.log 2>&1
  python - <<PY
import json
rows=[json.loads(l) for l in open("/tmp/full_$p/summary/scalars.jsonl")]
keep=[{"step":r["step"],"wall_s":r["wall_s"],"accuracy":r["accuracy"],"loss":r["loss"]} for r in rows if r["step"]%1000==0]
open("../gpurun_out/r02_training_accuracy_$p.jsonl","w").write("\n".join(json.dumps(k) for k in keep)+"\n")
PY
  tail -2 /tmp/full_$p.log
done
