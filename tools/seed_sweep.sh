#!/bin/bash
# Success-rate sweep of training.py over seeds and GEMM precisions (run on the GPU box).
# usage: tools/seed_sweep.sh <iterations> <out.jsonl> seeds...
iters=$1; out=$2; shift 2
cd "$(dirname "$0")/../tf-attend-infer-repeat_amd"
for p in fp32 bf16; do for s in "$@"; do
  python training.py -r /tmp/sweep_${p}_$s -o 1 --iterations $iters --print-every 0 --precision $p --seed $s > /tmp/sweep.log 2>&1
  python - <<PY >> "../$out"
import json
rows=[json.loads(l) for l in open("/tmp/sweep_${p}_$s/summary/scalars.jsonl")]
first=next((r["step"] for r in rows if r["accuracy"]>=0.98), None)
print(json.dumps({"precision":"$p","seed":$s,"iterations":$iters,"final_accuracy":rows[-1]["accuracy"],"best_accuracy":max(r["accuracy"] for r in rows),"first_step_at_98pct":first,"wall_s":rows[-1]["wall_s"]}))
PY
done; done
