#!/bin/bash
# training.py with one sampler-backward order up to an iteration and another one after it (bf16), over seeds:
# usage: tools/late_backward_sweep.sh <iterations> <out.jsonl> <first order> <late order> <switch iteration> seeds...
iters=$1; out=$2; first=$3; late=$4; at=$5; shift 5
mkdir -p "$(dirname "$out")"
cd "$(dirname "$0")/../tf-attend-infer-repeat_amd"
for s in "$@"; do
  python training.py -r /tmp/late_${first}_${late}_$s -o 1 --iterations $iters --print-every 0 --precision bf16 --seed $s --backward $first --late-backward $late --late-backward-from $at > /tmp/late.log 2>&1
  python - <<PY >> "../$out"
import json
rows=[json.loads(l) for l in open("/tmp/late_${first}_${late}_$s/summary/scalars.jsonl")]
first=next((r["step"] for r in rows if r["accuracy"]>=0.98), None)
print(json.dumps({"backward":"$first","late_backward":"$late","switch_at":$at,"precision":"bf16","seed":$s,"iterations":$iters,"final_accuracy":rows[-1]["accuracy"],"best_accuracy":max(r["accuracy"] for r in rows),"first_step_at_98pct":first,"wall_s":rows[-1]["wall_s"]}))
PY
done
