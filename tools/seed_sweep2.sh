#!/bin/bash
# Success-rate sweep of training.py over seeds, GEMM precisions and sampler-backward modes (run on the GPU box).
# usage: tools/seed_sweep2.sh <iterations> <out.jsonl> "<precisions>" "<backward modes>" seeds...
iters=$1; out=$2; precs=$3; modes=$4; shift 4
cd "$(dirname "$0")/../tf-attend-infer-repeat_amd"
for m in $modes; do for p in $precs; do for s in "$@"; do
  python training.py -r /tmp/sweep_${m}_${p}_$s -o 1 --iterations $iters --print-every 0 --precision $p --seed $s --backward $m > /tmp/sweep.log 2>&1
  python - <<PY >> "../$out"
import json, os
rows=[json.loads(l) for l in open("/tmp/sweep_${m}_${p}_$s/summary/scalars.jsonl")]
first=next((r["step"] for r in rows if r["accuracy"]>=0.98), None)
print(json.dumps({"backward":"$m","precision":"$p","seed":$s,"glyph_zoom":os.environ.get("AIR_GLYPH_ZOOM","1.5"),"iterations":$iters,"final_accuracy":rows[-1]["accuracy"],"best_accuracy":max(r["accuracy"] for r in rows),"acc_at":{str(r["step"]):r["accuracy"] for r in rows if r["step"]%10000==0},"first_step_at_98pct":first,"wall_s":rows[-1]["wall_s"]}))
PY
done; done; done
