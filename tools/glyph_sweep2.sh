#!/bin/bash
# second glyph-rendering sweep (GPU box): tools/glyph_sweep2.sh out.jsonl iters seeds...
out=$1; iters=$2; shift 2
cd "$(dirname "$0")/../tf-attend-infer-repeat_amd"
run() { # name zoom order contrast
  for s in $SEEDS; do
    AIR_GLYPH_ZOOM=$2 AIR_GLYPH_ORDER=$3 AIR_GLYPH_CONTRAST=$4 python training.py -r /tmp/g2_$1_$s -o 1 --iterations $iters --print-every 0 --precision fp32 --seed $s > /tmp/g2.log 2>&1
    python - <<PY >> "../$out"
import json
rows=[json.loads(l) for l in open("/tmp/g2_$1_$s/summary/scalars.jsonl")]
r=rows[-1]
first=next((q["step"] for q in rows if q["accuracy"]>=0.98), None)
print(json.dumps({"variant":"$1","zoom":$2,"order":$3,"contrast":"$4","seed":$s,"iterations":$iters,"final_accuracy":round(r["accuracy"],3),"acc012":[round(r["digit_acc_%d_dig"%i],2) for i in range(3)],"steps012":[round(r["steps_%d_dig"%i],2) for i in range(3)],"first_step_at_98pct":first}))
PY
  done
}
SEEDS="$@"
run F 1.75 3 0.25,0.65
run G 2.0 3 0.35,0.55
run H 1.5 3 0.25,0.65
