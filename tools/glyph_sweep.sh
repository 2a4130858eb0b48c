#!/bin/bash
# which stand-in glyph rendering trains most reliably? (GPU box)  usage: tools/glyph_sweep.sh out.jsonl iters seeds...
out=$1; iters=$2; shift 2
cd "$(dirname "$0")/../tf-attend-infer-repeat_amd"
run() { # name zoom order contrast
  for s in $SEEDS; do
    AIR_GLYPH_ZOOM=$2 AIR_GLYPH_ORDER=$3 AIR_GLYPH_CONTRAST=$4 python training.py -r /tmp/g_$1_$s -o 1 --iterations $iters --print-every 0 --precision fp32 --seed $s > /tmp/g.log 2>&1
    python - <<PY >> "../$out"
import json
rows=[json.loads(l) for l in open("/tmp/g_$1_$s/summary/scalars.jsonl")]
r=rows[-1]
first=next((q["step"] for q in rows if q["accuracy"]>=0.98), None)
print(json.dumps({"variant":"$1","zoom":$2,"order":$3,"contrast":"$4","seed":$s,"iterations":$iters,"final_accuracy":round(r["accuracy"],3),"acc012":[round(r["digit_acc_%d_dig"%i],2) for i in range(3)],"steps012":[round(r["steps_%d_dig"%i],2) for i in range(3)],"first_step_at_98pct":first}))
PY
  done
}
SEEDS="$@"
run A 2.0 1 0,1
run B 2.5 3 0.25,0.65
run C 2.5 1 0,1
run D 2.0 3 0.25,0.65
run E 3.0 3 0.25,0.65
