cd tf-attend-infer-repeat_amd
for s in 0 1; do
python training.py -r /tmp/clut_$s -o 1 --iterations 80000 --print-every 0 --precision fp32 --seed $s --bg-path ../tests/golden/backgrounds.npz:pattern1 --bg-max-intensity 0.3 > /tmp/clut_$s.log 2>&1
python - <<PY
import json
rows=[json.loads(l) for l in open("/tmp/clut_$s/summary/scalars.jsonl")]
print("clutter pattern1 0.3 seed $s", {r["step"]: round(r["accuracy"],3) for r in rows if r["step"]%10000==0}, "final", rows[-1]["accuracy"])
PY
done
