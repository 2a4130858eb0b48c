#!/bin/bash
# Round-6 closing run on the GPU box: profile rounds on the final sources (default / carried / fp32 / stress), the bench lines
# WITH the fresh kernel_profile.json in place, smoke(), the full `pytest -m gpu`, the fp32 full-length training run.
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
out=$root/gpurun_out/r06_final; mkdir -p $out
bash tools/profile_round.sh r06 > $out/profile_round.log 2>&1
bash tools/profile_round.sh r06_carried --backward reference_carried >> $out/profile_round.log 2>&1
bash tools/profile_round.sh r06_fp32 --precision fp32 >> $out/profile_round.log 2>&1
bash tools/profile_round.sh r06_stress --workload "configs[3]" >> $out/profile_round.log 2>&1
cp gpurun_out/prof_r06/kernel_profile.json profiles/kernel_profile.json
python bench.py > $out/bench_default.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2>> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_b.json 2>> $out/bench.err
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1
for p in bf16 fp32; do (cd tf-attend-infer-repeat_amd && python training.py -r /tmp/full_$p -o 1 --print-every 0 --precision $p --seed 0 > $out/full_$p.log 2>&1; cp /tmp/full_$p/summary/scalars.jsonl $out/full_${p}_scalars.jsonl; tail -1 $out/full_$p.log); done; python tools/exp/queue_cost.py 50 2>&1 | grep "ms per step\|us" > $out/queue_cost.txt
tail -3 $out/smoke.log; tail -4 $out/pytest_gpu.log; tail -1 $out/full_fp32.log; cut -c1-200 $out/bench_steps20.json; head -8 gpurun_out/prof_r06/kernel_stats.txt
