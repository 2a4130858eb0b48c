#!/bin/bash
# A/B of tuning switches on one box: each line = ms/step of `bench.py` (200 steps, 20 per graph replay) under an env setting
run() { echo -n "$1: "; env $1 python bench.py --no-extras --no-roofline --no-cpu-baseline --steps ${STEPS:-400} --warmup 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for cfg in "$@"; do run "$cfg"; done
