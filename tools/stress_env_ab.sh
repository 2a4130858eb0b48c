#!/bin/bash
# ms/step of the stress configuration (configs[3]) under env settings, one line each
run() { echo -n "$1: "; env $1 python bench.py --workload "configs[3]" --no-cpu-baseline --no-extras --steps ${STEPS:-80} --warmup 8 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for cfg in "$@"; do run "$cfg"; done
