#!/bin/bash
# A/B of tuning switches on one GPU box: ms/step of bench.py under each env setting ("-" = defaults), optionally --workload
# usage: tools/ab.sh [bench args --] "ENV1=a ENV2=b" "-" ...
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do case "$1" in --*) args+=("$1" "$2"); shift 2;; *) break;; esac; done
[ "$1" == "--" ] && shift
for cfg in "$@"; do
  e=$cfg; [ "$cfg" == "-" ] && e="AIR_DUMMY=0"
  echo -n "$cfg: "
  env $e python bench.py --no-extras --no-roofline --no-cpu-baseline --steps ${STEPS:-400} --warmup 40 "${args[@]}" 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])"
done
