#!/bin/bash
# per-kernel times (HIP events, bench.py's own table) under an env setting: tools/kt.sh "ENV=.." [bench args]
e=$1; shift; [ "$e" == "-" ] && e="AIR_DUMMY=0"
env $e python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 40 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'sum_kernel_us', d['step_roofline']['sum_kernel_us'])
for k,v in d['kernels'].items(): print('%7.2f %d  %-55s %s' % (v['us_per_step'], v['launches'], k, ','.join(v['ops'])))
"
