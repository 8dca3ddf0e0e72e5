#!/bin/bash
# bench.py over (fits per launch : fits in flight) combinations; usage: scripts/ab_groups.sh "1:4 4:4 4:8 ..."
mkdir -p gpurun_out/grp
for cfg in $1; do
  g=${cfg%%:*}; f=${cfg##*:}
  LBDRN_FIT_GROUP=$g python bench.py --steps 24 --warmup 4 --in-flight $f --no-cpu-baseline --no-other-configs > gpurun_out/grp/g${g}_f$f.json 2>>gpurun_out/grp/err.log || exit 1
  python -c "
import json
d=json.loads(open('gpurun_out/grp/g${g}_f$f.json').read().strip().splitlines()[-1])
print('group $g in flight $f:', d['value'], 'Mpx/s', d['ms_per_step_all_repeats'])"
done
