#!/bin/bash
# A/B of an environment switch on the headline bench: scripts/ab_env.sh VAR [rounds] -- prints ms per tile and the training
# launch's own duration (roofline.kernel_us) with VAR unset / set to 1, interleaved.
VAR=$1; N=${2:-2}
mkdir -p gpurun_out/ab
for r in $(seq 1 $N); do for v in 0 1; do
  if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
  python bench.py --steps 16 --warmup 4 --repeats 2 --no-cpu-baseline --no-other-configs > gpurun_out/ab/${VAR}_${v}_${r}.json 2>>gpurun_out/ab/err.log || exit 1
  python -c "
import json
d=json.loads(open('gpurun_out/ab/${VAR}_${v}_${r}.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$VAR =', $v, d['ms_per_step_all_repeats'], 'kernel_us', r['kernel_us'], 'marginal', r.get('marginal_us'), 'step pair', r.get('train_step_pair_us'), 'lone tile', d.get('single_tile_ms'))"
done; done
