#!/bin/bash
# bc = 256: tiles in flight and evaluation-overlap policy, ms per tile (one bench.py run each, same box)
for cfg in "2 0" "3 0" "4 0" "2 1" "3 1"; do set -- $cfg
  if [ "$2" = "1" ]; then export LBDRN_OVERLAP_EVAL=1; else unset LBDRN_OVERLAP_EVAL; fi
  python3 bench.py -bc 256 --in-flight $1 --steps $((2*$1)) --warmup $1 --repeats 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in flight $1 overlap_eval=$2:', d['ms_per_step'], 'ms per tile; alone', d['single_tile_ms'])"
done
