#!/bin/bash
# kernel-trace A/B inside one gpurun call: scripts/kt_ab.sh TAG KERNEL...   (KERNEL = stream | wave | tile)
# per kernel choice: rocprofv3 --kernel-trace --stats of bench.py with one and with four fits in flight
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for K in "$@"; do
  export LBDRN_TRAIN_KERNEL=$K
  for IF in 1 4; do
    ST=$([ $IF = 1 ] && echo 2 || echo 8)
    rocprofv3 --kernel-trace --stats -d $OUT/kt_${K}_$IF -o run -- python3 bench.py --no-cpu-baseline --in-flight $IF --steps $ST --warmup 1 > $OUT/bench_${K}_$IF.json 2> $OUT/kt_${K}_$IF.err
    DB=$(find $OUT/kt_${K}_$IF -name "*.db" | head -1)
    python scripts/rocprof_kernel_stats.py $DB $OUT/kernel_stats_${K}_$IF.csv > $OUT/kernel_stats_${K}_$IF.txt 2>&1
    rm -rf $OUT/kt_${K}_$IF
    echo "== $K in_flight=$IF"; head -6 $OUT/kernel_stats_${K}_$IF.txt; python -c "import json,sys; d=json.loads(open('$OUT/bench_${K}_$IF.json').read().strip().splitlines()[-1]); print('ms/tile', d['ms_per_step'], 'single', d['single_tile_ms'])"
  done
done
