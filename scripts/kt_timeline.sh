#!/bin/bash
# rocprofv3 --kernel-trace of scripts/prof_many.py + concurrency statistics of its timed part
# usage: scripts/kt_timeline.sh TAG IN_FLIGHT GROUP [TILES]
TAG=$1; IF=$2; GR=$3; N=${4:-8}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $OUT/kt -o run -- python3 scripts/prof_many.py $IF $GR $N > $OUT/run_if${IF}_g$GR.txt 2> $OUT/kt.err
cat $OUT/run_if${IF}_g$GR.txt
DB=$(find $OUT/kt -name "*.db" | head -1)
python scripts/timeline_stats.py $DB 0.55 | tee $OUT/timeline_if${IF}_g$GR.txt
rm -rf $OUT/kt
