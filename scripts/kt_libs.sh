#!/bin/bash
# rocprofv3 --kernel-trace of scripts/prof_many.py per library build: scripts/kt_libs.sh TAG IN_FLIGHT NAME...
TAG=$1; IF=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset LBDRN_HIP_LIB; else export LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_$v.so; fi
  rocprofv3 --kernel-trace -d $OUT/kt_$v -o run -- python3 scripts/prof_many.py $IF 1 4 > $OUT/run_$v.txt 2> $OUT/kt_$v.err
  DB=$(find $OUT/kt_$v -name "*.db" | head -1)
  echo "== $v: $(cat $OUT/run_$v.txt)"
  python scripts/rocprof_kernel_stats.py $DB $OUT/kernel_stats_$v.csv | head -4
  rm -rf $OUT/kt_$v
done
