#!/bin/bash
# A/B of library builds, interleaved in-process runs are impossible (one library per process): alternate processes.
# usage: scripts/ab_libs.sh "CONFIGS" NAME NAME ...   (base = shipped library), each NAME run twice, alternating
CFG=$1; shift
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset LBDRN_HIP_LIB; else export LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_$v.so; fi
  AB_TILES=${AB_TILES:-8} AB_REPEAT=${AB_REPEAT:-2} AB_CONFIGS="$CFG" python scripts/ab_group.py 2>/dev/null | sed "s/^/[$v] /"
done; done
