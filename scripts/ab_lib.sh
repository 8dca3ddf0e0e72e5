#!/bin/bash
# A/B of library variants inside one gpurun call: scripts/ab_lib.sh NAME1 NAME2 ...  ("base" = the shipped library)
for v in "$@"; do
  if [ "$v" = base ]; then unset LBDRN_HIP_LIB; else export LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_$v.so; fi
  AB_INFLIGHT=${AB_INFLIGHT:-1,2,3} python scripts/ab_inflight.py 2>/dev/null | sed "s/^/[$v] /"
done
