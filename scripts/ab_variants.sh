#!/bin/bash
# A/B of library builds inside one gpurun call: scripts/ab_variants.sh TAG NAME...   ("base" = the shipped library)
#   per build: scripts/ab_inflight.py at AB_INFLIGHT (default 1,4); a build named *_st (in-kernel stamps) runs
#   scripts/stamp_probe.py instead and prints the per-phase cycles
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for v in "$@"; do
  if [ "$v" = base ]; then unset LBDRN_HIP_LIB; else export LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_$v.so; fi
  case $v in
    *_st|stamps) python scripts/stamp_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$v] /" | tee -a gpurun_out/$TAG/stamps.txt ;;
    *) AB_INFLIGHT=${AB_INFLIGHT:-1,4} python scripts/ab_inflight.py 2>/dev/null | sed "s/^/[$v] /" | tee -a gpurun_out/$TAG/ab.txt ;;
  esac
done
