#!/bin/bash
# Same nine positional arguments as the reference's run.sh (DEVICE_ID D BC NL LR BS EPOCH SR OUTPUT_DIR);
# the sweep itself is sweep.py.  NGPU=8 ./run.sh ... deals the (image, K) points over eight GPUs;
# PER_GPU=2 (default) starts two processes per GPU: one fit is a chain of short dependent kernels (a training step on
# half of the CUs, then a small reduce launch), and a second independent chain on the same GPU fills the other half and
# the gaps (round 3, 8 x 2048^2 tiles: 128 ms per tile for one chain alone, 87 with two).  A process cannot see the
# other one's fits, so with more than one per GPU it is told (LBDRN_DEVICE_SHARED=1: codec.device_shared): its training
# steps then stay on the half-chip launch (no LBDRN_TRAIN_ALONE hint: the every-CU launch of a lone fit would take turns
# with the neighbour's), its evaluation passes stay in the chain instead of taking the half of the chip the neighbour is
# using, and fit_many sizes its fits in flight against half of the free memory.
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# DEVICE_ID (first argument) selects the GPU like the reference's CUDA_VISIBLE_DEVICES=$DEVICE_ID (ref run.sh:37-39),
# at shell level, before python starts; with NGPU set the launcher owns the placement and DEVICE_ID is ignored.
if [ -z "${NGPU}" ] && [[ "$1" =~ ^[0-9]+$ ]]; then
    export HIP_VISIBLE_DEVICES="$1"
fi
NPROC=$(( ${NGPU:-1} * ${PER_GPU:-2} ))
if [ "${PER_GPU:-2}" -gt 1 ]; then export LBDRN_DEVICE_SHARED="${LBDRN_DEVICE_SHARED:-1}" LBDRN_OVERLAP_EVAL="${LBDRN_OVERLAP_EVAL:-0}"; fi
if [ "${NPROC}" -gt 1 ]; then
    exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "${NPROC}" --master-addr 127.0.0.1 \
        --master-port "${MASTER_PORT:-29531}" "${HERE}/sweep.py" "$@"
fi
exec python "${HERE}/sweep.py" "$@"
