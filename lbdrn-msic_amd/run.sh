#!/bin/bash
# Same nine positional arguments as the reference's run.sh (DEVICE_ID D BC NL LR BS EPOCH SR OUTPUT_DIR);
# the sweep itself is sweep.py.  NGPU=8 ./run.sh ... deals the (image, K) points over eight GPUs;
# PER_GPU=2 (default) starts two processes per GPU: one fit is a chain of short dependent kernels, and a second
# independent chain on the same GPU fills its gaps (37 instead of 27 Mpixel/s per GPU).
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# DEVICE_ID (first argument) selects the GPU like the reference's CUDA_VISIBLE_DEVICES=$DEVICE_ID (ref run.sh:37-39),
# at shell level, before python starts; with NGPU set the launcher owns the placement and DEVICE_ID is ignored.
if [ -z "${NGPU}" ] && [[ "$1" =~ ^[0-9]+$ ]]; then
    export HIP_VISIBLE_DEVICES="$1"
fi
NPROC=$(( ${NGPU:-1} * ${PER_GPU:-2} ))
if [ "${NPROC}" -gt 1 ]; then
    exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "${NPROC}" --master-addr 127.0.0.1 \
        --master-port "${MASTER_PORT:-29531}" "${HERE}/sweep.py" "$@"
fi
exec python "${HERE}/sweep.py" "$@"
