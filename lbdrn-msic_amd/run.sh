#!/bin/bash
# Same nine positional arguments as the reference's run.sh (DEVICE_ID D BC NL LR BS EPOCH SR OUTPUT_DIR);
# the sweep itself is sweep.py.  NGPU=8 ./run.sh ... deals the (image, K) points over eight GPUs;
# PER_GPU=2 (default) starts two processes per GPU: one fit is a chain of short dependent kernels, and a second
# independent chain on the same GPU fills its gaps (37 instead of 27 Mpixel/s per GPU).
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
NPROC=$(( ${NGPU:-1} * ${PER_GPU:-2} ))
if [ "${NPROC}" -gt 1 ]; then
    exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "${NPROC}" --master-addr 127.0.0.1 \
        --master-port "${MASTER_PORT:-29531}" "${HERE}/sweep.py" "$@"
fi
exec python "${HERE}/sweep.py" "$@"
