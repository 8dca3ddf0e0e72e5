#!/bin/bash
# Same nine positional arguments as the reference's run.sh (DEVICE_ID D BC NL LR BS EPOCH SR OUTPUT_DIR);
# the sweep itself is sweep.py.  NGPU=8 ./run.sh ... deals the (image, K) points over eight GPUs.
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
if [ "${NGPU:-1}" -gt 1 ]; then
    exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "${NGPU}" --master-addr 127.0.0.1 \
        --master-port "${MASTER_PORT:-29531}" "${HERE}/sweep.py" "$@"
fi
HIP_VISIBLE_DEVICES="$1" exec python "${HERE}/sweep.py" "$@"
