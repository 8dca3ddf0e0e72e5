"""Host side of the MI355X LBDRN hot path: ctypes binding of liblbdrn_hip.so plus the per-image
fit / apply loops that sit where the reference's encode.train() / decode.test() sit.

There is no CPU implementation in this package: every compute entry point needs the HIP library
and a gfx950 device and raises otherwise.
"""
from ._lib import LbdrnError, lib, lib_path  # noqa: F401

import os as _os

# Fits in flight (codec.fit_many), the permutation side stream and torch's own streams should each own a
# hardware queue; the runtime's default of four per process can make two of them share one and serialise.
# Read when the HIP runtime initialises (the first device call), so setting it at import time is early enough.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
