"""Host side of the MI355X LBDRN hot path: ctypes binding of liblbdrn_hip.so plus the per-image
fit / apply loops that sit where the reference's encode.train() / decode.test() sit.

There is no CPU implementation in this package: every compute entry point needs the HIP library
and a gfx950 device and raises otherwise.
"""
from ._lib import LbdrnError, lib, lib_path  # noqa: F401
