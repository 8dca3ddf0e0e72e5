"""Diagnostic: one training epoch on a synthetic tile with the stamped build of the library
(LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_stamps.so); the library prints mean cycles per phase."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch  # noqa: E402
from lbdrn_hip import codec, ops  # noqa: E402
from lbdrn_hip.synth import synthetic_tile  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(0, 8, side, side), dev)
torch.manual_seed(19920517)
for _ in range(2):
    fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 1)
torch.cuda.synchronize()
print("done")
