"""Diagnostic: one training epoch of the bc = 256 network with the stamped build (LBDRN_HIP_LIB=.../liblbdrn_hip_stamps.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(0, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 2048), dev)
torch.manual_seed(19920517)
for _ in range(3):
    fit = codec.fit_device(img, 5, 2, 256, 2, 1e-3, 8192, 1)
torch.cuda.synchronize()
print("done")
