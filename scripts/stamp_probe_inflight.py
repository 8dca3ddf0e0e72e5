"""Diagnostic: N fits in flight (argv[1], default 4; fit_many, one stream each) with the stamped build of the library
(LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_stamps.so): every epoch of every fit prints the shader clock and the
per-phase cycles of its last training launch, taken while the other fits hold the rest of the chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bands = int(sys.argv[3]) if len(sys.argv) > 3 else 8
nl = int(sys.argv[4]) if len(sys.argv) > 4 else 2
group = int(sys.argv[5]) if len(sys.argv) > 5 else 1     # fits per launch (codec.fit_many: 2 = pairs)
dev = torch.device("cuda:0")
imgs = [ops.to_device_u16(synthetic_tile(i, bands, 2048, 2048), dev) for i in range(2 * n)]
codec.fit_many(imgs, 5, 2, 64, nl, 1e-3, 8192, epochs, seed=19920517, in_flight=n, group=group)
torch.cuda.synchronize()
print("done")
