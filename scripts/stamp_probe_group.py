"""Diagnostic: one training epoch of a GROUP of fits (argv[1], default 2) with the stamped build of the library
(LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_stamps.so): per-phase cycles of fit 0's waves while the other fits'
workgroups hold the rest of the chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
from lbdrn_hip.features import FeatCfg
embed = len(sys.argv) > 2 and sys.argv[2] == "embed"
bands = 4 if len(sys.argv) > 2 and sys.argv[2] == "bands4" else 8
cfg = FeatCfg(use_coordinates=embed, embedding=embed)
dev = torch.device("cuda:0")
imgs = [ops.to_device_u16(synthetic_tile(i, bands, 2048, 2048), dev) for i in range(n)]
for _ in range(4):   # (the stamped build prints one line per epoch call; collect_inkernel.py drops the first)
    if n == 1:
        torch.manual_seed(19920517); codec.fit_device(imgs[0], 5, 2, 64, 2, 1e-3, 8192, 1, cfg=cfg)
    else:
        codec.fit_group(imgs, 5, 2, 64, 2, 1e-3, 8192, 1, seed=19920517, cfg=cfg)
torch.cuda.synchronize()
print("done")
