import os, sys, time
sys.path.insert(0, "lbdrn-msic_amd")
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(1000, 8, 2048, 2048), dev)
torch.manual_seed(1)
fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 1, cfg=FeatCfg(activation="relu"))
torch.cuda.synchronize()
print("done")
