"""How many steps of a lone fit's epoch go out on the half-chip launch beside the previous epoch's background evaluation pass:
the model's guess (codec.background_steps: two measured constants) against what the fits' own events say from the second fit of
a shape on (codec._calibrated_head).  Round 6, one MI355X: 8 bands 137 guessed / 128-135 measured, 4 bands 115 / 113-114.
    python scripts/head_calibration_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
for bands in (8, 4):
    img = ops.to_device_u16(synthetic_tile(0, bands, 2048, 2048), dev)
    for rep in range(6):
        torch.manual_seed(19920517)
        f = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 3)
        torch.cuda.synchronize()
        guess = codec.background_steps(512, f.net, 2048 * 2048, 8192)
        print(bands, rep, "guess", guess, "measured", codec.head_calibration().get(codec._head_key(dev, f.net, 2048 * 2048, 8192)), flush=True)
