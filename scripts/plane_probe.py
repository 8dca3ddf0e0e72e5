"""Diagnostic: the LBB2 plane codec on the full-size MSB planes (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import numpy as np, torch
from lbdrn_hip import ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img = synthetic_tile(0, 8, 2048, 2048)
for K in (5, 9):
    x_d = ops.to_device_u16((img >> K).astype(np.uint16), dev)
    for _ in range(3):
        body = ops.plane_encode(x_d)
        back = ops.plane_decode(body, 8, 2048, 2048, dev)
    torch.cuda.synchronize()
    assert torch.equal(back, x_d)
    print(f"K={K}: {8 * len(body) / x_d.numel():.3f} bpsp")
