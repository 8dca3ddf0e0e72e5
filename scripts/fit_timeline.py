"""Diagnostic: where the wall time of one fit goes (host vs device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops, sampler
from lbdrn_hip.synth import synthetic_tile

dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(0, 8, 2048, 2048), dev)
for it in range(3):
    torch.manual_seed(19920517)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 10)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"fit: host-side return after {1e3*(t1-t0):.1f} ms, device done after {1e3*(t2-t0):.1f} ms")
