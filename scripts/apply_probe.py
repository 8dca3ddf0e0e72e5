"""Diagnostic: time of the fused apply kernel (eval mode) on a 2048^2x8 tile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import numpy as np, torch
from lbdrn_hip import ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img = synthetic_tile(0, 8, 2048, 2048)
img_d = ops.to_device_u16(img, dev)
msb_d, mx = ops.split_bits(img_d, 5)
geom = ops.FeatureGeometry(8, 2048, 2048, 5, 2, mx, FeatCfg(), dev)
net = ops.make_net(200, 64, 8, 2)
p = (torch.rand(17544, device=dev) - 0.5) * 0.05
ws = ops.ApplyWorkspace(geom, net, dev)
for _ in range(3): ops.eval_sse(geom, net, img_d, msb_d, p, ws=ws)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): s = ops.eval_sse(geom, net, img_d, msb_d, p, ws=ws)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"eval pass {dt*1e3:.3f} ms -> {34816*2048*2048/dt/1e12:.1f} TFLOP/s  sse={float(s.item()):.6f}")
