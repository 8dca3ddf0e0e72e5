"""Diagnostic: time of TrainWorkspace.prepare (the [N][F+C] row matrix) on a 2048^2 x 8 tile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img_d = ops.to_device_u16(synthetic_tile(0, 8, 2048, 2048), dev)
msb_d, mx = ops.split_bits(img_d, 5)
for cfg, F in ((FeatCfg(), 200), (FeatCfg(use_coordinates=True, embedding=True), 250)):
    geom = ops.FeatureGeometry(8, 2048, 2048, 5, 2, mx, cfg, dev)
    net = ops.make_net(F, 64, 8, 2)
    ws = ops.TrainWorkspace(geom, net, 8192, dev)
    ws.prepare(img_d, msb_d, ops._lib.PATH_MFMA)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ws.prepare(img_d, msb_d, ops._lib.PATH_MFMA)
    torch.cuda.synchronize(); print(f"F={F}: prepare {(time.perf_counter()-t)/5*1e3:.2f} ms")
