"""Timing of one whole-image evaluation / decode pass at bc = 256 (8 x 2048^2): k_apply_wide against the generic per-layer kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import numpy as np, torch
from lbdrn_hip import ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
bc = int(sys.argv[1]) if len(sys.argv) > 1 else 256
img = synthetic_tile(0, 8, 2048, 2048)
img_d = ops.to_device_u16(img, dev)
msb_d, mx = ops.split_bits(img_d, 5)
geom = ops.FeatureGeometry(8, 2048, 2048, 5, 2, mx, FeatCfg(), dev)
net = ops.make_net(200, bc, 8, 2)
rng = np.random.default_rng(1)
p = torch.from_numpy((rng.standard_normal(ops.param_count(net)) * 0.02).astype(np.float32)).to(dev)
for name, path, bg in (("mfma", ops._lib.PATH_MFMA, False), ("mfma-background", ops._lib.PATH_MFMA, True), ("generic", ops._lib.PATH_GENERIC, False)):
    ws = ops.ApplyWorkspace(geom, net, dev)
    for _ in range(2):
        s = ops.eval_sse(geom, net, img_d, msb_d, p, path, ws, background=bg)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        s = ops.eval_sse(geom, net, img_d, msb_d, p, path, ws, background=bg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"bc={bc} eval {name}: {dt*1e3:.2f} ms  sse={s.item():.6f}")
