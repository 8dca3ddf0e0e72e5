import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "lbdrn-msic_amd"))
import numpy as np, torch
from lbdrn_hip import ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img = synthetic_tile(0, 8, 2048, 2048)
img_d = ops.to_device_u16(img, dev)
msb_d, mx = ops.split_bits(img_d, 5)
for bc, embed in ((64, False), (256, False), (64, True)):
    cfg = FeatCfg(embed, embed)
    geom = ops.FeatureGeometry(8, 2048, 2048, 5, 2, mx, cfg, dev)
    net = ops.make_net(geom.F, bc, 8, 2)
    torch.manual_seed(0)
    p = (torch.rand(ops.param_count(net), device=dev) - 0.5) * 0.05
    ws = ops.ApplyWorkspace(geom, net, dev)
    for fast in (False, True):
        for _ in range(3): ops.eval_sse(geom, net, img_d, msb_d, p, ws=ws, fast=fast)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): s = ops.eval_sse(geom, net, img_d, msb_d, p, ws=ws, fast=fast)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(f"bc={bc} embed={embed} fast={fast}: eval pass {dt*1e3:.3f} ms  sse={float(s.item()):.9f}")
