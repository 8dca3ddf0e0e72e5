import os, sys, itertools
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "lbdrn-msic_amd"), os.path.join(ROOT, "tests")]
from lbdrn_hip import ops
from lbdrn_hip.features import FeatCfg
import test_gpu_fuzz as T
dev = torch.device("cuda:0")
for bc, nl, C, D, rel, gain in itertools.product([32, 64, 128], [1, 2, 3], [3, 15], [0, 1], [False, True], [1.0, 2.5]):
    rng = np.random.default_rng(1)
    H, W, K = 14, 27, 6
    cfg = FeatCfg(False, False, 1.4, 12, True, rel)
    img = rng.integers(0, 65536, (C, H, W)).astype(np.uint16)
    msb = img >> K; mx = int(msb.max())
    F = cfg.feature_dim(C, D)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
    net = ops.make_net(F, bc, C, nl)
    p = torch.from_numpy(T._params(rng, F, bc, C, nl, gain)).to(dev)
    msb_d = ops.to_device_u16(msb, dev)
    try:
        a, ya = ops.decode_fused(geom, net, msb_d, p, want_y=True, path=T.MFMA)
    except ops._lib.LbdrnError as e:
        print("skip", bc, nl, C, D, rel, str(e)[:60]); continue
    b, yb = ops.decode_fused(geom, net, msb_d, p, want_y=True, path=T.GEN)
    d = (ya - yb).abs().max().item()
    neq = (ya.view(torch.int32) != yb.view(torch.int32)).sum().item()
    if neq:
        print(f"MISMATCH bc={bc} nl={nl} C={C} D={D} rel={rel} gain={gain} F={F}: maxdiff {d:.3e} n={neq}/{ya.numel()}")
print("done")
