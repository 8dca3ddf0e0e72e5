"""Replay tests/test_gpu_fuzz.py's apply cases and print every mismatch with the oracle's verdict."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "lbdrn-msic_amd"), os.path.join(ROOT, "tests")]
import oracle as O
from lbdrn_hip import ops
import test_gpu_fuzz as T

dev = torch.device("cuda:0")
rng = np.random.default_rng(20240101)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 480
bad = 0
for it in range(n):
    C, H, W, K, D, bc, nl, cfg, img = T._random_case(rng, train=False)
    F = cfg.feature_dim(C, D)
    if F > 400:
        continue
    msb = img >> K
    mx = int(msb.max())
    if mx == 0:
        continue
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
    net = ops.make_net(F, bc, C, nl)
    pn = T._params(rng, F, bc, C, nl, 2.5)
    p = torch.from_numpy(pn).to(dev)
    msb_d = ops.to_device_u16(msb, dev)
    try:
        a, ya = ops.decode_fused(geom, net, msb_d, p, want_y=True, path=T.MFMA)
    except ops._lib.LbdrnError as e:
        continue
    b, yb = ops.decode_fused(geom, net, msb_d, p, want_y=True, path=T.GEN)
    if not torch.equal(ya.view(torch.int32), yb.view(torch.int32)):
        bad += 1
        ocfg = O.FeatCfg(cfg.use_coordinates, cfg.embedding, 1.4, 12, cfg.use_colors, cfg.relative)
        feats = O.features(msb, D, ocfg, mx)
        yo = O.forward(pn, F, bc, C, nl, feats)
        ya_n, yb_n = ya.cpu().numpy().reshape(yo.shape), yb.cpu().numpy().reshape(yo.shape)
        da, db = np.abs(ya_n - yo), np.abs(yb_n - yo)
        rows = np.nonzero((ya_n != yb_n).any(axis=1))[0]
        print(f"it={it} C={C} H={H} W={W} K={K} D={D} bc={bc} nl={nl} F={F} {vars(cfg)}")
        print(f"   mfma-vs-oracle max {da.max():.3e}  generic-vs-oracle max {db.max():.3e}  bad rows {len(rows)}/{H*W}"
              f" first {rows[:12]} cols {np.nonzero((ya_n != yb_n).any(axis=0))[0][:20]}")
print("mismatching cases:", bad)
