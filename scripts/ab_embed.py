"""A/B aid: wall time per tile of codec.fit_many for BASELINE configs[4] (USE_COORDINATES + EMBEDDING, F = 250)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(4)] * 2
cfg = FeatCfg(use_coordinates=True, embedding=True)
out = []
for infl in [int(x) for x in os.environ.get('AB_INFLIGHT', '1,4').split(',')]:
    codec.fit_many(tiles[:max(2, infl)], 5, 2, 64, 2, 1e-3, 8192, 10, cfg=cfg, seed=19920517, in_flight=infl)
    torch.cuda.synchronize(); t = time.perf_counter()
    fits = codec.fit_many(tiles, 5, 2, 64, 2, 1e-3, 8192, 10, cfg=cfg, seed=19920517, in_flight=infl)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / len(tiles)
    out.append(f"in_flight={infl}: {dt*1e3:.2f} ms/tile")
print("embed", " | ".join(out))
