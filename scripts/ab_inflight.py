"""A/B aid: wall time per tile of codec.fit_many over 6 tiles with 1 and 2 fits in flight, for the train
library named by LBDRN_HIP_LIB (e.g. a variant built with csrc/build.py --variant tile -DLBDRN_EXP_TILE_KERNEL: every shape on k_train_mfma)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # like bench.py: one hardware queue per fit in flight
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(4)] * int(os.environ.get('AB_REPEAT', '2'))
args = (5, 2, int(os.environ.get('AB_BC', '64')), 2, 1e-3, 8192, 10)
out = []
for infl in [int(x) for x in os.environ.get('AB_INFLIGHT', '1,2').split(',')]:
    codec.fit_many(tiles[:max(2, infl)], *args, seed=19920517, in_flight=infl)   # every in-flight stream warm
    torch.cuda.synchronize(); t = time.perf_counter()
    fits = codec.fit_many(tiles, *args, seed=19920517, in_flight=infl)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / len(tiles)
    out.append(f"in_flight={infl}: {dt*1e3:.2f} ms/tile")
print(os.path.basename(os.environ.get("LBDRN_HIP_LIB", "liblbdrn_hip.so")), " | ".join(out), "mse", [round(float(f.mse_log[:, 0].min()), 6) for f in fits[:3]])
