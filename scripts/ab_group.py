"""A/B inside ONE process, configurations interleaved and repeated (boxes and runs differ by ~10 %): wall time per
tile of codec.fit_many over 8 tiles for (group, in_flight) pairs given as AB_CONFIGS="1:4,2:4,2:6" (default)."""
import os, sys, time, statistics
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
ntiles = int(os.environ.get("AB_TILES", "12"))
tiles = [ops.to_device_u16(synthetic_tile(i % 4, 8, 2048, 2048), dev) for i in range(ntiles)]
args = (5, 2, int(os.environ.get("AB_BC", "64")), 2, 1e-3, 8192, 10)
cfgs = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("AB_CONFIGS", "1:4,2:4,2:6").split(",")]
for g, infl in cfgs:   # warm every stream / allocator pool
    codec.fit_many(tiles[:infl], *args, seed=19920517, in_flight=infl, group=g)
torch.cuda.synchronize()
res = {c: [] for c in cfgs}
for rep in range(int(os.environ.get("AB_REPEAT", "3"))):
    for g, infl in cfgs:
        torch.cuda.synchronize(); t = time.perf_counter()
        codec.fit_many(tiles, *args, seed=19920517, in_flight=infl, group=g)
        torch.cuda.synchronize(); res[(g, infl)].append((time.perf_counter() - t) / len(tiles) * 1e3)
for c in cfgs:
    print(f"group={c[0]} in_flight={c[1]}: median {statistics.median(res[c]):.2f} ms/tile  samples {[round(x, 2) for x in res[c]]}")
