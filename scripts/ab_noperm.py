"""Experiment: how much of a tile's wall time is the GPU permutation generation?  Same as ab_inflight.py with
ops.randperm replaced by a cached result (wrong minibatch orders, same work for every other kernel)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(4)] * 2
args = (5, 2, 64, 2, 1e-3, 8192, 10)
real = ops.randperm
cache = {}
def fake(seeds, n, device):
    k = (len(seeds), n)
    if k not in cache:
        cache[k] = real(seeds, n, device)
    return cache[k]
for name, fn in (("real", real), ("cached", fake), ("real", real), ("cached", fake)):
    ops.randperm = fn
    out = []
    for infl in (1, 4):
        codec.fit_many(tiles[:max(2, infl)], *args, seed=19920517, in_flight=infl)
        torch.cuda.synchronize(); t = time.perf_counter()
        codec.fit_many(tiles, *args, seed=19920517, in_flight=infl)
        torch.cuda.synchronize(); out.append(f"in_flight={infl}: {(time.perf_counter() - t) / len(tiles) * 1e3:.2f} ms/tile")
    print(name, " | ".join(out))
