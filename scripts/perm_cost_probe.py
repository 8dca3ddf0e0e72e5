"""Diagnostic (not a product path): what the permutation pipeline costs a tile in the four-in-flight mix.  Every fit of the
bench seeds alike, so its ten permutations are the same tensors for every tile of one size; this script times
codec.fit_many over 16 tiles as shipped (every fit computes its permutations on the GPU, what the reference's
DataLoader does per image) and with ops.randperm memoised IN THIS SCRIPT ONLY -- the difference is the pipeline's cost
(k_mt19937_raw, buckets, links, chase: ~6 ms of kernel time per tile on the side stream)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(8)] * 2
real = ops.randperm
cache = {}
def memo(seeds, n, device):
    key = (tuple(int(s) for s in seeds), int(n))
    if key not in cache:
        cache[key] = real(seeds, n, device)
    return cache[key]
def run():
    t = time.perf_counter()
    codec.fit_many(tiles, 5, 2, 64, 2, 1e-3, 8192, 10, seed=19920517, in_flight=4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / len(tiles) * 1e3
run()
for rep in range(2):
    ops.randperm = real
    a = run()
    ops.randperm = memo
    run()
    b = run()
    print(f"ms per tile, four in flight: permutations computed per fit {a:.2f} | memoised (diagnostic) {b:.2f} | the pipeline costs {a - b:.2f}")
ops.randperm = real
