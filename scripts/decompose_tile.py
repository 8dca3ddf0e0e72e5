"""Diagnostic: where a tile's time goes with four fits in flight -- wall time per tile of codec.fit_many over 16 tiles
(a) as the bench runs it, (b) with ONE evaluation pass instead of ten (val_duration = epochs), (c) as (a) plus the decode
pass; the differences are what nine evaluation passes / the decode cost in the mix."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
bc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
infl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(8)] * 2
fin = lambda fit: codec.apply_device(fit.geom, fit.net, fit.msb, codec.truncate_device(fit.best_params, 16))
def run(vd, then):
    t = time.perf_counter()
    codec.fit_many(tiles, 5, 2, bc, 2, 1e-3, 8192, 10, val_duration=vd, seed=19920517, in_flight=infl, then=then)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / len(tiles) * 1e3
run(1, fin); run(10, None)
for rep in range(2):
    a, b, c = run(1, None), run(10, None), run(1, fin)
    print(f"bc={bc} in flight {infl}: fit with ten evaluation passes {a:.2f} ms/tile | with one {b:.2f} | nine passes cost {a - b:.2f} | fit + decode {c:.2f} (decode {c - a:.2f})")
