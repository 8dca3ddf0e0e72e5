"""What the nine per-epoch evaluation passes before the last cost a tile: fit_many with val_duration = 1 against 10 (one pass
at the end), four fits in flight as pairs and one fit alone.  Round 4: 57.6 against 44.9 ms per tile in flight (1.41 ms
per pass = its own duration: no penalty in the mix), 117.1 against 111.3 alone (0.64 ms each: in the background of the
next epoch)."""
import os, sys, time, statistics
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i % 4, 8, 2048, 2048), dev) for i in range(12)]
def run(vd, infl, group):
    torch.cuda.synchronize(); t = time.perf_counter()
    codec.fit_many(tiles, 5, 2, 64, 2, 1e-3, 8192, 10, vd, seed=19920517, in_flight=infl, group=group)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / len(tiles) * 1e3
for vd in (1, 10): run(vd, 4, 2)
res = {}
for rep in range(3):
    for vd in (1, 10):
        for infl, g in ((4, 2), (1, 1)):
            res.setdefault((vd, infl), []).append(run(vd, infl, g))
for k, v in sorted(res.items()):
    print(f"val_duration={k[0]} in_flight={k[1]}: median {statistics.median(v):.2f} ms/tile {['%.2f' % x for x in v]}")
