"""Profiling driver: codec.fit_many over 4 warm-up + N timed synthetic 8 x 2048^2 tiles (argv: in_flight group [tiles]),
nothing else -- for rocprofv3 --kernel-trace + scripts/timeline_stats.py.  Prints the wall time per timed tile and the
fraction of the trace the timed part takes (so that the statistics can be cut to it)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
infl, group = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i % 4, 8, 2048, 2048), dev) for i in range(n)]
args = (5, 2, 64, 2, 1e-3, 8192, 10)
t0 = time.perf_counter()
codec.fit_many(tiles[:max(infl, 2)], *args, seed=19920517, in_flight=infl, group=group)
torch.cuda.synchronize(); t1 = time.perf_counter()
codec.fit_many(tiles, *args, seed=19920517, in_flight=infl, group=group)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"in_flight={infl} group={group}: {(t2 - t1) / n * 1e3:.2f} ms/tile; timed part = last {(t2 - t1) / (t2 - t0):.3f} of the run")
