"""How much of a fit's wall time its host thread spends inside lbdrn_train_epoch (enqueueing 1024 launches per
epoch) with 1 and with 4 fits in flight: a host thread that is always inside the call is the bottleneck."""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i % 4, 8, 2048, 2048), dev) for i in range(8)]
args = (5, 2, 64, 2, 1e-3, 8192, 10)
orig = ops.train_epoch
acc = {}
lock = threading.Lock()
def timed(*a, **k):
    t = time.perf_counter(); orig(*a, **k); dt = time.perf_counter() - t
    with lock: acc[threading.get_ident()] = acc.get(threading.get_ident(), 0.0) + dt
ops.train_epoch = timed
for infl in (1, 4):
    codec.fit_many(tiles[:max(2, infl)], *args, seed=19920517, in_flight=infl, group=1)
    torch.cuda.synchronize(); acc.clear(); t = time.perf_counter()
    codec.fit_many(tiles, *args, seed=19920517, in_flight=infl, group=1)
    torch.cuda.synchronize(); wall = time.perf_counter() - t
    print(f"in_flight={infl}: wall {wall*1e3:.0f} ms for 8 tiles ({wall/8*1e3:.1f} ms/tile); host time inside train_epoch per thread: "
          + ", ".join(f"{v*1e3:.0f} ms" for v in acc.values()) + f"  = {sum(acc.values())/len(acc)/wall*100:.0f} % of the wall time each; "
          f"{sum(acc.values())/ (8*10*1024) *1e6:.2f} us per launch")
