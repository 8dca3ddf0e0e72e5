"""What the chains of N fits in flight spend their time on, WITHOUT a profiler: HIP events around every
lbdrn_train_epoch and lbdrn_eval_sse call of every fit (on the fit's own stream), all measured against one reference
event.  usage: inflight_events.py [in_flight=4] [tiles=8]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
infl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i % 4, 8, 2048, 2048), dev) for i in range(n)]
args = (5, 2, 64, 2, 1e-3, 8192, 10)
rec, lock = [], threading.Lock()
def wrap(name, fn):
    def w(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = fn(*a, **k); e.record()
        with lock: rec.append((threading.get_ident(), name, s, e))
        return r
    return w
ops.train_epoch = wrap("train", ops.train_epoch)
ops.eval_sse = wrap("eval", ops.eval_sse)
orig_prepare = ops.TrainWorkspace.prepare
ops.TrainWorkspace.prepare = wrap("prepare", orig_prepare)
codec.fit_many(tiles[:max(infl, 2)], *args, seed=19920517, in_flight=infl)
torch.cuda.synchronize(); rec.clear()
ref = torch.cuda.Event(enable_timing=True); ref.record()
t0 = time.perf_counter()
codec.fit_many(tiles, *args, seed=19920517, in_flight=infl)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
print(f"in_flight={infl}: {wall / n:.2f} ms per tile ({n} tiles, wall {wall:.1f} ms)")
by = {}
for tid, name, s, e in rec: by.setdefault(tid, []).append((ref.elapsed_time(s), ref.elapsed_time(e), name))
tot = {"train": 0.0, "eval": 0.0, "prepare": 0.0}
for tid, lst in by.items():
    lst.sort()
    for a, b, name in lst: tot[name] += b - a
    ep = [b - a for a, b, name in lst if name == "train"]
    ev = [b - a for a, b, name in lst if name == "eval"]
    print(f"  chain {tid % 10000}: {len(ep)} epochs mean {sum(ep)/len(ep):.2f} ms (min {min(ep):.2f} max {max(ep):.2f}); "
          f"{len(ev)} evaluation passes mean {sum(ev)/len(ev):.2f} ms (min {min(ev):.2f} max {max(ev):.2f}); busy {sum(b - a for a, b, _ in lst):.1f} of {wall:.1f} ms")
print(f"  per tile: training epochs {tot['train']/n:.1f} ms of chain time, evaluation {tot['eval']/n:.1f}, row build {tot['prepare']/n:.1f}; "
      f"chain time available per tile {wall * infl / n:.1f} ms")
