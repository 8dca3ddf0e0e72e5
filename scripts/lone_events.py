"""Where the wall time of one fit alone goes WITHOUT a profiler attached: HIP events on the fit's stream around every
lbdrn_train_epoch call (ops.train_epoch is wrapped), the time before the first and after the last.
usage: lone_events.py [fits=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(0, 8, 2048, 2048), dev)
orig = ops.train_epoch
marks = []
def wrapped(*a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(*a, **k); e.record(); marks.append((s, e))
ops.train_epoch = wrapped
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    marks.clear()
    torch.cuda.synchronize()
    b, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); b.record()
    fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 10, seed=19920517)
    f.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    ep = [s.elapsed_time(e) for s, e in marks]
    between = [marks[i][1].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)]
    print(f"fit {it}: wall {1e3 * (t1 - t0):.1f} ms; start -> first epoch {b.elapsed_time(marks[0][0]):.2f} ms; epochs "
          + " ".join(f"{x:.2f}" for x in ep) + f" (sum {sum(ep):.1f}); between epochs " + " ".join(f"{x:.2f}" for x in between)
          + f" (sum {sum(between):.2f}); last epoch -> end {marks[-1][1].elapsed_time(f):.2f} ms")
