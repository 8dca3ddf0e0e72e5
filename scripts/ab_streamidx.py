"""Experiment: does it matter which pool stream the evaluation passes of a fit run on?  N dummy torch streams are
created before the first fit (so the fit's side streams come later in torch's pool)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tile = ops.to_device_u16(synthetic_tile(0, 8, 2048, 2048), dev)
args = (5, 2, 64, 2, 1e-3, 8192, 10)
n = int(sys.argv[1])
use = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dummies = [torch.cuda.Stream(device=dev) for _ in range(n)]
for s in dummies[:use]:      # make them "used" streams: a kernel each
    with torch.cuda.stream(s):
        torch.zeros(16, device=dev)
torch.cuda.synchronize()
out = []
for k in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    if os.environ.get("AB_OWN"):
        if k == 0: own = torch.cuda.Stream(device=dev)
        own.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(own):
            codec.fit_device(tile, *args, seed=19920517)
    else:
        codec.fit_device(tile, *args, seed=19920517)
    torch.cuda.synchronize(); out.append(f"{(time.perf_counter() - t) * 1e3:.1f}")
print(f"{n} streams created first ({use} of them used): fits", " ".join(out), "ms", flush=True)
