"""Experiment: one fit alone on the legacy default stream vs on a torch stream, in a fresh process and after a phase
with several fits in flight."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
bc = int(os.environ.get("AB_BC", "64"))
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(4)]
args = (5, 2, bc, 2, 1e-3, 8192, 10)
own = torch.cuda.Stream(device=dev)
def single(tag, stream):
    for k in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        if stream is None:
            codec.fit_device(tiles[0], *args, seed=19920517)
        else:
            stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(stream):
                codec.fit_device(tiles[0], *args, seed=19920517)
        torch.cuda.synchronize(); print(tag, k, f"{(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
single("fresh, null stream", None)
single("fresh, own stream", own)
codec.fit_many(tiles, *args, seed=19920517, in_flight=4)
torch.cuda.synchronize()
single("after four in flight, null stream", None)
single("after four in flight, own stream", own)
