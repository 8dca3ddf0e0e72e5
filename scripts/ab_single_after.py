"""Experiment: one bc=256 fit alone, timed before and after a phase with two fits in flight."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
bc = int(os.environ.get("AB_BC", "256"))
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(2)]
args = (5, 2, bc, 2, 1e-3, 8192, 10)
def single(tag):
    for k in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        codec.fit_many(tiles[:1], *args, seed=19920517, in_flight=1)
        torch.cuda.synchronize(); print(tag, k, f"{(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
if not os.environ.get("AB_SKIP_BEFORE"): single("before")
if os.environ.get("AB_PREALLOC"):   # leave the null stream's pool holding early blocks of the sizes a fit asks for
    g = ops.FeatureGeometry(8, 2048, 2048, 5, 2, 100, None, dev) if False else None
    blocks = [torch.empty(int(float(x) * 2**30), dtype=torch.uint8, device=dev) for x in os.environ["AB_PREALLOC"].split(",")]
    del blocks
for k in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    codec.fit_many(tiles, *args, seed=19920517, in_flight=2)
    torch.cuda.synchronize(); print("two in flight", k, f"{(time.perf_counter() - t) * 1e3 / 2:.1f} ms/tile", flush=True)
single("after")
if os.environ.get("AB_EMPTY"):
    torch.cuda.empty_cache()
    single("after empty_cache")
