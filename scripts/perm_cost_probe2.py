"""Diagnostic: is the permutation pipeline's cost in the four-in-flight mix its KERNELS or its BUFFERS?  Three modes of
codec.fit_many over 16 tiles: (a) as shipped; (b) the real kernels every time, but writing into buffers that are
allocated once per (count, n) and reused round-robin (no allocator traffic on the side stream); (c) results memoised
(no kernels at all: scripts/perm_cost_probe.py)."""
import os, sys, time, ctypes
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(8)] * 2
real = ops.randperm
pool, cache = {}, {}
def pooled(seeds, n, device):
    seeds = list(seeds)
    key = (len(seeds), int(n))
    ring = pool.setdefault(key, {"k": 0, "bufs": []})
    if len(ring["bufs"]) < 12:      # enough that a buffer is never reused while a fit still reads it
        nbytes = ops.lib().lbdrn_randperm_workspace(n, len(seeds))
        ring["bufs"].append((torch.empty((len(seeds), n), dtype=torch.int64, device=device), torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device), nbytes))
        out, ws, nbytes = ring["bufs"][-1]
    else:
        out, ws, nbytes = ring["bufs"][ring["k"] % 12]
        ring["k"] += 1
    arr = (ctypes.c_uint64 * len(seeds))(*[s & 0xFFFFFFFFFFFFFFFF for s in seeds])
    ops._call(ops.lib().lbdrn_randperm, out, arr, len(seeds), n, ops._ptr(out), ops._ptr(ws), nbytes)
    return out
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
    res = {}
    for name, fn in (("as shipped", real), ("kernels into reused buffers", pooled), ("memoised", memo)):
        ops.randperm = fn
        run()
        res[name] = run()
    print(" | ".join(f"{k} {v:.2f}" for k, v in res.items()), "ms per tile")
ops.randperm = real
