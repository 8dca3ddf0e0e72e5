"""Experiment: fits on streams created with hipExtStreamCreateWithCUMask (each chain confined to a set of CUs).
usage: cumask_probe.py"""
import ctypes, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    if rc != 0: raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(h.value, device=dev)
tiles = [ops.to_device_u16(synthetic_tile(i % 4, 8, 2048, 2048), dev) for i in range(8)]
args = (5, 2, 64, 2, 1e-3, 8192, 10)
def lone(stream, label):
    for it in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        with torch.cuda.stream(stream):
            codec.fit_device(tiles[0], *args, seed=19920517, alone=False)
        torch.cuda.synchronize()
    print(f"{label}: one fit alone (evaluation in the chain) {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
lone(torch.cuda.Stream(device=dev), "unmasked stream")
masks = {"low half": set(range(128)), "high half": set(range(128, 256)), "even CUs": set(range(0, 256, 2)),
         "first quarter": set(range(64)), "all": set(range(256))}
streams = {k: masked_stream(v) for k, v in masks.items()}
for k in ("low half",): lone(streams[k], k)
def many(pool, infl, label, n=8):
    codec._FIT_STREAMS[dev] = list(pool)
    codec.fit_many(tiles[:infl], *args, seed=19920517, in_flight=infl)
    out = []
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        codec.fit_many((tiles * 4)[:n], *args, seed=19920517, in_flight=infl)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / n * 1e3)
    print(f"{label}: {infl} in flight, {n} tiles: " + " ".join(f"{x:.2f}" for x in out) + " ms per tile", flush=True)
plain = [torch.cuda.Stream(device=dev) for _ in range(8)]
lo = [masked_stream(masks["low half"]) for _ in range(4)]
hi = [masked_stream(masks["high half"]) for _ in range(4)]
il = lambda k: [x for pair in zip(lo[:k], hi[:k]) for x in pair]
many(plain, 4, "plain streams", 12)
many(plain, 3, "plain streams", 12)
many(il(1), 2, "low | high", 12)
many(il(2), 4, "low | high, two chains each", 12)
many(il(3), 6, "low | high, three chains each", 12)
many(il(4), 8, "low | high, four chains each", 16)
many(plain, 6, "plain streams", 12)
many(plain, 4, "plain streams again", 12)
