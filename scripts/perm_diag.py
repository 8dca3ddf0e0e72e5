"""Diagnostic: ms per tile (16 tiles, four in flight, fits without decode) -- run once per timing-only library variant:
    python lbdrn-msic_amd/csrc/build.py --variant permdiagN -DLBDRN_EXP_RANDPERM_DIAG=N    (N = 1: no MT19937 launch, 2: none of the
    launches behind it, 3: neither; the permutations are then garbage) and LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_permdiagN.so"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
tiles = [ops.to_device_u16(synthetic_tile(i, 8, 2048, 2048), dev) for i in range(8)] * 2
def run():
    t = time.perf_counter()
    codec.fit_many(tiles, 5, 2, 64, 2, 1e-3, 8192, 10, seed=19920517, in_flight=4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / len(tiles) * 1e3
run()
print("library =", os.path.basename(os.environ.get("LBDRN_HIP_LIB", "liblbdrn_hip.so")), ["%.2f" % run() for _ in range(3)], "ms per tile")
