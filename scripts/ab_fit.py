"""A/B aid: median wall time of a full 10-epoch fit of one 8x2048^2 tile with the library named by
LBDRN_HIP_LIB (run old and new builds alternately inside ONE gpurun call: boxes differ by a few percent)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(0, 8, 2048, 2048), dev)
ts = []
for it in range(6):
    torch.manual_seed(19920517)
    torch.cuda.synchronize(); t = time.perf_counter()
    fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 10)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
ts = sorted(ts[1:])
print(f"{os.path.basename(os.environ.get('LBDRN_HIP_LIB', 'liblbdrn_hip.so'))}: fit median {ts[len(ts)//2]*1e3:.2f} ms  min {ts[0]*1e3:.2f}  mse {float(fit.mse_log[:,0].min()):.6f}")
