"""Diagnostic: where the wall time of one fit goes (host vs device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops, sampler
from lbdrn_hip.synth import synthetic_tile

dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(0, 8, 2048, 2048), dev)
for it in range(3):
    torch.manual_seed(19920517)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 10)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"fit: host-side return after {1e3*(t1-t0):.1f} ms, device done after {1e3*(t2-t0):.1f} ms")
N = 2048 * 2048
t = time.perf_counter(); p = sampler.permutation(12345, N); print(f"one randperm: {1e3*(time.perf_counter()-t):.1f} ms")
t = time.perf_counter(); pp = p.pin_memory(); print(f"pin: {1e3*(time.perf_counter()-t):.1f} ms")
t = time.perf_counter(); d = pp.to(dev, non_blocking=True); torch.cuda.synchronize(); print(f"h2d: {1e3*(time.perf_counter()-t):.1f} ms")
t = time.perf_counter(); st = sampler.PermutationStream(N, 10, 1, workers=4); 
for e in range(1, 11):
    st.get(e); print(f"  perm {e} ready at {1e3*(time.perf_counter()-t):.1f} ms")
t = time.perf_counter(); st = sampler.PermutationStream(N, 10, 1, workers=10); 
for e in range(1, 11):
    st.get(e); print(f"  [10 workers] perm {e} ready at {1e3*(time.perf_counter()-t):.1f} ms")
t = time.perf_counter(); q = torch.randperm(N, device=dev); torch.cuda.synchronize(); print(f"torch gpu randperm (different order!): {1e3*(time.perf_counter()-t):.1f} ms")
