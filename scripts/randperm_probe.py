"""Diagnostic: time of lbdrn_randperm for the ten permutations of a 2048^2 fit, alone on the device (HIP events), and
equality of the first one with torch.randperm."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import ops
dev = torch.device("cuda:0")
n = 2048 * 2048
seeds = [1234567 + 17 * k for k in range(10)]
for _ in range(2):
    p = ops.randperm(seeds, n, dev)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); 
for _ in range(3): p = ops.randperm(seeds, n, dev)
e.record(); e.synchronize()
print(f"ten permutations of {n}: {s.elapsed_time(e) / 3:.2f} ms (batch of ten seeds: MT19937 waves side by side, then the per-permutation kernels one after another)")
g = torch.Generator(); g.manual_seed(seeds[0])
print("equal to torch.randperm:", bool(torch.equal(p[0].cpu(), torch.randperm(n, generator=g))))
