"""What a step of a fit ALONE on the device costs, kernel by kernel, for both training kernels of the headline shape:
k_train_stream (128 workgroups of 64 rows; alone=False) and k_train_split (256 workgroups of 32 rows; alone=True).
HIP events around whole 512-step epochs on the full 8 x 2048^2 tile, lbdrn_train_profile_mode as in bench.py:
mode 0 = the real epoch, 3 = training launches alone (cold rows), 2 = training launch doubled, 1 = reduce doubled.
usage: lone_step_probe.py [repeats=3] [--embed | --bands4] [-bc N] [--only stream|split] [--modes 0,3,2,1]
(--only / --modes: what scripts/collect_inkernel.py runs under the stamped and timeline builds -- one kernel, real epochs only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import numpy as np
import torch
from lbdrn_hip import ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
embed = "--embed" in sys.argv
bc = int(sys.argv[sys.argv.index("-bc") + 1]) if "-bc" in sys.argv else 64
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
modes = tuple(int(x) for x in sys.argv[sys.argv.index("--modes") + 1].split(",")) if "--modes" in sys.argv else (0, 3, 2, 1)
kinds = [k for k in (False, True) if only is None or (only == "split") == k]
dev = torch.device("cuda:0")
C, H, W, K, D, nl, bs = (4 if "--bands4" in sys.argv else 8), 2048, 2048, 5, 2, 2, 8192
cfg = FeatCfg(True, True, 1.4, 12, True, True) if embed else FeatCfg(False, False, 1.4, 12, True, True)
img = synthetic_tile(0, C, H, W)
img_d = ops.to_device_u16(img, dev)
msb_d, mx = ops.split_bits(img_d, K)
geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
F = cfg.feature_dim(C, D)
net = ops.make_net(F, bc, C, nl)
torch.manual_seed(1)
from lbdrn_hip.model import LBDRNModel
p0 = LBDRNModel(F, bc, C, nl).flat_parameters().to(dev)
N = H * W
nsteps = N // bs
perm = torch.randperm(N, device=dev)
ws = ops.TrainWorkspace(geom, net, bs, dev).prepare(img_d, msb_d, ops._lib.PATH_MFMA)
torch.cuda.synchronize()
stream = torch.cuda.Stream(device=dev)     # (a stream of its own, as the product's fits have)
torch.cuda.set_stream(stream)


def epoch_ms(alone, mode):
    p = p0.clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    ops.train_profile_mode(mode)
    try:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record(stream)
        ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 0, 1e-3, None, ops._lib.PATH_MFMA, ws, alone=alone)
        e.record(stream)
        torch.cuda.synchronize()
        return s.elapsed_time(e), p
    finally:
        ops.train_profile_mode(0)


for alone in kinds:
    epoch_ms(alone, 0)
res = {}
for r in range(reps):
    for alone in kinds:
        for mode in modes:
            t, p = epoch_ms(alone, mode)
            res.setdefault((alone, mode), []).append(t)
            if mode == 0:
                res.setdefault(("p", alone), p)
same = torch.equal(res[("p", False)].view(torch.int32), res[("p", True)].view(torch.int32)) if len(kinds) == 2 else None
for alone in kinds:
    if modes != (0, 3, 2, 1):
        print(f"{'k_train_split (alone)' if alone else 'k_train_stream       '}: step {min(res[(alone, 0)]) / nsteps * 1e3:.2f} us")
        continue
    t0, t3, t2, t1 = (min(res[(alone, m)]) / nsteps * 1e3 for m in (0, 3, 2, 1))
    print(f"{'k_train_split (alone)' if alone else 'k_train_stream       '}: step {t0:.2f} us | training launch alone {t3:.2f} | one more training launch "
          f"{t2 - t0:.2f} | one more reduce launch {t1 - t0:.2f} | rest {t0 - (t2 - t0) - (t1 - t0):.2f}   (epoch {min(res[(alone, 0)]):.2f} ms; all: "
          + " ".join(f"{x:.2f}" for x in res[(alone, 0)]) + ")")
print("parameters after one epoch identical bit for bit:", same)
