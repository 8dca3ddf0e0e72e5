"""Diagnostic: what the launches of the bc = 256 training step cost, by HIP events over one 512-step epoch in each
profile mode (lbdrn_hip.h): 0 the real step; 3 forward/backward + weight-gradient launches without the reduce/Adam launch;
4 the forward/backward launch alone; 5 the weight-gradient launch doubled; 1 the reduce launch doubled; 2 the
forward/backward launch doubled.     python3 scripts/wide_probe.py [BC [SIDE [embed]]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch  # noqa: E402
from lbdrn_hip import ops  # noqa: E402
from lbdrn_hip.features import FeatCfg  # noqa: E402
from lbdrn_hip.synth import synthetic_tile  # noqa: E402

bc = int(sys.argv[1]) if len(sys.argv) > 1 else 256
side = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
embed = len(sys.argv) > 3 and sys.argv[3] == "embed"
dev = torch.device("cuda:0")
cfg = FeatCfg(use_coordinates=embed, embedding=embed)
img = ops.to_device_u16(synthetic_tile(0, 8, side, side), dev)
msb, mx = ops.split_bits(img, 5)
geom = ops.FeatureGeometry(8, side, side, 5, 2, mx, cfg, dev)
net = ops.make_net(geom.F, bc, 8, 2)
from lbdrn_hip.model import LBDRNModel  # noqa: E402
torch.manual_seed(1)
p = LBDRNModel(geom.F, bc, 8, 2).flat_parameters().to(dev)
N, bs = side * side, 8192
perm = torch.randperm(N, device=dev)
st = (p.clone(), torch.zeros_like(p), torch.zeros_like(p))
ws = ops.TrainWorkspace(geom, net, bs, dev).prepare(img, msb, ops.PATH_AUTO)
run = lambda: ops.train_epoch(geom, net, img, msb, perm, bs, *st, 0, 1e-6, None, ops.PATH_AUTO, ws)
run()
stream = torch.cuda.current_stream()
steps = (N + bs - 1) // bs
t = {}
for rep in range(2):
    for mode in (0, 3, 4, 5, 1, 2):
        ops.train_profile_mode(mode)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream); run(); e.record(stream); e.synchronize()
        t.setdefault(mode, []).append(s.elapsed_time(e) / steps * 1e3)
ops.train_profile_mode(0)
m = {k: min(v) for k, v in t.items()}
print(f"bc={bc} per step (us): step {m[0]:.2f} | fwd/bwd + dW without reduce {m[3]:.2f} | fwd/bwd alone {m[4]:.2f} | "
      f"dW = {m[3] - m[4]:.2f} (one more: {m[5] - m[0]:.2f}) | one more reduce {m[1] - m[0]:.2f} | one more fwd/bwd {m[2] - m[0]:.2f}")
