"""Diagnostic: what two waves per SIMD buy a training launch.  N host threads (argv[2], default 2), each with its own
stream, fit state and workspace, run epochs of the SAME launch back to back in profile mode MODE (argv[3]; 4 = the
forward/backward launch of the bc >= 128 step alone, 3 = the training launches without reduce) -- k_train_half's
workgroups (72 KB of LDS, 194 registers) co-reside two per CU, so the two chains' launches share every SIMD.  Reported:
wall time per launch with one chain and with N chains side by side; N x (one chain) means no overlap at all, 1 x
perfect overlap.     python3 scripts/overlap_probe.py [BC [N [MODE]]]"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch  # noqa: E402
from lbdrn_hip import ops  # noqa: E402
from lbdrn_hip.features import FeatCfg  # noqa: E402
from lbdrn_hip.model import LBDRNModel  # noqa: E402
from lbdrn_hip.synth import synthetic_tile  # noqa: E402

bc = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 4
side, bs, epochs = 2048, 8192, 3
dev = torch.device("cuda:0")
cfg = FeatCfg()
chains = []
for k in range(nthreads):
    img = ops.to_device_u16(synthetic_tile(k, 8, side, side), dev)
    msb, mx = ops.split_bits(img, 5)
    geom = ops.FeatureGeometry(8, side, side, 5, 2, mx, cfg, dev)
    net = ops.make_net(geom.F, bc, 8, 2)
    torch.manual_seed(k)
    p = LBDRNModel(geom.F, bc, 8, 2).flat_parameters().to(dev)
    chains.append(dict(img=img, msb=msb, geom=geom, net=net, st=(p.clone(), torch.zeros_like(p), torch.zeros_like(p)),
                       perm=torch.randperm(side * side, device=dev),
                       ws=ops.TrainWorkspace(geom, net, bs, dev).prepare(img, msb, ops.PATH_AUTO),
                       stream=torch.cuda.Stream(device=dev)))
torch.cuda.synchronize()
steps = side * side // bs


def run(c, barrier, out, idx):
    ops.train_profile_mode(mode)
    with torch.cuda.stream(c["stream"]):
        f = lambda: ops.train_epoch(c["geom"], c["net"], c["img"], c["msb"], c["perm"], bs, *c["st"], 0, 1e-6, None, ops.PATH_AUTO, c["ws"])
        f()
        c["stream"].synchronize()
        barrier.wait()
        t = time.perf_counter()
        for _ in range(epochs):
            f()
        c["stream"].synchronize()
        out[idx] = (time.perf_counter() - t) / (epochs * steps) * 1e6
    ops.train_profile_mode(0)


for n in (1, nthreads, 1):
    barrier = threading.Barrier(n)
    out = [0.0] * n
    ts = [threading.Thread(target=run, args=(chains[k], barrier, out, k)) for k in range(n)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    print(f"bc={bc} mode {mode}: {n} chain(s) side by side: {max(out):.2f} us per launch of each chain" + (f" = {max(out) / n:.2f} us per launch overall" if n > 1 else ""))
