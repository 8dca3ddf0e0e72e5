"""Profiling driver: ONE chain of two fits per launch (codec.fit_group), alone on the device -- the launch sequence of
`bench.py`'s default run (two such chains in flight) without the second chain.  For rocprofv3 --kernel-trace --stats:
the average duration of k_train_stream here is that of the 2 x 128-workgroup launch.
    python3 scripts/prof_pair.py [SIDE EPOCHS [embed | bands4]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch  # noqa: E402
from lbdrn_hip import codec, ops  # noqa: E402
from lbdrn_hip.synth import synthetic_tile  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
from lbdrn_hip.features import FeatCfg  # noqa: E402
embed = len(sys.argv) > 3 and sys.argv[3] == "embed"
bands = 4 if len(sys.argv) > 3 and sys.argv[3] == "bands4" else 8
cfg = FeatCfg(use_coordinates=embed, embedding=embed)
dev = torch.device("cuda:0")
imgs = [ops.to_device_u16(synthetic_tile(i, bands, side, side), dev) for i in range(2)]
for _ in range(2):
    fits = codec.fit_group(imgs, 5, 2, 64, 2, 1e-3, 8192, epochs, seed=19920517, cfg=cfg)
torch.cuda.synchronize()
print("done", [float(f.mse_log[:, 0].min()) for f in fits])
