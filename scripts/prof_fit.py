"""Profiling driver: the hot path over one synthetic 8 x 2048^2 tile, twice (the first pass warms allocator pools and
caches), one fit at a time on the default stream -- fit (epochs, an evaluation pass after each), weight truncation,
decode.  Used under `rocprofv3 --pmc ...` (scripts/profile_round.sh): bench.py's worker threads and event probes are
not needed for counter collection, and the counter passes serialise every dispatch anyway.
    python3 scripts/prof_fit.py SIDE BC EPOCHS [embed | bands4]      (bands4: the reference's 4-band shape, F = 100)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch  # noqa: E402
from lbdrn_hip import codec, ops  # noqa: E402
from lbdrn_hip.features import FeatCfg  # noqa: E402
from lbdrn_hip.synth import synthetic_tile  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 10
embed = len(sys.argv) > 4 and sys.argv[4] == "embed"
bands = 4 if len(sys.argv) > 4 and sys.argv[4] == "bands4" else 8
cfg = FeatCfg(use_coordinates=embed, embedding=embed)
dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(0, bands, side, side), dev)
for _ in range(2):
    fit = codec.fit_device(img, 5, 2, bc, 2, 1e-3, 8192, epochs, seed=19920517, cfg=cfg)
    rec = codec.apply_device(fit.geom, fit.net, fit.msb, codec.truncate_device(fit.best_params, 16))
torch.cuda.synchronize()
print("done", float(fit.mse_log[:, 0].min()))
