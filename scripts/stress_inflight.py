"""Stress: many fits of mixed sizes with 2-4 in flight must equal the one-by-one results bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import numpy as np, torch
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
shapes = [(int(rng.integers(1, 9)), int(rng.integers(40, 400)), int(rng.integers(40, 400))) for _ in range(24)]
shapes += [(8, 2048, 2048), (8, 1024, 1500), (4, 1536, 700)]
imgs = [ops.to_device_u16(synthetic_tile(100 + i, *s), dev) for i, s in enumerate(shapes)]
args = (5, 2, 64, 2, 1e-3, 2048, 4)
ref = codec.fit_many(imgs, *args, seed=19920517, in_flight=1)
torch.cuda.synchronize()
for infl in (2, 3, 4, 2):
    t = time.perf_counter()
    got = codec.fit_many(imgs, *args, seed=19920517, in_flight=infl,
                         then=lambda fit: (fit, codec.apply_device(fit.geom, fit.net, fit.msb, codec.truncate_device(fit.best_params, 16))))
    torch.cuda.synchronize()
    bad = sum(not (torch.equal(a.best_params.view(torch.int32), b[0].best_params.view(torch.int32)) and torch.equal(a.mse_log, b[0].mse_log))
              for a, b in zip(ref, got))
    print(f"in_flight={infl}: {len(imgs)} fits in {time.perf_counter() - t:.2f}s, mismatches {bad}", flush=True)
    assert bad == 0
print("stress ok")
