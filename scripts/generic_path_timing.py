"""What the generic (LDS-tiled GEMM) kernels cost on the headline tile -- the path every shape without a fused kernel takes --
beside the fused kernels, for both hidden activations (lbdrn_net.act: Sine(30), and nn.ReLU, which has been a template
argument of the fused kernels since round 6).  One 8 x 2048 x 2048 tile, two epochs + decode, per activation and path.
usage: generic_path_timing.py [side=2048] [epochs=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(1000, 8, side, side), dev)
for act, path in (("sine", ops._lib.PATH_MFMA), ("relu", ops._lib.PATH_MFMA), ("relu", ops._lib.PATH_AUTO),
                  ("sine", ops._lib.PATH_GENERIC), ("relu", ops._lib.PATH_GENERIC)):
    cfg = FeatCfg(activation=act)
    for rep in range(2):
        torch.manual_seed(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, epochs, cfg=cfg, path=path)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        rec = ops.decode_fused(fit.geom, fit.net, fit.msb, fit.best_params, path=path)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{act:5s} path {('auto', 'generic', 'mfma')[path]:8s}: fit of {epochs} epochs {1e3 * (t1 - t0):8.1f} ms = {1e3 * (t1 - t0) / epochs:7.1f} ms per epoch "
          f"(training + evaluation pass) | decode {1e3 * (t2 - t1):7.1f} ms", flush=True)
