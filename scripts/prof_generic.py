"""One epoch of a ReLU fit of the headline tile -- the generic (LDS-tiled GEMM) kernels -- for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_generic -o gp -- python3 scripts/prof_generic.py
    python scripts/rocprof_kernel_stats.py gpurun_out/prof_generic/gp_results.db profiles/rNN_kernel_stats_generic_relu.csv"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile
dev = torch.device("cuda:0")
img = ops.to_device_u16(synthetic_tile(1000, 8, 2048, 2048), dev)
torch.manual_seed(1)
fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 1, cfg=FeatCfg(activation="relu"))
torch.cuda.synchronize()
print("done")
