"""One tile alone on the device, as bench.py's `single_tile_ms` measures it (fit + truncation + decode, median of 3), under
whatever environment the caller set (LBDRN_LONE_HEAD_FRAC, LBDRN_OVERLAP_EVAL, LBDRN_HIP_LIB): prints one line.
usage: lone_tile.py [label] [bench.py flags]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
label = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
sys.argv = [sys.argv[0]] + [x for x in sys.argv[1:] if x.startswith("-") or x.isdigit()]
import torch
import bench
from lbdrn_hip import codec, ops
from lbdrn_hip.synth import synthetic_tile
a = bench.parse()
dev = torch.device("cuda:0")
path = {"auto": ops._lib.PATH_AUTO, "generic": ops._lib.PATH_GENERIC, "mfma": ops._lib.PATH_MFMA}[a.path]
img = ops.to_device_u16(synthetic_tile(1000, a.bands, a.height, a.width), dev)
bench.lone_tile(codec, ops, img, a, path)
r = [bench.lone_tile(codec, ops, img, a, path) for _ in range(2)]
print(f"{label:24s} lone tile {min(x['ms'] for x in r):.2f} ms (encode {min(x['encode_ms'] for x in r):.2f}, decode {min(x['decode_ms'] for x in r):.2f})")
