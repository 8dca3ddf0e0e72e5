"""End-to-end CLI timing on one full-size tile: encode.py then decode.py in-process (module import and
library load excluded by a small warm-up image first), as the log's own 'Time elapsed' records."""
import os, re, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import numpy as np
import decode, encode
from lbdrn_hip import raster_io
from lbdrn_hip.synth import synthetic_tile

with tempfile.TemporaryDirectory() as d:
    for name, img in (("warm", synthetic_tile(1, 8, 128, 128)), ("tile", synthetic_tile(0, 8, 2048, 2048))):
        for ext in (".tif", ".npy"):
            src = os.path.join(d, name + ext)
            t0 = time.time(); raster_io.write_raster(src, img); tw = time.time() - t0
            out = os.path.join(d, "out" + ext[1:])
            t0 = time.time(); encode.main(["-i", src, "-o", out]); te = time.time() - t0
            sub = os.path.join(out, f"{name}_r1_K5_bc64_nl2_D2_prec16_lr0.001_bs8192_e10")
            t0 = time.time(); decode.main(["-i", os.path.join(sub, name + ".bin"), "-org", src]); td = time.time() - t0
            log = open(os.path.join(sub, "decode.txt")).read()
            size = os.path.getsize(os.path.join(sub, name + ".bin"))
            print(f"== {name}{ext}: write {tw:.2f}s encode.main {te:.3f}s decode.main(+metrics) {td:.3f}s  bin {size} B  "
                  + " ".join(re.findall(r"(PSNR: \S+|bpsp=\S+|Time elapsed: \S+)", log)), flush=True)
