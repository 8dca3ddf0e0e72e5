"""What the reference's own MSB format costs a drop-in run, file to file: encode.main's wall time with
the default LBB2 payload (coded on the GPU), with LBDRN_BASE_CODEC=jp2 coded after the fits (round 5) and coded beside them
(round 6: encode.BasePayloadsAhead), on one 8 x 2048^2 tile and -- with `scene` -- on the 8 x 6000 x 6000 scene at -sr 1 / 3.
    python scripts/jp2_ahead_timing.py [scene]"""
import os, re, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import numpy as np
import torch
import encode
from lbdrn_hip import jp2, raster_io
from lbdrn_hip.synth import synthetic_tile

cases = [("tile", 2048, 1, 0)] + ([("scene", 6000, 1, 7), ("scene", 6000, 3, 7)] if "scene" in sys.argv[1:] else [])
print(f"host threads available {len(os.sched_getaffinity(0))}, OpenJPEG worker threads {jp2.default_threads()}", flush=True)
with tempfile.TemporaryDirectory() as d:
    warm = os.path.join(d, "warm.npy")
    raster_io.write_raster(warm, synthetic_tile(1, 8, 128, 128))
    encode.main(["-i", warm, "-o", os.path.join(d, "w")])
    made = {}
    for name, side, sr, seed in cases:
        if (name, side) not in made:
            src = os.path.join(d, f"{name}.npy")
            raster_io.write_raster(src, synthetic_tile(seed, 8, side, side))
            made[(name, side)] = src
        src = made[(name, side)]
        for label, codec_name, ahead in (("LBB2 (GPU)", "LBB2", "1"), ("jp2 after the fits", "jp2", "0"), ("jp2 beside the fits", "jp2", "1"),
                                         ("LBB2 (GPU)", "LBB2", "1"), ("jp2 beside the fits", "jp2", "1")):
            encode.BASE_CODEC = codec_name
            os.environ["LBDRN_JP2_AHEAD"] = ahead
            out = os.path.join(d, f"o_{name}_{sr}_{codec_name}_{ahead}_{time.time_ns()}")
            torch.cuda.synchronize()
            t0 = time.time()
            encode.main(["-i", src, "-o", out, "-sr", str(sr)])
            te = time.time() - t0
            sub = os.path.join(out, f"{name}_r{sr}_K5_bc64_nl2_D2_prec16_lr0.001_bs8192_e10")
            log = open(os.path.join(sub, "encode.txt")).read()
            fit = re.findall(r"fit (\S+)s on", log)
            waited = re.findall(r"waited (\S+)s more", log)
            msb = sum(int(v) for v in re.findall(r"MSB: (\d+) bytes", log))
            print(f"{name} {side}x{side}x8 -sr {sr}  {label:20s}: encode.main {te:7.3f} s | fit per tile {fit[0] if fit else '?'} s | "
                  f"waited for payloads {sum(float(w) for w in waited):.3f} s | MSB payload {msb} B = {msb * 8 / (8 * side * side):.3f} bpsp", flush=True)
