"""The reference's real image shape, once: a synthetic 8-band 6000 x 6000 uint16 scene -- the size
of the GF6-WFI scenes of the reference's tables -- through encode.py / decode.py in-process, with
-sr 1 (one 36 M-pixel fit: a 30 GB row matrix, ten 288 MB permutations from lbdrn_randperm's n > 2048^2 path) and -sr 3
(nine 2000 x 2000 tiles, codec.fit_many sizing its fits in flight against the free memory).  Prints one JSON record per
run: wall times, the device memory high-water mark, bytes, PSNR, whether the high bits came back exact.
usage: scene_timing.py [side=6000] [epochs=10] [sr ...=1 3]       LBDRN_SCENE_BANDS=4: the 4-band scenes of the reference's list
(GF-2, GF6-PMS: 9 of the 13 images of run.sh:14-28; side 7550 ~ a 57 M-pixel GF-2 scene)"""
import json, os, re, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import numpy as np
import torch
import decode, encode
from lbdrn_hip import codec, ops, raster_io, sampler
from lbdrn_hip.synth import synthetic_tile

side = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
srs = [int(x) for x in sys.argv[3:]] or [1, 3]
C, K = int(os.environ.get("LBDRN_SCENE_BANDS", "8")), 5
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as d:
    warm = os.path.join(d, "warm.npy")
    raster_io.write_raster(warm, synthetic_tile(1, C, 96, 96))
    encode.main(["-i", warm, "-o", os.path.join(d, "w"), "-e", "2", "-bs", "1024"])
    t0 = time.time()
    img = synthetic_tile(7, C, side, side)
    src = os.path.join(d, "scene.npy")
    raster_io.write_raster(src, img)
    t_make = time.time() - t0
    # the permutation of one epoch of the whole scene, timed on its own (n > 2048^2: the memory-side atomic path)
    n = side * side
    torch.cuda.synchronize()
    ops.randperm([1], n, dev)
    torch.cuda.synchronize()
    t0 = time.time()
    p = ops.randperm([123456789], n, dev)
    torch.cuda.synchronize()
    t_perm = time.time() - t0
    ok_perm = bool(torch.equal(p[0].cpu(), torch.randperm(n, generator=torch.Generator().manual_seed(123456789))))   # the whole vector, any n
    del p
    for sr in srs:
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        out = os.path.join(d, f"out{sr}")
        t0 = time.time()
        assert encode.main(["-i", src, "-o", out, "-sr", str(sr), "-e", str(epochs)]) in (0, None)
        torch.cuda.synchronize()
        t_enc = time.time() - t0
        sub = os.path.join(out, f"scene_r{sr}_K{K}_bc64_nl2_D2_prec16_lr0.001_bs8192_e{epochs}")
        t0 = time.time()
        assert decode.main(["-i", os.path.join(sub, "scene.bin")]) in (0, None)     # (without -org: the raster stays for the checks below)
        t_dec = time.time() - t0
        peak = torch.cuda.max_memory_allocated(dev)
        out_img = raster_io.read_raster(os.path.join(sub, "scene_recon.tif"))
        high_ok = bool(np.array_equal(out_img >> K, img >> K))
        mse = float(np.mean((img.astype(np.float32) - out_img.astype(np.float32)) ** 2))
        nbytes = os.path.getsize(os.path.join(sub, "scene.bin"))
        tile = (side // sr + side % sr)
        rec = {"scene": f"{C} x {side} x {side} uint16, synthetic", "split_ratio": sr, "epochs": epochs,
               "make_and_write_s": round(t_make, 2), "encode_main_s": round(t_enc, 2), "decode_main_s": round(t_dec, 2),
               "device_peak_allocated_GiB": round(peak / 2**30, 2),
               "fit_bytes_estimate_GiB_per_fit": round(codec.fit_bytes(C, tile, tile, K, 2, 64, 2, 8192, epochs) / 2**30, 2),
               "bin_bytes": nbytes, "bpsp": round(8 * nbytes / img.size, 4), "mse": round(mse, 3),
               "psnr_peak_10000": round(10 * np.log10(1e8 / mse), 2), "high_bits_exact": high_ok,
               "randperm_whole_scene_ms": round(t_perm * 1e3, 2), "randperm_n": n, "randperm_equals_torch": ok_perm}
        del out_img
        print(json.dumps(rec), flush=True)
