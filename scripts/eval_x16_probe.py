"""The evaluation pass three ways on a full-size tile (8 and 4 bands): canonical, fast (LBDRN_EVAL_FAST: what a fit runs), and
fast with layer 0's colour features on the f16 matrix pipe with exact operands (LBDRN_EVAL_X16, opt-in): time per
pass by HIP events, the float64 sums and their relative distances, and a whole tile (fit + decode) with and without the flag.
    python scripts/eval_x16_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
import torch
from lbdrn_hip import codec, ops
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile

dev = torch.device("cuda:0")
for bands in (8, 4):
    img = ops.to_device_u16(synthetic_tile(0, bands, 2048, 2048), dev)
    torch.manual_seed(19920517)
    fit = codec.fit_device(img, 5, 2, 64, 2, 1e-3, 8192, 2)
    ws = ops.ApplyWorkspace(fit.geom, fit.net, dev)
    res = {}
    for name, kw in (("canonical", {}), ("fast", dict(fast=True)), ("fast + x16", dict(fast=True, x16=True))):
        for _ in range(2):
            ops.eval_sse(fit.geom, fit.net, img, fit.msb, fit.best_params, ws=ws, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            sse = ops.eval_sse(fit.geom, fit.net, img, fit.msb, fit.best_params, ws=ws, **kw)
        e.record(); e.synchronize()
        res[name] = (s.elapsed_time(e) / 5, float(sse.item()))
    c = res["canonical"][1]
    for name, (ms, v) in res.items():
        print(f"{bands} bands  {name:11s}: {ms:7.3f} ms per pass   sse {v:.12e}   relative distance to canonical {abs(v - c) / c:.2e}", flush=True)
    for flag in ("0", "1", "0", "1"):
        os.environ["LBDRN_EVAL_X16"] = flag
        laps = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tiles = [img] * 4
            fits = codec.fit_many(tiles, 5, 2, 64, 2, 1e-3, 8192, 10, seed=19920517, in_flight=4)
            torch.cuda.synchronize(); laps.append((time.perf_counter() - t0) / 4 * 1e3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lone = codec.fit_many([img], 5, 2, 64, 2, 1e-3, 8192, 10, seed=19920517, in_flight=1)
        torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) * 1e3
        print(f"{bands} bands  LBDRN_EVAL_X16={flag}: four fits in flight {min(laps):7.2f} ms per tile (fit only), one alone {t1:7.2f} ms", flush=True)
    os.environ.pop("LBDRN_EVAL_X16", None)
