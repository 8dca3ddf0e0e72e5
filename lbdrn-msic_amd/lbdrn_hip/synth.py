"""Synthetic multispectral tiles for benchmarks and tests: per band a sum of six
low-frequency 2-D sinusoids scaled to [500, 9500] plus N(0, 40^2) noise, rounded and clipped to
[0, 10000] -- spatially correlated high bits, near-uniform low bits."""
import math

import numpy as np


def synthetic_tile(i, C=8, H=2048, W=2048):
    rng = np.random.default_rng(1000 + i)
    yy = (np.arange(H, dtype=np.float32) / max(H, 1))[:, None]
    xx = (np.arange(W, dtype=np.float32) / max(W, 1))[None, :]
    img = np.empty((C, H, W), np.uint16)
    for c in range(C):
        acc = np.zeros((H, W), np.float32)
        for _ in range(6):
            fy, fx = rng.uniform(0.5, 6.0, 2)
            ph = rng.uniform(0, 2 * math.pi)
            amp = rng.uniform(0.3, 1.0)
            acc += np.float32(amp) * np.sin(np.float32(2 * math.pi) * (np.float32(fy) * yy + np.float32(fx) * xx) + np.float32(ph))
        lo, hi = float(acc.min()), float(acc.max())
        acc = 500.0 + (acc - lo) / max(hi - lo, 1e-12) * 9000.0
        acc += rng.normal(0.0, 40.0, (H, W)).astype(np.float32)
        img[c] = np.clip(np.rint(acc), 0, 10000).astype(np.uint16)
    return img
