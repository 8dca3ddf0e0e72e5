"""INTEGRATION.md is executable: the binding stub and the examples it prints for a maintainer of the reference are run here,
verbatim, against the built libraries -- the decode body and the training loop on the GPU (results against the codec's own
path), the weight-payload and JPEG 2000 examples on the host."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lbdrn-msic_amd")


def _blocks():
    s = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    b = re.findall(r"```python\n(.*?)```", s, re.S)
    tags = {"stub": "lbdrn_ffi.py", "decode": "base_d = ", "train": "img_d  = ", "weights": "flat = np.concatenate", "jp2": "liblbdrn_jp2.so"}
    out = {}
    for k, needle in tags.items():
        hit = [x for x in b if needle in x.splitlines()[0] or needle in x]
        assert hit, f"INTEGRATION.md lost its {k} example"
        out[k] = hit[0]
    return out


def _stub_namespace():
    src = _blocks()["stub"].replace('"liblbdrn_hip.so"', repr(os.path.join(PKG, "liblbdrn_hip.so")))
    ns = {}
    exec(compile(src, "INTEGRATION.md:stub", "exec"), ns)
    L = ns["L"]
    # "declare argtypes/restype for every entry point used ... the full table is lbdrn_hip/_lib.py:SIGNATURES"
    from lbdrn_hip import _lib
    for name, (res, args) in _lib.SIGNATURES.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = [a if a not in (_lib._GP, _lib._NP) else ctypes.c_void_p for a in args]
    return ns


def test_stub_struct_layout_matches_the_header():
    ns = _stub_namespace()
    from lbdrn_hip import _lib
    assert ctypes.sizeof(ns["Net"]) == ctypes.sizeof(_lib.Net) and ctypes.sizeof(ns["Geom"]) == ctypes.sizeof(_lib.Geom)
    assert [f[0] for f in ns["Net"]._fields_] == [f[0] for f in _lib.Net._fields_]
    assert [f[0] for f in ns["Geom"]._fields_] == [f[0] for f in _lib.Geom._fields_]


@pytest.mark.gpu
def test_decode_and_training_examples_run_and_agree_with_the_codec(dev):
    import torch
    from lbdrn_hip import codec, ops, sampler
    from lbdrn_hip.features import FeatCfg
    from lbdrn_hip.model import LBDRNModel
    from lbdrn_hip.synth import synthetic_tile
    ns = _stub_namespace()
    C, H, W, K, D, bc, nl, bs, epochs = 8, 96, 80, 5, 2, 64, 2, 1024, 3
    img = synthetic_tile(21, C, H, W)
    torch.manual_seed(123)
    model = LBDRNModel(dim_in=C * (2 * D + 1) ** 2, dim_hidden=bc, dim_out=C, num_layers=nl)
    seeds = sampler.draw_pass_seeds(sampler.epoch_plan(epochs, 1))
    steps_per_epoch = (H * W + bs - 1) // bs
    env = dict(ns, torch=torch, np=np, ctypes=ctypes, img=img, C=C, H=H, W=W, K=K, D=D, bc=bc, nl=nl, bs=bs, epochs=epochs,
               model=model, lr=codec.lr_schedule(1e-3, epochs), steps_per_epoch=steps_per_epoch,
               sampler_permutation=lambda e: sampler.permutation(int(seeds[e]), H * W))
    exec(compile(_blocks()["train"], "INTEGRATION.md:train", "exec"), env)
    torch.cuda.synchronize()
    # the codec's own loop on the same draws
    draws = codec.FitDraws(model.flat_parameters(), seeds)
    fit = codec.fit_device(ops.to_device_u16(img, dev), K, D, bc, nl, 1e-3, bs, epochs, cfg=FeatCfg(), draws=draws)
    torch.cuda.synchronize()
    assert torch.equal(env["best_p"], fit.best_params)           # same steps, same bits; the same epoch picked
    assert abs(env["best"] - float(fit.mse_log[:, 0].min().item())) <= 2e-6 * env["best"]
    # decode body on the truncated weights
    params_d = codec.truncate_device(fit.best_params, 16)
    env2 = dict(ns, torch=torch, np=np, ctypes=ctypes, base=(img >> K).astype(np.uint16), C=C, H=H, W=W, K=K, D=D, bc=bc, nl=nl,
                params_d=params_d)
    exec(compile(_blocks()["decode"], "INTEGRATION.md:decode", "exec"), env2)
    rec = codec.apply_device(fit.geom, fit.net, fit.msb, params_d)
    rec = rec[0] if isinstance(rec, tuple) else rec
    assert np.array_equal(env2["image"], rec.cpu().numpy().view(np.uint16))


def test_weight_payload_example_round_trips():
    import torch
    from lbdrn_hip import container
    from lbdrn_hip.model import LBDRNModel
    ns = _stub_namespace()
    torch.manual_seed(3)
    model = LBDRNModel(200, 64, 8, 2)

    class Args:
        precision = 16
    env = dict(ns, torch=torch, np=np, ctypes=ctypes, model=model, args=Args)
    try:
        exec(compile(_blocks()["weights"], "INTEGRATION.md:weights", "exec"), env)
    except RuntimeError as e:     # the payload codec runs its model on the device: no GPU here
        if "device" in str(e).lower() or "gfx950" in str(e).lower():
            pytest.skip("weight payload codec needs the device")
        raise
    back = container.decode_weights(env["compressed_bytes"], expected=env["flat"].size)
    want = container.truncate_precision(env["flat"], 16)
    assert np.array_equal(np.asarray(back, np.float32).view(np.uint32), want.view(np.uint32))
    assert env["compressed_bytes"] == container.encode_weights(env["flat"], 16)      # the bytes encode.py's drop-in writes


def test_jp2_example_round_trips():
    from lbdrn_hip import jp2
    if not jp2.available():
        pytest.skip("liblbdrn_jp2.so not built")
    rng = np.random.default_rng(2)
    C, H, W = 3, 70, 90
    MSB = rng.integers(0, 300, (C, H, W)).astype(np.uint16)
    env = dict(ctypes=ctypes, os=os, np=np, _dir=PKG, MSB=MSB, C=C, H=H, W=W)
    exec(compile(_blocks()["jp2"], "INTEGRATION.md:jp2", "exec"), env)
    assert np.array_equal(env["base"], MSB) and jp2.is_jp2(env["base_jp2"])
    assert np.array_equal(jp2.decode(env["base_jp2"]), MSB)
