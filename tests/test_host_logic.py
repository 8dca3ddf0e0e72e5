"""Host-side logic and the C-ABI surface, no GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    from lbdrn_hip import _lib
    hdr = open(os.path.join(ROOT, "include", "lbdrn_hip.h")).read()
    declared = set(re.findall(r"\b(lbdrn_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = ctypes.CDLL(_lib.lib_path())
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().lbdrn_abi_version() == _lib.ABI_VERSION == 2
    # the constants the Python host mirrors
    consts = {k: int(v, 0) for k, v in re.findall(r"#define (LBDRN_[A-Z_]+) (0x[0-9a-fA-F]+|\d+)\b", hdr)}
    assert (consts["LBDRN_PATH_AUTO"], consts["LBDRN_PATH_GENERIC"], consts["LBDRN_PATH_MFMA"]) == \
        (_lib.PATH_AUTO, _lib.PATH_GENERIC, _lib.PATH_MFMA)
    assert consts["LBDRN_EVAL_BACKGROUND"] == _lib.EVAL_BACKGROUND and _lib.EVAL_BACKGROUND > _lib.PATH_MFMA
    assert consts["LBDRN_EVAL_FAST"] == _lib.EVAL_FAST and not (_lib.EVAL_FAST & (_lib.EVAL_BACKGROUND | 3))


def test_geometry_helpers_without_device():
    from lbdrn_hip import _lib
    L = _lib.lib()
    net = _lib.Net(200, 64, 8, 2)
    assert L.lbdrn_param_count(ctypes.byref(net)) == 17544
    net = _lib.Net(200, 256, 8, 2)
    assert L.lbdrn_param_count(ctypes.byref(net)) == 119304
    g = _lib.Geom(8, 16, 16, 5, 2, 100, 1, 1, 0, 0, None, None)
    assert L.lbdrn_feature_dim(ctypes.byref(g)) == 200
    g.P = 25
    assert L.lbdrn_feature_dim(ctypes.byref(g)) == 250
    # plans of the fused training steps (host arithmetic only): which shapes step in groups, how many features a step
    # multiplies once the always-zero window centres are left out (LBDRNdataset.py:126-128)
    g.P = 0
    rel = lambda bc, nl, F=200: (L.lbdrn_train_step_features(ctypes.byref(g), ctypes.byref(_lib.Net(F, bc, 8, nl))),
                                 L.lbdrn_train_group_size(ctypes.byref(g), ctypes.byref(_lib.Net(F, bc, 8, nl))))
    assert rel(64, 2) == (192, L.lbdrn_train_group_max())      # the headline shape: streamed step, groups
    assert rel(256, 2) == (192, 1)                               # BASELINE configs[2]: the wide step, one fit per launch
    assert rel(64, 3) == (200, 1)                                # three hidden layers: the tile kernel multiplies all of F
    g.relative = 0
    assert rel(64, 2) == (200, L.lbdrn_train_group_max())        # absolute colours: no feature is an exact zero
    g.relative, g.P = 1, 25
    assert rel(64, 2, 250) == (242, L.lbdrn_train_group_max())   # configs[4]: 50 positional + 192 colour features


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_compute_fails_loudly_without_gpu():
    from lbdrn_hip import _lib, ops
    assert _lib.lib().lbdrn_device_check() != 0
    assert b"no CPU path" in _lib.lib().lbdrn_last_error() or b"HIP" in _lib.lib().lbdrn_last_error()
    with pytest.raises(_lib.LbdrnError):
        ops.split_bits(torch.zeros((1, 4, 4), dtype=torch.int16), 5)
    from LBDRNmodel import LBDRNModel
    with pytest.raises(_lib.LbdrnError):
        LBDRNModel(8, 32, 2, 1)(torch.zeros(3, 8))


def test_product_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "lbdrn-msic_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(base, f)).read()
                code = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith(("//", "#", "*", "/*")) or l.lstrip().startswith("#include"))
                if (re.search(r"^\s*(import|from)\s+(oracle|torch_port)\b", code, re.M)
                        or re.search(r"#include.*oracle", code) or re.search(r"[\"']oracle[\"'/]", code)):
                    bad.append(f)
    assert not bad, bad


def test_header_pack_unpack_vs_reference_bytes(golden):
    from lbdrn_hip import container as c
    G = golden["header"]
    for i in range(4):
        a = [int(v) for v in G[f"h{i}/args"]]
        nn, bb = [int(v) for v in G[f"h{i}/nn"]], [int(v) for v in G[f"h{i}/base"]]
        raw = c.pack_header(a[0], a[1], a[2], a[3], a[4], a[5], a[6], nn, bb)
        assert raw == G[f"h{i}/bytes"].tobytes()
        got = c.unpack_header(raw)
        assert list(got[:8]) + got[8] + got[9] == [int(v) for v in G[f"h{i}/parsed"]]
    for bad in (dict(bc=48), dict(K=16), dict(D=16), dict(nl=16), dict(width=65536), dict(split_ratio=6)):
        kw = dict(split_ratio=1, width=8, height=8, K=5, bc=64, nl=2, D=2)
        kw.update(bad)
        t = kw["split_ratio"] ** 2
        with pytest.raises((ValueError, OverflowError)):
            c.pack_header(kw["split_ratio"], kw["width"], kw["height"], kw["K"], kw["bc"], kw["nl"], kw["D"],
                          [1] * t, [1] * t)
    with pytest.raises(OverflowError):
        c.pack_header(1, 8, 8, 5, 64, 2, 2, [1 << 24], [1])
    # the hidden activation rides in ONE extension byte behind the reference's fields, counted by the length byte -- which is
    # all the reference's reader goes by (ref decode.py:26, 182: bitstream[n_bytes_header:]); the default network's header is
    # the reference's, byte for byte (above)
    for i in range(4):
        a = [int(v) for v in G[f"h{i}/args"]]
        nn, bb = [int(v) for v in G[f"h{i}/nn"]], [int(v) for v in G[f"h{i}/base"]]
        ref = G[f"h{i}/bytes"].tobytes()
        assert c.header_activation(ref) is None
        if ref[0] == 255:
            with pytest.raises(OverflowError):
                c.pack_header(a[0], a[1], a[2], a[3], a[4], a[5], a[6], nn, bb, activation="relu")
            continue
        raw = c.pack_header(a[0], a[1], a[2], a[3], a[4], a[5], a[6], nn, bb, activation="relu")
        assert raw == bytes([ref[0] + 1]) + ref[1:] + b"\x01" and c.header_activation(raw) == "relu"
        assert c.unpack_header(raw)[1:] == c.unpack_header(ref)[1:] and c.unpack_header(raw)[0] == len(raw)
        with pytest.raises(ValueError):
            c.header_activation(raw[:-1] + b"\x02")
        assert c.header_activation(raw[:-1] + b"\x00") == "sine"
    with pytest.raises(ValueError):
        c.pack_header(1, 8, 8, 5, 64, 2, 2, [1], [1], activation="tanh")


def test_model_init_and_rng_position_vs_reference(golden):
    from LBDRNmodel import LBDRNModel
    G = golden["init"]
    for name in ("bc64_nl2", "bc16_nl3", "bc32_nl1"):
        F, bc, C, nl = [int(v) for v in G[name + "/dims"]]
        torch.manual_seed(19920517)
        m = LBDRNModel(dim_in=F, dim_hidden=bc, dim_out=C, num_layers=nl)
        assert list(m.state_dict().keys()) == [str(k) for k in G[name + "/keys"]]
        assert np.array_equal(m.flat_parameters().numpy(), G[name + "/params"])
        nxt = [torch.empty((), dtype=torch.int64).random_().item() for _ in range(2)]
        assert nxt == [int(v) for v in G[name + "/next_draws"]]


def test_sampler_replays_dataloader_draw_order():
    """train, eval, train, eval ... iterators of a real DataLoader vs lbdrn_hip.sampler."""
    from torch.utils.data import DataLoader, TensorDataset
    from lbdrn_hip import sampler
    n, bs, epochs = 103, 16, 3
    ds = TensorDataset(torch.arange(n))
    torch.manual_seed(7)
    loader = DataLoader(ds, batch_size=bs, shuffle=True)
    want = []
    for _ in range(epochs):
        want.append(torch.cat([b[0] for b in loader]))   # trainer pass
        for _ in loader:                                  # evaluator pass (order irrelevant)
            pass
    torch.manual_seed(7)
    seeds = sampler.draw_pass_seeds(sampler.epoch_plan(epochs, 1))   # what a fit draws (codec.draw_fit)
    for e in range(1, epochs + 1):
        assert torch.equal(sampler.permutation(seeds[e - 1], n), want[e - 1])
    assert sampler.epoch_plan(1, 1) == [("train", 1)]      # epochs == 1: no evaluation (encode.py:100)
    assert sampler.epoch_plan(4, 2) == [("train", 1), ("train", 2), ("eval", 2), ("train", 3), ("train", 4), ("eval", 4)]
    # after the stream the global generator sits where the DataLoader run left it
    a = torch.empty((), dtype=torch.int64).random_().item()
    torch.manual_seed(7)
    for _ in range(epochs):
        for _ in loader:
            pass
        for _ in loader:
            pass
    assert a == torch.empty((), dtype=torch.int64).random_().item()


def test_lr_schedule_equals_torch_steplr():
    from lbdrn_hip.codec import lr_schedule
    for epochs in (1, 2, 3, 10, 11):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1e-3)
        sch = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(epochs / 3)), gamma=0.1)
        want = []
        for _ in range(epochs):
            want.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        assert lr_schedule(1e-3, epochs) == want


def test_positional_tables_equal_reference_features(golden):
    from lbdrn_hip.features import FeatCfg, pos_tables
    G = golden["features"]
    f = G["G_embed/features"].reshape(12, 10, 250)
    rt, ct = pos_tables(12, 10, FeatCfg(use_coordinates=True, embedding=True))
    assert np.array_equal(rt, f[:, 0, :25]) and np.array_equal(ct, f[0, :, 25:50])
    f = G["F_coords/features"].reshape(12, 10, 202)
    rt, ct = pos_tables(12, 10, FeatCfg(use_coordinates=True))
    assert np.array_equal(rt[:, 0], f[:, 0, 0]) and np.array_equal(ct[:, 0], f[0, :, 1])


def test_weight_payload_value_map_on_every_class_of_float():
    """truncate_precision (the lossy value map of the weight payload, ref encode.py:129 precision=16) on negative
    numbers, both zeros, denormals, infinities and NaNs, and the same map through the fpzip-syntax stream of the C
    ABI (lbdrn_weights_encode / _decode) and through the independent Python restatement (oracle/fpz_port.py):
    identical bits from all three, identical stream bytes from the two coders."""
    import fpz_port as FP
    from lbdrn_hip import container as c
    special = np.array([0x00000000, 0x80000000,              # +0, -0
                        0x00000001, 0x80000001, 0x007FFFFF,  # denormals
                        0x00800000, 0x3F800000, 0xBF800000,  # smallest normal, +-1
                        0x3F80FFFF, 0xBF80FFFF, 0x3F810000,  # just below / at a kept-bit boundary
                        0x7F7FFFFF, 0xFF7FFFFF,              # +-max
                        0x7F800000, 0xFF800000,              # +-inf
                        0x7FC00000, 0xFFC00000, 0x7F800001, 0x7FBFFFFF,   # quiet NaN, NaNs with low payloads
                        0x12345678, 0x87654321], np.uint32)
    w = special.view(np.float32)
    t16 = c.truncate_precision(w, 16).view(np.uint32)
    assert np.array_equal(t16, special & np.uint32(0xFFFF0000))          # sign-magnitude truncation, every class
    assert t16[0] == 0 and t16[1] == 0x80000000                            # the sign of zero survives
    assert t16[2] == 0 and t16[4] == 0x007F0000                            # denormals: truncated like anything else
    assert t16[17] == 0x7F800000 and np.isinf(t16[17:18].view(np.float32))[0]   # NaN payload in dropped bits -> inf
    assert np.isnan(t16[15:17].view(np.float32)).all()                      # quiet NaNs stay NaN
    rng = np.random.default_rng(5)
    big = np.concatenate([w, rng.normal(0, 0.05, 5000).astype(np.float32), np.zeros(40, np.float32)])
    for prec in (16, 9, 12, 24, 32, 0):
        stream = c.encode_weights(big, prec)
        bits = big.view(np.uint32)
        expect = c.truncate_precision(big, prec).view(np.uint32)
        assert stream == FP.compress(bits.tolist(), prec), prec               # two independent coders, same bytes
        assert np.array_equal(c.decode_weights(stream).view(np.uint32), expect), prec
        got, p = FP.decompress(stream)
        assert p == (prec or 32) and np.array_equal(np.array(got, np.uint32), expect), prec
        assert np.array_equal(np.array([FP.truncate_bits(int(b), prec or 32) for b in bits], np.uint32), expect)
    # foreign and damaged streams are refused, not decoded into garbage lengths
    from lbdrn_hip import _lib
    good = c.encode_weights(big, 16)
    for bad in (b"", b"JUNKJUNKJUNK", good[:6]):
        with pytest.raises(_lib.LbdrnError):
            c.decode_weights(bad)
    with pytest.raises(_lib.LbdrnError):
        c.decode_weights(good[:len(good) // 2])
    assert c.decode_weights(c.encode_weights(np.zeros(0, np.float32), 16)).size == 0
    # the count in the stream is checked against what the network needs before anything is allocated by it
    with pytest.raises(_lib.LbdrnError):
        c.decode_weights(good, expected=big.size + 1)
    assert c.decode_weights(good, expected=big.size).size == big.size
    # bytes behind the coder's flush are not part of a stream
    with pytest.raises(_lib.LbdrnError):
        c.decode_weights(good + b"\0\0")
    # fpzip codes precisions <= 8 with a narrow residual coder that is not restated here: refused on both sides
    for prec in (2, 8):
        with pytest.raises(_lib.LbdrnError):
            c.encode_weights(big, prec)


@pytest.mark.skipif(__import__("importlib").util.find_spec("fpzip") is None, reason="fpzip is not installed here")
def test_weight_payload_is_fpzips_own_bytes_where_fpzip_exists():
    """Runs only on a box that has the fpzip module (none in this image: parity unpinned until it does): the
    restated coder and fpzip must write the same bytes and read each other's streams."""
    import fpzip
    from lbdrn_hip import container as c
    rng = np.random.default_rng(11)
    w = np.concatenate([rng.normal(0, 0.05, 17544), np.zeros(64)]).astype(np.float32)
    for prec in (9, 16, 24, 32):
        ours = c._native_encode(w, prec)
        theirs = fpzip.compress(w, precision=prec, order="C")
        assert ours == theirs, prec
        want = c.truncate_precision(w, prec).view(np.uint32)
        assert np.array_equal(np.asarray(fpzip.decompress(ours, order="C"), np.float32).reshape(-1).view(np.uint32), want)
        assert np.array_equal(c._native_decode(theirs, w.size).view(np.uint32), want)


def test_payload_round_trips_and_precision_model():
    from lbdrn_hip import container as c
    rng = np.random.default_rng(0)
    w = rng.normal(0, 0.1, 17544).astype(np.float32)
    for prec in (16, 20, 32, 9):
        q = c.decode_weights(c.encode_weights(w, prec))
        assert np.array_equal(q.view(np.uint32), c.truncate_precision(w, prec).view(np.uint32))
    assert np.all(c.truncate_precision(w, 16).view(np.uint32) & 0xFFFF == 0)
    for dt, hi in ((np.uint16, 2000), (np.uint8, 255)):
        x = rng.integers(0, hi, (3, 37, 41)).astype(dt)
        y = c.decode_base(c.encode_base(x, codec="LBB1"))   # the host codec; LBB2 is covered by the GPU suite
        assert y.dtype == dt and np.array_equal(x, y)
    m = __import__("LBDRNmodel").LBDRNModel(10, 8, 2, 2)
    flat = c.flatten_state(m.state_dict())
    sd = c.unflatten_state(flat, m.state_dict())
    assert all(torch.equal(sd[k], v) for k, v in m.state_dict().items())


def test_precision_the_coder_refuses_is_refused_before_any_fit(tmp_path, monkeypatch):
    """`-prec 8` used to spend the whole fit and then raise in encode_weights.  One shared range
    (container.WEIGHT_PRECISIONS) for the CLI parsers and the coder; the parsers stop before any image is opened."""
    from lbdrn_hip import container as c
    import encode
    import sweep
    monkeypatch.delenv("LBDRN_WEIGHTS_CODEC", raising=False)
    for good in (0, 9, 16, 32):
        assert c.check_weight_precision(good) == good
    for bad in (-1, 1, 2, 8, 33):
        with pytest.raises(ValueError):
            c.check_weight_precision(bad)
        with pytest.raises((ValueError, __import__("lbdrn_hip")._lib.LbdrnError)):   # the coder's own refusal: same set
            c.encode_weights(np.zeros(4, np.float32), bad)
    missing = str(tmp_path / "never_opened.tif")          # the parser must stop before it looks at the image
    with pytest.raises(SystemExit) as e:
        encode.main(["-i", missing, "-o", str(tmp_path), "-prec", "8"], shard_tiles=False)
    assert e.value.code == 2
    with pytest.raises(SystemExit):
        sweep.parse(["--images", missing, "-prec", "5"])
    assert not any(tmp_path.iterdir())                     # no output directory, no log


def test_raster_io_round_trip(tmp_path):
    from lbdrn_hip import raster_io as r
    rng = np.random.default_rng(1)
    for C, dt in ((8, np.uint16), (1, np.uint8), (3, np.float32)):
        a = rng.uniform(0, 250, (C, 13, 17)).astype(dt)
        p = str(tmp_path / f"t{C}.tif")
        r.write_raster(p, a)
        b = r.read_raster(p)
        assert np.array_equal(b.reshape(a.shape), a) and b.dtype == a.dtype
    with pytest.raises(ValueError):
        r.write_raster(str(tmp_path / "x.tif"), np.zeros((1, 2, 2), np.int64))


def test_tile_windows_cover_image():
    from LBDRNdataset import tile_windows
    cov = np.zeros((31, 50), int)
    for i, j, x0, y0, w, h in tile_windows(50, 31, 3):
        cov[y0:y0 + h, x0:x0 + w] += 1
    assert (cov == 1).all()


def test_torch_port_forms_agree_and_fit_something():
    """oracle/torch_port.py: DataLoader form and index_select form walk the same RNG stream."""
    import torch_port as TP
    from lbdrn_hip.synth import synthetic_tile
    img = synthetic_tile(3, 4, 24, 32)
    torch.manual_seed(19920517)
    a = TP.fit(img, 5, 1, 16, 2, 1e-3, 128, 3, faithful=True)
    torch.manual_seed(19920517)
    b = TP.fit(img, 5, 1, 16, 2, 1e-3, 128, 3, faithful=False)
    assert np.array_equal(a["params"], b["params"]) and a["best_epoch"] == b["best_epoch"]
    rec = TP.apply(a["msb"], a["params"], 5, 1, 16, 2)
    assert np.array_equal(rec >> 5, img >> 5)
    import oracle as O
    rec_o = O.decode(a["msb"], 5, 1, O.FeatCfg(), a["params"], 16, 2)
    assert (rec_o != rec).mean() < 1e-3   # canonical arithmetic vs torch: boundary flips only


def test_tile_windows_equal_the_reference_split_and_merge(golden):
    """The windows the reference's split_image cuts and merge_tiles pastes (tests/golden/make_golden_tiles.py
    ran them with a recording gdal stand-in) for six sizes / split ratios, last row and column ragged."""
    from LBDRNdataset import tile_windows
    G = golden["tiles"]
    assert len(G.files) == 6
    for key in G.files:
        w, h, sr = (int(x[1:]) if x[0] in "wh" else int(x[2:]) for x in key.split("_"))
        assert [tuple(int(v) for v in row) for row in G[key]] == list(tile_windows(w, h, sr)), key


def test_results_summary_parser_equals_the_reference(tmp_path):
    """extract_metrics on four log files against what the reference's results_summary.extract_metrics returned
    for the same texts (tests/golden/make_golden_summary.py): last record wins, integers without a decimal
    point are not matched, missing records stay missing."""
    import json
    import results_summary as R
    with open(os.path.join(ROOT, "tests", "golden", "summary.json")) as f:
        cases = json.load(f)
    assert len(cases) == 4
    for name, case in cases.items():
        path = tmp_path / f"{name}.txt"
        path.write_text(case["log"])
        assert R.extract_metrics(str(path)) == case["metrics"], name


def test_constants_and_log_record_layout_equal_the_reference(tmp_path):
    """constants.py switches and the '[timestamp] message' layout of a logged record, against what the
    reference's own constants.py / logger.py produced (tests/golden/make_golden_misc.py)."""
    import json
    import re
    import constants as C
    import logger as L
    with open(os.path.join(ROOT, "tests", "golden", "misc.json")) as f:
        G = json.load(f)
    mine = {k: getattr(C, k) for k in dir(C) if k.isupper()}
    # one name the reference does not have: it stands for the reference's "edit this source line" switch of the hidden
    # activation (ref encode.py:75, decode.py:108), and its default is the reference's default
    assert mine.pop("HIDDEN_ACTIVATION") == "sine"
    assert mine == G["constants"]
    L.create_logger(str(tmp_path / "sub" / "dir"), "run.txt", log_file_only=True)
    L.log.info("MSE: 1.25")
    L.log.info(f"Total size: {1234} bytes, bpsp={0.5}")
    L.destroy_logger()
    lines = (tmp_path / "sub" / "dir" / "run.txt").read_text().splitlines()
    layout = [re.sub(r"\d", "d", re.match(r"^\[[^\]]*\]", ln).group(0)) + ln[ln.index("]") + 1:] for ln in lines]
    assert layout == G["log_layout"]


def _mt19937_words(seed, total):
    """x[0..total): the seeded state of at::mt19937(seed), then the raw (untempered) recurrence -- plain numpy, three
    slices per regeneration (the recurrence reaches back 227 words)."""
    n, m = 624, 397
    x = np.zeros(((total + n - 1) // n + 1) * n, np.uint32)
    v = seed & 0xFFFFFFFF
    x[0] = v
    for k in range(1, n):
        v = (1812433253 * (v ^ (v >> 30)) + k) & 0xFFFFFFFF
        x[k] = v
    for base in range(0, len(x) - n, n):
        for lo, hi in ((0, 227), (227, 454), (454, 624)):
            a, b, c = x[base + lo:base + hi], x[base + lo + 1:base + hi + 1], x[base + lo + m:base + hi + m]
            y = (a & np.uint32(0x80000000)) | (b & np.uint32(0x7FFFFFFF))
            x[base + n + lo:base + n + hi] = c ^ (y >> np.uint32(1)) ^ np.where(b & np.uint32(1), np.uint32(0x9908B0DF), np.uint32(0))
    return x[:total]


def test_mt19937_jump_polynomials_reproduce_the_sequence():
    """lbdrn_randperm generates a long permutation's MT19937 words as segments side by side; segment s starts from
    x[J + k] = XOR over the set bits i of g_s of x[i + k], J = s * words_per_segment (csrc/mt_jump.inc).  Host-side
    arithmetic only: checked here against a plain MT19937 for the first, second and last polynomial."""
    from lbdrn_hip import _lib
    L = ctypes.CDLL(_lib.lib_path())
    L.lbdrn_mt19937_jump_poly.restype = ctypes.c_int64
    L.lbdrn_mt19937_jump_poly.argtypes = [ctypes.c_int32, ctypes.c_void_p]
    g = np.zeros(624, np.uint32)
    seg = L.lbdrn_mt19937_jump_poly(1, g.ctypes.data)
    assert seg > 0 and seg % 624 == 0
    assert L.lbdrn_mt19937_jump_poly(0, g.ctypes.data) < 0 and L.lbdrn_mt19937_jump_poly(32, g.ctypes.data) < 0
    x = _mt19937_words(19920517, 31 * seg + 624)
    rs = np.random.RandomState(19920517)   # mt19937ar's init_genrand, as at::mt19937: a check of the helper, not of the library
    assert np.array_equal(rs.get_state()[1], x[:624])
    def temper(y):
        y = int(y); y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680; y ^= (y << 15) & 0xEFC60000; y ^= y >> 18
        return y
    assert [temper(v) for v in x[624:628]] == [int(v) for v in rs.randint(0, 2**32, size=4, dtype=np.uint32)]
    for s in (1, 2, 31):
        assert L.lbdrn_mt19937_jump_poly(s, g.ctypes.data) == seg
        idx = np.nonzero(np.unpackbits(g.view(np.uint8), bitorder="little"))[0]
        assert 1000 < len(idx) and idx[-1] < 19937
        for k in (1, 2, 311, 623):
            assert np.bitwise_xor.reduce(x[idx + k]) == x[s * seg + k], (s, k)
        assert np.bitwise_xor.reduce(x[idx]) >> 31 == x[s * seg] >> 31   # word 0 of a state window: its top bit is all the recurrence reads


def test_fits_in_flight_are_sized_against_the_free_memory(monkeypatch):
    """codec.fit_many no longer takes `in_flight` on faith: the count is cut to what 85 % of the
    free device memory holds of the largest fit; a fit that does not fit at all raises with the numbers; a process that
    shares its GPU (run.sh with PER_GPU > 1: LBDRN_DEVICE_SHARED) sizes itself against half."""
    import torch
    from lbdrn_hip import codec, ops

    class T:                                      # (only .shape and .device are looked at)
        def __init__(self, *shape):
            self.shape, self.device = shape, "cuda:0"
    one = codec.fit_bytes(8, 2048, 2048, 5, 2, 64, 2, 8192, 10)
    big = codec.fit_bytes(8, 6000, 6000, 5, 2, 64, 2, 8192, 10)
    assert 3.9 * 2**30 < one < 5.5 * 2**30 and 31 * 2**30 < big < 36 * 2**30     # row matrix 832 B a pixel + permutations + planes
    assert codec.fit_bytes(8, 2048, 2048, 5, 2, 256, 2, 8192, 10) > one - 2**30  # (the wide step: 848 B rows, hand-over buffers)
    monkeypatch.delenv("LBDRN_DEVICE_SHARED", raising=False)
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda dev=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda dev=None: 0)
    args = (5, 2, 64, 2, 8192, 10)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (280 << 30, 288 << 30))
    assert codec.memory_limited_in_flight([T(8, 2048, 2048)] * 8, 4, *args) == 4
    assert codec.memory_limited_in_flight([T(8, 6000, 6000)] * 4, 4, *args) == 4          # 4 x 34 GiB fit on a 288 GB part
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (100 << 30, 288 << 30))
    assert codec.memory_limited_in_flight([T(8, 6000, 6000)] * 4, 4, *args) == 2
    assert codec.memory_limited_in_flight([T(8, 2000, 2000), T(8, 6000, 6000)], 4, *args) == 2   # the largest fit decides
    monkeypatch.setenv("LBDRN_DEVICE_SHARED", "1")
    assert codec.memory_limited_in_flight([T(8, 6000, 6000)] * 4, 4, *args) == 1
    assert codec.device_shared()
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (20 << 30, 288 << 30))
    with pytest.raises(ops._lib.LbdrnError, match="split the image"):
        codec.memory_limited_in_flight([T(8, 6000, 6000)], 4, *args)
    # what fit_many takes when the caller names no count: two chains of pair launches at bc = 64, three chains of the
    # three-launch step at bc >= 128 (a fourth only queues: DESIGN 4.5)
    assert codec.default_in_flight(8, 2048, 2048, 5, 2, 64, 2) == 4
    assert codec.default_in_flight(8, 2048, 2048, 5, 2, 256, 2) == codec.default_in_flight(8, 2048, 2048, 5, 2, 128, 1) == 3


def test_hidden_activation_switch_host_side():
    """lbdrn_net.act on the host side: FeatCfg / constants.HIDDEN_ACTIVATION -> the descriptor; the drop-in model maps
    torch.nn.ReLU() (ref encode.py:75) to it and initialises exactly as the default model does (the activation owns no
    parameter and draws nothing: ref LBDRNmodel.py:32-37)."""
    import ctypes
    import subprocess
    import tempfile
    import torch
    from lbdrn_hip import _lib, ops
    from lbdrn_hip.features import FeatCfg
    from lbdrn_hip.model import LBDRNModel
    assert FeatCfg().act == 0 and FeatCfg(activation="relu").act == 1 and FeatCfg.from_constants().act == 0
    with pytest.raises(ValueError):
        FeatCfg(activation="tanh")
    assert ops.make_net(200, 64, 8, 2).act == ops.ACT_SINE and ops.make_net(200, 64, 8, 2, ops.ACT_RELU).act == 1
    # the ctypes mirror has the header's size and field offsets (gcc on include/lbdrn_hip.h)
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        with open(src, "w") as f:
            f.write('#include <stdio.h>\n#include <stddef.h>\n#include "lbdrn_hip.h"\nint main(void){printf("%zu %zu %d %d %d",'
                    'sizeof(lbdrn_net), offsetof(lbdrn_net, act), LBDRN_ACT_SINE, LBDRN_ACT_RELU, LBDRN_ABI_VERSION);return 0;}\n')
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(d, "s")])
        got = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    assert [int(x) for x in got] == [ctypes.sizeof(_lib.Net), _lib.Net.act.offset, ops.ACT_SINE, ops.ACT_RELU, _lib.ABI_VERSION]
    torch.manual_seed(5)
    a = LBDRNModel(200, 64, 8, 2)
    ra = torch.rand(1)
    torch.manual_seed(5)
    b = LBDRNModel(200, 64, 8, 2, activation=torch.nn.ReLU())
    rb = torch.rand(1)
    assert torch.equal(a.flat_parameters(), b.flat_parameters()) and torch.equal(ra, rb)
    assert a.hip_net().act == ops.ACT_SINE and b.hip_net().act == ops.ACT_RELU
    assert not LBDRNModel(200, 64, 8, 2, activation=torch.nn.Tanh())._fused_ok
    assert not LBDRNModel(200, 64, 8, 2, activation=torch.nn.ReLU(), final_activation=torch.nn.Identity())._fused_ok


def test_lone_fit_schedule_calibration_logic():
    """codec._calibrated_head (host logic only: stand-in events): the model's guess until a fit's events have completed, then
    head x t_pass / t_head kept within [guess / 2, 2 guess] and within the epoch; events that have not completed yet are left for
    a later fit (never waited for)."""
    from lbdrn_hip import codec

    class Ev:
        def __init__(self, t, done=True):
            self.t, self.done = t, done

        def query(self):
            return self.done

        def elapsed_time(self, other):
            return other.t - self.t

    key = ("test", 1)
    codec._HEAD_MEASURED.pop(key, None)
    codec._HEAD_PENDING.pop(key, None)
    assert codec._calibrated_head(key, 137, 512) == 137                     # nothing measured: the guess
    codec._HEAD_PENDING[key] = ((Ev(0.0), Ev(3.0, done=False), Ev(0.0), Ev(2.0)), 137)
    assert codec._calibrated_head(key, 137, 512) == 137 and key in codec._HEAD_PENDING     # pass still running: not waited for
    codec._HEAD_PENDING[key] = ((Ev(0.0), Ev(3.0), Ev(0.0), Ev(2.0)), 100)  # the pass lasted 1.5 x the head
    assert codec._calibrated_head(key, 137, 512) == 150 and key not in codec._HEAD_PENDING
    codec._HEAD_PENDING[key] = ((Ev(0.0), Ev(30.0), Ev(0.0), Ev(1.0)), 150)  # absurd ratio: clamped to twice the guess
    assert codec._calibrated_head(key, 137, 512) == 274
    codec._HEAD_PENDING[key] = ((Ev(0.0), Ev(0.1), Ev(0.0), Ev(9.0)), 274)   # ... and to half of it
    assert codec._calibrated_head(key, 137, 512) == 68
    assert codec._calibrated_head(key, 137, 40) == 40                        # never more than the epoch has
    assert codec.head_calibration()[key] == 68
    codec._HEAD_MEASURED.pop(key, None)


def test_base_payloads_coded_ahead_equal_the_ones_coded_afterwards():
    """encode.BasePayloadsAhead (host threads, no GPU): the JPEG 2000 payloads of an image's tiles coded on a background thread
    are the bytes container.encode_base gives for tile >> K with the reference's dtype rule (LBDRNdataset.py:100); an error in
    the thread is raised where the payload is asked for."""
    from lbdrn_hip import container, jp2
    if not jp2.available():
        pytest.skip("liblbdrn_jp2.so not built (OpenJPEG absent)")
    import encode
    rng = np.random.default_rng(5)
    tiles = [rng.integers(0, 10000, (4, 40, 56)).astype(np.uint16), rng.integers(0, 6000, (2, 33, 21)).astype(np.uint16)]
    ahead = encode.BasePayloadsAhead(tiles, 5)
    for k in (1, 0):
        msb = tiles[k] >> 5
        msb = msb.astype(np.uint8) if int(msb.max()) <= 255 else msb
        want = container.encode_base(msb, codec="jp2")
        got = ahead.take(k)
        assert got == want and np.array_equal(container.decode_base(got), msb) and container.decode_base(got).dtype == msb.dtype
    bad = encode.BasePayloadsAhead([np.zeros((2, 3), np.float32)], 5)       # (not a raster of integers: the shift raises in the thread)
    with pytest.raises(Exception):
        bad.take(0)
