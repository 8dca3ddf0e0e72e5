"""The fused MFMA training step against the generic kernels and the oracle over the shapes it
supports, end-to-end encode/decode through the CLIs, and full-size properties."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle as O
from lbdrn_hip import codec, container, ops, raster_io
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.synth import synthetic_tile

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN, MFMA = ops._lib.PATH_GENERIC, ops._lib.PATH_MFMA


def _params(rng, F, bc, C, nl):
    parts = []
    for l in range(nl):
        nin = F if l == 0 else bc
        b = 1.0 / nin if l == 0 else np.sqrt(6.0 / nin) / 30.0
        parts += [rng.uniform(-b, b, bc * nin), rng.uniform(-b, b, bc)]
    b = np.sqrt(6.0 / bc) / 30.0
    parts += [rng.uniform(-b, b, C * bc), rng.uniform(-b, b, C)]
    return np.concatenate(parts).astype(np.float32)


CASES = [
    # C, H, W, K, D, nl, bs, flags(coords, embed, colors, relative)
    (8, 40, 52, 5, 2, 2, 512, (0, 0, 1, 1)),      # F=200: the north-star shape (192 features multiplied, LQ=48), short last batch
    (8, 40, 52, 5, 2, 2, 512, (0, 0, 1, 0)),      # F=200 with absolute colours: no feature is an exact zero, LQ=52
    (4, 30, 41, 5, 2, 1, 300, (0, 0, 1, 1)),      # F=100 (LQ=24), one hidden layer, ragged workgroups
    (4, 30, 41, 5, 2, 2, 300, (0, 0, 1, 1)),      # F=100, two hidden layers: the reference's 4-band shape (96 features multiplied, six strips)
    (5, 30, 41, 5, 2, 2, 300, (0, 0, 1, 1)),      # F=125: 120 features multiplied (LQ=32, eight strips) -- the loop schedule at two hidden layers
    (3, 25, 33, 3, 1, 3, 256, (0, 0, 1, 1)),      # F=27 (LQ=16), three hidden layers
    (16, 20, 24, 6, 0, 2, 128, (0, 0, 1, 1)),     # D=0: F=C=16, all 16 output slots used
    (8, 24, 36, 5, 2, 2, 400, (1, 1, 1, 1)),      # F=250 (LQ=64): positional embedding, config 5
    (1, 31, 29, 4, 3, 2, 200, (1, 0, 1, 0)),      # one band, coords without embedding, absolute colours
    (8, 40, 52, 5, 2, 2, 512, (0, 0, 1, 1), 256), # bc=256 (BASELINE configs[2]): the wide wave-local kernel, NT=16
    (4, 30, 41, 5, 2, 1, 300, (0, 0, 1, 1), 128), # bc=128, one hidden layer (NT=8), ragged workgroups
    (8, 24, 36, 5, 2, 2, 400, (1, 1, 1, 1), 256), # bc=256 with the positional embedding (F=250, LQ=64)
]


@pytest.mark.parametrize("alone", (False, True))
@pytest.mark.parametrize("case", CASES)
def test_mfma_epoch_matches_generic_and_oracle(dev, case, alone):
    """(alone: the LBDRN_TRAIN_ALONE hint on the fused path -- k_train_split where the shape has it, held to the oracle
    directly; the shapes with one kernel ignore it.)"""
    C, H, W, K, D, nl, bs, flags = case[:8]
    bc = case[8] if len(case) > 8 else 64
    rng = np.random.default_rng(sum(case[:7]))
    cfg = FeatCfg(bool(flags[0]), bool(flags[1]), 1.4, 12, bool(flags[2]), bool(flags[3]))
    ocfg = O.FeatCfg(cfg.use_coordinates, cfg.embedding, 1.4, 12, cfg.use_colors, cfg.relative)
    img = synthetic_tile(int(rng.integers(100)), C, H, W)
    msb, lab, mx = O.split_bits(img, K)
    F = cfg.feature_dim(C, D)
    p0 = _params(rng, F, bc, C, nl)
    perm_np = rng.permutation(H * W).astype(np.int64)
    perm = torch.from_numpy(perm_np).to(dev)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
    net = ops.make_net(F, bc, C, nl)
    img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
    nsteps = (H * W + bs - 1) // bs
    res = {}
    for path in (GEN, MFMA):
        p = torch.from_numpy(p0.copy()).to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        losses = torch.zeros(nsteps, dtype=torch.float32, device=dev)
        ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 3, 1e-3, losses, path=path, alone=alone and path == MFMA)
        res[path] = (p.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy(), losses.cpu().numpy())
    # oracle, step by step
    feats = O.features(msb, D, ocfg, mx)
    po, mo, vo = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    lo = []
    for s in range(nsteps):
        b = perm_np[s * bs:(s + 1) * bs]
        l, _ = O.train_step(po, mo, vo, F, bc, C, nl, feats[b], lab[b], 1e-3, 3 + s + 1)
        lo.append(l)
    for path in (GEN, MFMA):
        p, m, v, losses = res[path]
        np.testing.assert_allclose(losses, np.array(lo), rtol=1e-5), path   # north_star tolerance
        # bc > 64: two float32 evaluations of these steps differ by that much -- torch's own run of the bc = 256 fixture sits
        # 5.3e-5 / 6.8e-5 of the largest moment away from float64 (test_wide_train_kernel_matches_reference_fixture measures
        # both sides against float64); the oracle here is float32 too
        # (the same bound as there: twice the reference's own 6.8e-5 = 7 x 2e-5)
        wide = 7.0 if bc > 64 else 1.0
        # absolute colours at F = 200: every feature is ~0.5 and the moments are small (largest 6e-4), so the fused
        # step's 4.5e-7 per activation (hardware sin / exp, DESIGN "Two arithmetics") shows as 7.5e-5 of the largest one
        if F >= 200 and not cfg.relative:
            wide = 8.0
        assert np.linalg.norm(p - po) <= 2e-5 * np.linalg.norm(po), path
        assert np.abs(m - mo).max() <= wide * 2e-5 * np.abs(mo).max(), path
        assert np.abs(v - vo).max() <= wide * 5e-5 * np.abs(vo).max(), path


def test_window_centre_columns_take_no_part_in_a_fit(dev):
    """With RELATIVE and D > 0 the window centre minus itself is an exact 0.0f (LBDRNdataset.py:126-128): the fused step
    leaves those C features out of its products (lbdrn_train_step_features = F - C).  Their columns of W_0 must come
    out of an epoch as they went in -- what the reference's Adam does with a gradient that has always been 0 -- with zero
    moments, on the generic path (which multiplies them) and on the fused one (which does not), and as in the oracle."""
    C, H, W, K, D, nl, bc, bs = 8, 40, 52, 5, 2, 2, 64, 512
    rng = np.random.default_rng(7)
    cfg = FeatCfg(False, False, 1.4, 12, True, True)
    img = synthetic_tile(3, C, H, W)
    msb, lab, mx = O.split_bits(img, K)
    F = cfg.feature_dim(C, D)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
    net = ops.make_net(F, bc, C, nl)
    assert ops.train_step_features(geom, net) == F - C
    absolute = ops.FeatureGeometry(C, H, W, K, D, mx, FeatCfg(False, False, 1.4, 12, True, False), dev)
    assert ops.train_step_features(absolute, net) == F
    assert ops.train_group_size(C, H, W, K, D, cfg, bc, nl) == ops.train_group_max()
    assert ops.train_group_size(C, H, W, K, D, cfg, 256, nl) == 1          # the wide step takes one fit per launch
    side = 2 * D + 1
    centre = np.array([c * side * side + D * side + D for c in range(C)])
    feats = O.features(msb, D, O.FeatCfg(False, False, 1.4, 12, True, True), mx)
    assert not feats[:, centre].any()                                        # exact zeros in the oracle's matrix too
    p0 = _params(rng, F, bc, C, nl)
    cols = (np.arange(bc)[:, None] * F + centre[None, :]).ravel()            # W_0[n][centre]
    perm = torch.from_numpy(rng.permutation(H * W).astype(np.int64)).to(dev)
    img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
    for path in (GEN, MFMA):
        p = torch.from_numpy(p0.copy()).to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 0, 1e-3, None, path=path)
        pn, mn, vn = p.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()
        assert np.array_equal(pn[cols].view(np.int32), p0[cols].view(np.int32)), path
        assert not mn[cols].any() and not vn[cols].any(), path
        assert np.abs(pn - p0).max() > 0                                     # (the rest did train)


def test_alone_hint_changes_no_number(dev):
    """LBDRN_TRAIN_ALONE (lbdrn_hip.h) is a performance hint: the epoch leaves the same parameters, moments and losses bit
    for bit with and without it, on every shape of step.  Since round 5 that is a statement about TWO kernels: the shapes
    BASELINE.json names (F = 200 and the embedding's F = 250 at bc = 64, two hidden layers) step on k_train_split -- 256
    workgroups of 32 rows, units halved between two waves, two gradient slabs per 64-row group -- when the caller says the
    fit has the device to itself, and on k_train_stream (128 workgroups of 64 rows) otherwise; both follow one summation
    tree (train_split.inc), so which of them ran must not show in a single bit.  Ragged last minibatches (a group with one
    parity half-empty, a minibatch shorter than a workgroup), odd batch sizes; the shapes that have one kernel only (one
    hidden layer, bc = 256, the generic path) ignore the hint; the group call refuses nothing when the flag rides on a
    group of one."""
    rng = np.random.default_rng(11)
    plain, embed = FeatCfg(False, False, 1.4, 12, True, True), FeatCfg(True, True, 1.4, 12, True, True)
    for (C, H, W, K, D, nl, bc, bs, cfg) in [(8, 70, 90, 5, 2, 2, 64, 512, plain), (3, 33, 47, 4, 1, 1, 64, 256, plain),
                                             (8, 40, 52, 5, 2, 2, 256, 512, plain), (8, 24, 20, 5, 2, 2, 64, 96, embed),
                                             (8, 30, 31, 5, 2, 2, 64, 77, plain), (8, 9, 11, 5, 2, 2, 64, 8192, plain),
                                             (4, 70, 90, 5, 2, 2, 64, 512, plain), (4, 30, 31, 5, 2, 2, 64, 77, plain),   # the 4-band shape: k_train_split<24, 6>
                                             (4, 13, 5, 5, 2, 2, 64, 64, plain)]:                                         # ... 65 rows: a tail minibatch of ONE row
        img = synthetic_tile(5, C, H, W)
        msb, lab, mx = O.split_bits(img, K)
        F = cfg.feature_dim(C, D)
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        net = ops.make_net(F, bc, C, nl)
        p0 = _params(rng, F, bc, C, nl)
        perm = torch.from_numpy(rng.permutation(H * W).astype(np.int64)).to(dev)
        img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
        steps = (H * W + bs - 1) // bs
        for path in (GEN, MFMA):
            got = []
            for alone in (False, True):
                p = torch.from_numpy(p0.copy()).to(dev)
                m, v = torch.zeros_like(p), torch.zeros_like(p)
                losses = torch.zeros(steps, dtype=torch.float32, device=dev)
                ws = ops.TrainWorkspace(geom, net, bs, dev).prepare(img_d, msb_d, path)
                for e in range(2):
                    ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, e * steps, 1e-3, losses, path=path, ws=ws, alone=alone)
                got.append([t.cpu().numpy().view(np.int32) for t in (p, m, v, losses)])
            for a, b in zip(*got):
                assert np.array_equal(a, b), (bc, F, bs, path)
            assert np.isfinite(got[0][0].view(np.float32)).all() and np.abs(got[0][1].view(np.float32)).max() > 0


def test_mfma_train_rejects_unsupported_shapes(dev):
    img = synthetic_tile(1, 4, 16, 16)
    msb, _, mx = O.split_bits(img, 5)
    geom = ops.FeatureGeometry(4, 16, 16, 5, 2, mx, FeatCfg(), dev)
    net = ops.make_net(100, 32, 4, 2)    # bc=32: generic only (the fused steps take bc = 64, 128, 256)
    ws = ops.TrainWorkspace(geom, net, 64, dev)
    with pytest.raises(ops._lib.LbdrnError):
        ws.prepare(ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev), MFMA)
    # the generic path takes it
    p = torch.from_numpy(_params(np.random.default_rng(0), 100, 32, 4, 2)).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    perm = torch.randperm(256, device=dev)
    ops.train_epoch(geom, net, ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev), perm, 64, p, m, v, 0, 1e-3)
    assert torch.isfinite(p).all()


def test_fit_is_independent_of_path_choice_and_reproducible(dev):
    """Whole fit (3 epochs, eval + best-epoch selection) twice on the default path: identical bits;
    the generic path lands within training tolerance and picks the same best epoch."""
    img = synthetic_tile(5, 8, 48, 64)
    outs = []
    for path in (ops._lib.PATH_AUTO, ops._lib.PATH_AUTO, GEN):
        torch.manual_seed(19920517)
        fit = codec.fit_device(ops.to_device_u16(img, dev), 5, 2, 64, 2, 1e-3, 512, 3, path=path)
        outs.append((fit.best_params.cpu().numpy(), fit.mse_log.cpu().numpy()))
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.linalg.norm(outs[0][0] - outs[2][0]) <= 1e-4 * np.linalg.norm(outs[2][0])
    np.testing.assert_allclose(outs[0][1][:, 0], outs[2][1][:, 0], rtol=1e-5)
    assert np.array_equal(outs[0][1][:, 1], outs[2][1][:, 1])


def test_fit_matches_torch_port_trajectory(dev):
    """Same seed, same draws: the first epochs of the HIP fit follow the torch-CPU restatement of the
    reference loop (loss per step within 1e-4 while the trajectories have not yet drifted)."""
    import torch_port as TP
    img = synthetic_tile(2, 8, 32, 48)
    torch.manual_seed(19920517)
    r = TP.fit(img, 5, 2, 64, 2, 1e-3, 256, 2, faithful=False)
    torch.manual_seed(19920517)
    fit = codec.fit_device(ops.to_device_u16(img, dev), 5, 2, 64, 2, 1e-3, 256, 2, keep_losses=True)
    l_hip = fit.losses.cpu().numpy().reshape(-1)
    np.testing.assert_allclose(l_hip, np.array(r["losses"]), rtol=1e-4)
    mse = fit.mse_log.cpu().numpy()
    for (e, m_ref, imp), row in zip(r["epoch_mse"], mse):
        assert abs(row[0] - m_ref) <= 1e-4 * m_ref and bool(row[1]) == imp


def test_cli_round_trip(dev, tmp_path):
    """encode.py -> .bin -> decode.py on a small TIFF: container layout, log records the reference's
    results_summary.py regexes expect, high bits preserved, and the decoded raster equals the oracle's
    decode of the very payload the file carries."""
    import re
    img = synthetic_tile(11, 8, 40, 56)
    src = tmp_path / "tile.tif"
    raster_io.write_raster(str(src), img)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "lbdrn-msic_amd"))
    enc = [sys.executable, os.path.join(ROOT, "lbdrn-msic_amd", "encode.py"), "-i", str(src), "-o", str(tmp_path / "out"),
           "-K", "5", "-D", "2", "-bc", "64", "-nl", "2", "-lr", "0.001", "-bs", "512", "-e", "3", "-sr", "1", "-prec", "16"]
    subprocess.run(enc, check=True, env=env, capture_output=True)
    outdir = tmp_path / "out" / "tile_r1_K5_bc64_nl2_D2_prec16_lr0.001_bs512_e3"
    binp = outdir / "tile.bin"
    raw = binp.read_bytes()
    n, sr, w, h, K, bc, nl, D, nn, base = container.unpack_header(raw)
    assert (n, sr, w, h, K, bc, nl, D) == (15, 1, 56, 40, 5, 64, 2, 2)
    assert n + nn[0] + base[0] == len(raw)
    log = (outdir / "encode.txt").read_text()
    assert re.search(r"nn: (\d+) bytes", log) and re.search(r"MSB: (\d+) bytes", log)
    assert re.search(r"Time elapsed: (\d+\.\d+)", log) and "best epoch" in log
    # decode with -org: metrics logged, recon removed (ref decode.py:223-224)
    dec = [sys.executable, os.path.join(ROOT, "lbdrn-msic_amd", "decode.py"), "-i", str(binp), "-org", str(src)]
    subprocess.run(dec, check=True, env=env, capture_output=True)
    dlog = (outdir / "decode.txt").read_text()
    mse = float(re.search(r"MSE: (\d+\.\d+)", dlog).group(1))
    psnr = float(re.search(r"PSNR: (\d+\.\d+)", dlog).group(1))
    assert re.search(r"bpsp=(\d+\.\d+)", dlog) and re.search(r"Total size: (\d+) bytes", dlog)
    assert not (outdir / "tile_recon.tif").exists()
    assert abs(psnr - 10 * np.log10(1e8 / mse)) < 1e-3 and psnr > 55
    # decoded raster == oracle decode of the payload in the file
    params = container.decode_weights(raw[n:n + nn[0]])
    msb = container.decode_base(raw[n + nn[0]:])
    assert np.array_equal(msb, img >> 5)
    rec_o = O.decode(msb.astype(np.uint16), 5, 2, O.FeatCfg(), params, 64, 2)
    rec = codec.apply_image(msb, params, 5, 2, 64, 2, cfg=FeatCfg())
    assert np.array_equal(rec, rec_o)
    assert abs(float(np.mean((img.astype(np.float32) - rec.astype(np.float32)) ** 2)) - mse) < 1e-3 * mse
    # second invocation: both CLIs see their completion markers and skip
    assert "already" in subprocess.run(enc, check=True, env=env, capture_output=True, text=True).stdout
    assert "already" in subprocess.run(dec, check=True, env=env, capture_output=True, text=True).stdout


def test_cli_split_ratio_tiles(dev, tmp_path):
    img = synthetic_tile(12, 4, 33, 47)
    src = tmp_path / "t.tif"
    raster_io.write_raster(str(src), img)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "lbdrn-msic_amd"))
    subprocess.run([sys.executable, os.path.join(ROOT, "lbdrn-msic_amd", "encode.py"), "-i", str(src), "-o",
                    str(tmp_path / "o"), "-K", "4", "-D", "1", "-bs", "128", "-e", "1", "-sr", "2"],
                   check=True, env=env, capture_output=True)
    outdir = tmp_path / "o" / "t_r2_K4_bc64_nl2_D1_prec16_lr0.001_bs128_e1"
    raw = (outdir / "t.bin").read_bytes()
    n, sr, w, h, K, bc, nl, D, nn, base = container.unpack_header(raw)
    assert (n, sr, w, h) == (8 + 7 * 4, 2, 47, 33) and n + sum(nn) + sum(base) == len(raw)
    import decode as dec_mod
    sys.argv = ["decode.py"]
    assert dec_mod.main(["-i", str(outdir / "t.bin")]) == 0
    rec = raster_io.read_raster(str(outdir / "t_recon.tif"))
    assert rec.shape == img.shape and np.array_equal(rec >> 4, img >> 4)


def test_tiles_sharded_over_two_ranks_equal_the_serial_run(dev, tmp_path):
    """torchrun with two ranks (on a one-GPU box both share cuda:0 -- placement on distinct GPUs is the next test's
    subject): the 3x3 tiles of
    an image fitted round-robin by the ranks give byte for byte the .bin of the single-process run -- the RNG
    replay of the tiles a rank skips included -- and the sharded decode reports the serial run's metrics.  Then
    the sweep driver over two K values on two ranks, summarised into the CSV."""
    import re
    img = synthetic_tile(13, 4, 50, 62)
    src = tmp_path / "scene.tif"
    raster_io.write_raster(str(src), img)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "lbdrn-msic_amd"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    flags = ["-K", "4", "-D", "1", "-bs", "96", "-e", "3", "-sr", "3"]
    pkg = os.path.join(ROOT, "lbdrn-msic_amd")
    run2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
            "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200)]
    subprocess.run([sys.executable, os.path.join(pkg, "encode.py"), "-i", str(src), "-o", str(tmp_path / "one")] + flags,
                   check=True, env=env, capture_output=True)
    subprocess.run(run2 + [os.path.join(pkg, "encode.py"), "-i", str(src), "-o", str(tmp_path / "two")] + flags,
                   check=True, env=env, capture_output=True, timeout=600)
    sub = "scene_r3_K4_bc64_nl2_D1_prec16_lr0.001_bs96_e3"
    a, b = (tmp_path / "one" / sub / "scene.bin").read_bytes(), (tmp_path / "two" / sub / "scene.bin").read_bytes()
    assert a == b
    # ... and of the run that fits the tiles strictly one after another (the default keeps two in flight)
    subprocess.run([sys.executable, os.path.join(pkg, "encode.py"), "-i", str(src), "-o", str(tmp_path / "seq")] + flags,
                   check=True, env=dict(env, LBDRN_IN_FLIGHT="1"), capture_output=True)
    assert (tmp_path / "seq" / sub / "scene.bin").read_bytes() == a
    subprocess.run([sys.executable, os.path.join(pkg, "decode.py"), "-i", str(tmp_path / "one" / sub / "scene.bin"),
                    "-org", str(src)], check=True, env=env, capture_output=True)
    subprocess.run(run2 + [os.path.join(pkg, "decode.py"), "-i", str(tmp_path / "two" / sub / "scene.bin"), "-org", str(src)],
                   check=True, env=env, capture_output=True, timeout=600)
    pick = lambda d: re.findall(r"(MSE: \S+|PSNR: \S+|Total size: .*)", (tmp_path / d / sub / "decode.txt").read_text())
    assert pick("one") == pick("two") and len(pick("one")) == 3
    # sweep: (image, K) points dealt over the two ranks, CSV written by rank 0
    out = subprocess.run(run2 + [os.path.join(pkg, "sweep.py"), "--images", str(src), "--k", "3", "4", "-o",
                                 str(tmp_path / "sw"), "-D", "1", "-bs", "96", "-e", "2", "--summary"],
                         check=True, env=env, capture_output=True, text=True, timeout=600).stdout
    assert "All files processed." in out and out.count(": ok in") == 2
    import csv
    rows = list(csv.reader(open(tmp_path / "sw" / "results_r1_bc64_nl2_D1_prec16_lr0.001_bs96_e2.csv")))
    assert rows[0] == ["K", "scene_MSE", "scene_PSNR", "scene_bpsp", "scene_bits"]
    assert [r[0] for r in rows[1:]] == ["K3", "K4"] and all(float(x) > 0 for r in rows[1:] for x in r[1:])
    assert float(rows[1][1]) < float(rows[2][1])        # more bits kept exactly (smaller K) -> smaller error


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: ranks must land on distinct devices")
def test_two_ranks_run_on_their_own_gpus(dev, tmp_path):
    """On a node with >= 2 GPUs: rank r's fits, payload coding and decode run on cuda:r (the log names the device of
    every fit; ops._call takes device and stream from the tensors), and the .bin equals the single-process one."""
    img = synthetic_tile(17, 4, 40, 48)
    src = tmp_path / "scene.tif"
    raster_io.write_raster(str(src), img)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "lbdrn-msic_amd"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    flags = ["-K", "4", "-D", "1", "-bs", "96", "-e", "2", "-sr", "2"]
    pkg = os.path.join(ROOT, "lbdrn-msic_amd")
    run2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
            "--master-addr", "127.0.0.1", "--master-port", str(29900 + os.getpid() % 90)]
    subprocess.run([sys.executable, os.path.join(pkg, "encode.py"), "-i", str(src), "-o", str(tmp_path / "one")] + flags,
                   check=True, env=env, capture_output=True)
    subprocess.run(run2 + [os.path.join(pkg, "encode.py"), "-i", str(src), "-o", str(tmp_path / "two")] + flags,
                   check=True, env=env, capture_output=True, timeout=600)
    sub = "scene_r2_K4_bc64_nl2_D1_prec16_lr0.001_bs96_e2"
    assert (tmp_path / "one" / sub / "scene.bin").read_bytes() == (tmp_path / "two" / sub / "scene.bin").read_bytes()
    log = (tmp_path / "two" / sub / "encode.txt").read_text()
    assert log.count(" on cuda:0") == 2 and log.count(" on cuda:1") == 2      # four tiles, two per rank, own GPU each
    subprocess.run(run2 + [os.path.join(pkg, "decode.py"), "-i", str(tmp_path / "two" / sub / "scene.bin"), "-org", str(src)],
                   check=True, env=env, capture_output=True, timeout=600)
    assert "PSNR" in (tmp_path / "two" / sub / "decode.txt").read_text()


def test_full_size_properties(dev):
    """BASELINE.json configs[1] size: the two independent HIP implementations (generic tiled FMA
    kernels, fused MFMA kernel) agree bit for bit on the 33.5 M decoded sub-pixels, high bits are
    preserved exactly, the reconstruction error is bounded by the dropped bits, and the whole-image
    SSE of the two paths agrees to 1e-12."""
    C, H, W, K, D = 8, 2048, 2048, 5, 2
    img = synthetic_tile(0, C, H, W)
    img_d = ops.to_device_u16(img, dev)
    msb_d, mx = ops.split_bits(img_d, K)
    assert mx == int((img >> K).max())
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, FeatCfg(), dev)
    net = ops.make_net(200, 64, C, 2)
    p = torch.from_numpy(_params(np.random.default_rng(1), 200, 64, C, 2) * 2.0).to(dev)
    a = ops.decode_fused(geom, net, msb_d, p, path=MFMA)
    b = ops.decode_fused(geom, net, msb_d, p, path=GEN)
    assert torch.equal(a, b)
    rec = ops.from_device_u16(a)
    assert np.array_equal(rec >> K, img >> K)
    assert np.abs(rec.astype(np.int32) - img.astype(np.int32)).max() <= 31
    s1 = float(ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA).item())
    s2 = float(ops.eval_sse(geom, net, img_d, msb_d, p, path=GEN).item())
    assert abs(s1 - s2) <= 1e-12 * s2
    # a background pass (half as many workgroups, each walking two virtual ones) gives the same double, bit for bit
    s3 = ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA, background=True)
    assert float(s3.item()) == s1
    # SSE is consistent with the decoded raster: sum((y - lab)^2) vs residuals, within rounding of y*31
    lsb_err = ((rec & 31).astype(np.float64) - (img & 31)) / 31.0
    assert abs(np.sum(lsb_err ** 2) - s1) / s1 < 0.05


def test_full_size_properties_wide_network(dev):
    """BASELINE.json configs[2] size (bc = 256): the streaming apply kernel and the per-layer generic kernels agree bit
    for bit on the 33.5 M decoded sub-pixels and to 1e-12 on the whole-image SSE; a background pass gives the same sum."""
    C, H, W, K, D = 8, 2048, 2048, 5, 2
    img = synthetic_tile(3, C, H, W)
    img_d = ops.to_device_u16(img, dev)
    msb_d, mx = ops.split_bits(img_d, K)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, FeatCfg(), dev)
    net = ops.make_net(200, 256, C, 2)
    p = torch.from_numpy(_params(np.random.default_rng(2), 200, 256, C, 2) * 2.0).to(dev)
    a = ops.decode_fused(geom, net, msb_d, p, path=MFMA)
    b = ops.decode_fused(geom, net, msb_d, p, path=GEN)
    assert torch.equal(a, b)
    rec = ops.from_device_u16(a)
    assert np.array_equal(rec >> K, img >> K)
    s1 = ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA).item()
    s2 = ops.eval_sse(geom, net, img_d, msb_d, p, path=GEN).item()
    s3 = ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA, background=True).item()
    assert abs(s1 - s2) <= 1e-12 * s2 and s3 == s1


def test_full_size_fit_is_reproducible_and_self_consistent(dev):
    """BASELINE.json configs[1] end to end (2 epochs to keep it short): two runs give identical bits
    (fixed reduction orders, no atomics on the path); the evaluation MSE the fit selected on equals the MSE
    recomputed from the decode kernel's sigmoid outputs and the label matrix (independent kernels); the
    reconstruction keeps the high bits and beats the no-model baseline (predicting mid-range low bits)."""
    C, H, W, K, D = 8, 2048, 2048, 5, 2
    img = synthetic_tile(3, C, H, W)
    img_d = ops.to_device_u16(img, dev)
    fits = []
    for _ in range(2):
        torch.manual_seed(19920517)
        fits.append(codec.fit_device(img_d, K, D, 64, 2, 1e-3, 8192, 2, keep_losses=True))
    a, b = fits
    assert torch.equal(a.best_params.view(torch.int32), b.best_params.view(torch.int32))
    assert torch.equal(a.losses, b.losses) and torch.equal(a.mse_log, b.mse_log)
    losses = a.losses.cpu().numpy()
    assert np.isfinite(losses).all() and losses[1].mean() <= losses[0].mean()
    mse_log = a.mse_log.cpu().numpy()
    best = int(np.argmin(mse_log[:, 0]))
    out, y = ops.decode_fused(a.geom, a.net, a.msb, a.best_params, want_y=True)
    lab = ops.labels(img_d, K)
    mse_indep = float(((y.double() - lab.double()) ** 2).mean().item())
    assert abs(mse_indep - float(mse_log[best, 0])) <= 2e-6 * mse_indep
    rec = ops.from_device_u16(out)
    assert np.array_equal(rec >> K, img >> K)
    err = float(np.mean((rec.astype(np.float32) - img.astype(np.float32)) ** 2))
    base = float(np.mean(((((img >> K) << K) + 16).astype(np.float32) - img.astype(np.float32)) ** 2))
    assert err < base      # strictly better than no model at all (this line used to allow 2 % worse)


def test_pairs_of_full_size_fits_equal_the_same_fits_alone(dev):
    """The launch sequence bench.py times, at its size: two 8 x 2048^2 tiles through fit_many(in_flight=4) -- which steps
    them as a PAIR per launch (lbdrn_train_epoch_group: blockIdx.y = fit, one reduce launch for both), on a worker
    thread's stream, evaluation passes in the tolerance arithmetic -- against the same two fits alone, one after the
    other, on the caller's stream: best weights, evaluation log and decoded raster bit for bit (the
    grouped launch had only been compared with itself at 8 x 70 x 90)."""
    C, H, W, K, D = 8, 2048, 2048, 5, 2
    tiles = [ops.to_device_u16(synthetic_tile(11 + k, C, H, W), dev) for k in range(2)]
    assert ops.train_group_size(C, H, W, K, D, FeatCfg(), 64, 2) >= 2
    together = codec.fit_many(tiles, K, D, 64, 2, 1e-3, 8192, 2, cfg=FeatCfg(), seed=19920517, in_flight=4)
    torch.cuda.synchronize()
    for k, t in enumerate(tiles):
        alone = codec.fit_many([t], K, D, 64, 2, 1e-3, 8192, 2, cfg=FeatCfg(), seed=19920517, in_flight=1)[0]
        assert torch.equal(alone.best_params.view(torch.int32), together[k].best_params.view(torch.int32)), k
        assert torch.equal(alone.mse_log, together[k].mse_log), k
        ra = codec.apply_device(alone.geom, alone.net, alone.msb, codec.truncate_device(alone.best_params, 16))
        rt = codec.apply_device(together[k].geom, together[k].net, together[k].msb,
                                codec.truncate_device(together[k].best_params, 16))
        assert torch.equal(ra, rt), k
    # the two tiles are different images: their fits must differ (a pair launch that read fit 0's rows twice would pass the above)
    assert not torch.equal(together[0].best_params, together[1].best_params)


def test_wide_train_kernels_at_full_size(dev):
    """BASELINE.json configs[2] at its size: the bc = 256 training step on the 8 x 2048^2 tile (3.5 GB row matrix, N =
    4.19 M rows, the gradient workspace of a full 8192-row minibatch).  Three minibatches -- the last rows, the first
    rows, rows from all over -- give the same losses (2e-5) and Adam moments on the fused path and on the generic path
    (window gather, one GEMM launch per layer), and the fused path is bitwise reproducible."""
    C, H, W, K, D = 8, 2048, 2048, 5, 2
    img = synthetic_tile(4, C, H, W)
    img_d = ops.to_device_u16(img, dev)
    msb_d, mx = ops.split_bits(img_d, K)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, FeatCfg(), dev)
    net = ops.make_net(200, 256, C, 2)
    rng = np.random.default_rng(9)
    p0 = _params(rng, 200, 256, C, 2)
    N, bs = H * W, 8192
    perm_np = np.concatenate([np.arange(N - bs, N), np.arange(bs), rng.integers(0, N, bs)]).astype(np.int64)
    perm = torch.from_numpy(perm_np).to(dev)
    res = {}
    for tag, path in (("fused", MFMA), ("generic", GEN), ("again", MFMA)):
        p = torch.from_numpy(p0.copy()).to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        losses = torch.zeros(3, dtype=torch.float32, device=dev)
        ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 0, 1e-3, losses, path=path)
        res[tag] = (losses.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy(), p.cpu().numpy())
    np.testing.assert_allclose(res["fused"][0], res["generic"][0], rtol=2e-5)
    assert np.abs(res["fused"][1] - res["generic"][1]).max() <= 2e-4 * np.abs(res["generic"][1]).max()
    assert np.abs(res["fused"][2] - res["generic"][2]).max() <= 4e-4 * np.abs(res["generic"][2]).max()
    for k in range(4):
        assert np.array_equal(res["fused"][k].view(np.uint32), res["again"][k].view(np.uint32)), k


def test_large_tile_64bit_indexing(dev):
    """A 4096 x 4096 x 8 tile: the row matrix (3.5 G floats) and the band planes are indexed past 2^31.
    Minibatches drawn from the top, the bottom and all over the raster give the same losses and moments
    on the fused path (materialised rows) and the generic path (window gather), and the fused decode of
    the last rows equals the generic one bit for bit."""
    C, H, W, K, D = 8, 4096, 4096, 5, 2
    base = synthetic_tile(5, C, 1024, 1024)
    img = np.ascontiguousarray(np.tile(base, (1, 4, 4)))
    img[:, -1024:, -1024:] = synthetic_tile(6, C, 1024, 1024)   # the far corner differs from the rest
    img_d = ops.to_device_u16(img, dev)
    msb_d, mx = ops.split_bits(img_d, K)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, FeatCfg(), dev)
    net = ops.make_net(200, 64, C, 2)
    rng = np.random.default_rng(8)
    p0 = _params(rng, 200, 64, C, 2)
    N, bs = H * W, 8192
    perm_np = np.concatenate([np.arange(N - bs, N), np.arange(bs), rng.integers(0, N, bs)]).astype(np.int64)
    perm = torch.from_numpy(perm_np).to(dev)
    res = {}
    for path in (MFMA, GEN):
        p = torch.from_numpy(p0.copy()).to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        losses = torch.zeros(3, dtype=torch.float32, device=dev)
        ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 0, 1e-3, losses, path=path)
        res[path] = (losses.cpu().numpy(), m.cpu().numpy())
    np.testing.assert_allclose(res[MFMA][0], res[GEN][0], rtol=2e-5)
    assert np.abs(res[MFMA][1] - res[GEN][1]).max() <= 2e-4 * np.abs(res[GEN][1]).max()
    p = torch.from_numpy(p0 * 2.0).to(dev)
    a = ops.decode_fused(geom, net, msb_d, p, path=MFMA)
    b = ops.decode_fused(geom, net, msb_d, p, path=GEN)
    assert torch.equal(a, b)
    rec = ops.from_device_u16(a[:, -64:, :])
    assert np.array_equal(rec >> K, img[:, -64:, :] >> K)


@pytest.mark.parametrize("shape", [(8, 48, 64), (3, 17, 200), (4, 130, 70), (1, 9, 9)])
def test_background_evaluation_pass_is_the_same_sum(dev, shape):
    """LBDRN_EVAL_BACKGROUND: any launch shape, the same float64 -- small rasters with fewer tiles than CUs, ragged
    tiles, a single tile; and a whole fit whose passes ran beside its training equals one whose passes stood in the chain."""
    C, H, W = shape
    img = synthetic_tile(7 + C, C, H, W)
    img_d = ops.to_device_u16(img, dev)
    msb_d, mx = ops.split_bits(img_d, 5)
    geom = ops.FeatureGeometry(C, H, W, 5, 2, mx, FeatCfg(), dev)
    F = FeatCfg().feature_dim(C, 2)
    net = ops.make_net(F, 64, C, 2)
    p = torch.from_numpy(_params(np.random.default_rng(C), F, 64, C, 2) * 2.0).to(dev)
    a = ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA)
    b = ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA, background=True)
    c = ops.eval_sse(geom, net, img_d, msb_d, p, path=GEN, background=True)   # the generic path ignores the hint
    assert a.item() == b.item() and a.item() > 0
    assert abs(c.item() - a.item()) <= 1e-11 * a.item()
    fits = []
    for alone in (True, False):
        torch.manual_seed(19920517)
        fits.append(codec.fit_device(img_d, 5, 2, 64, 2, 1e-3, 256, 3, alone=alone))
    assert torch.equal(fits[0].best_params.view(torch.int32), fits[1].best_params.view(torch.int32))
    assert torch.equal(fits[0].mse_log, fits[1].mse_log)


def test_fast_evaluation_pass_ranks_epochs_like_the_canonical_one(golden, dev):
    """LBDRN_EVAL_FAST: the per-epoch evaluation in the training step's arithmetic.  Its float64 sum is within 1e-6
    relative of the canonical pass (untrained weights, the trained weights of a learnable image, the wide-argument
    weights of the "scaled" raster; bc = 64 and 256, with and without the positional embedding), it is reproducible
    bit for bit, the same on a background launch, and a fit that ranks its epochs with it picks the same epoch and
    ends on the same weights as one that ranks them canonically."""
    for tag in ("bc64", "bc256", "embed", "scaled"):
        G = golden["rasters_learn_" + tag]
        img, K, D, bc, nl = G["img"], int(G["K"]), int(G["D"]), int(G["bc"]), int(G["nl"])
        f = G["flags"]
        cfg = FeatCfg(bool(f[0]), bool(f[1]), 1.4, 12, bool(f[2]), bool(f[3]))
        C, H, W = img.shape
        img_d = ops.to_device_u16(img, dev)
        msb_d, mx = ops.split_bits(img_d, K)
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        net = ops.make_net(cfg.feature_dim(C, D), bc, C, nl)
        p = torch.from_numpy(G["params"]).to(dev)
        canon = float(ops.eval_sse(geom, net, img_d, msb_d, p).item())
        fast = [float(ops.eval_sse(geom, net, img_d, msb_d, p, fast=True, background=b).item()) for b in (False, False, True)]
        assert fast[0] == fast[1] == fast[2], tag
        assert abs(fast[0] - canon) <= 1e-6 * canon, (tag, fast[0], canon)
    # whole fits ranked either way: the learnable image at bc = 64 and bc = 256 (k_apply_wide's pass), the positional
    # embedding, and a noise tile -- whose late epochs (lr 1e-5, 1e-6) differ by parts in 1e5 of the MSE: the case where
    # a ranking in another arithmetic could pick another epoch (this comparison ran on one image)
    cases = [(golden["rasters_learn_bc64"]["img"], 64, FeatCfg(), 12, 8192),
             (golden["rasters_learn_bc256"]["img"], 256, FeatCfg(), 6, 8192),
             (golden["rasters_learn_embed"]["img"], 64, FeatCfg(True, True, 1.4, 12, True, True), 6, 8192),
             (synthetic_tile(31, 8, 96, 128), 64, FeatCfg(), 10, 1024)]
    for img, bc, cfg, epochs, bs in cases:
        img_d = ops.to_device_u16(img, dev)
        fits = []
        for canonical in ("0", "1"):
            os.environ["LBDRN_EVAL_CANONICAL"] = canonical
            try:
                torch.manual_seed(19920517)
                fits.append(codec.fit_device(img_d, 5, 2, bc, 2, 1e-3, bs, epochs, cfg=cfg))
            finally:
                os.environ.pop("LBDRN_EVAL_CANONICAL", None)
        a, b = fits
        assert torch.equal(a.best_params.view(torch.int32), b.best_params.view(torch.int32)), (bc, epochs)
        assert torch.equal(a.mse_log[:, 1], b.mse_log[:, 1]), (bc, epochs)
        np.testing.assert_allclose(a.mse_log[:, 0].cpu().numpy(), b.mse_log[:, 0].cpu().numpy(), rtol=1e-6)


def test_fit_learns_what_the_torch_port_learns(golden, dev):
    """A fit that has something to learn: the smooth 8 x 128 x 128 image of the learnable reference rasters
    (tests/golden/make_golden_round3.py; its low bits follow from the neighbours' high bits), 90 epochs = 180 Adam steps,
    same seed and draws on both sides.  The HIP fit and the torch-CPU restatement of the reference loop
    (oracle/torch_port.py) must reach the same reconstruction quality -- PSNR within 0.05 dB of each other -- and that
    quality must be well above "predict mid-range" (the assertion a noise tile cannot make)."""
    import torch_port as TP
    img = golden["rasters_learn_bc64"]["img"]
    K, D, bc, nl, lr, bs, epochs = 5, 2, 64, 2, 1e-3, 8192, 90
    torch.manual_seed(19920517)
    r = TP.fit(img, K, D, bc, nl, lr, bs, epochs, faithful=False)
    rec_t = TP.apply(r["msb"], container.truncate_precision(r["params"], 16), K, D, bc, nl)
    torch.manual_seed(19920517)
    fit = codec.fit_device(ops.to_device_u16(img, dev), K, D, bc, nl, lr, bs, epochs)
    rec_h = ops.from_device_u16(codec.apply_device(fit.geom, fit.net, fit.msb, codec.truncate_device(fit.best_params, 16)))
    psnr = lambda rec: 10 * np.log10(10000.0 ** 2 / np.mean((rec.astype(np.float64) - img.astype(np.float64)) ** 2))
    mid = psnr(((img >> K) << K) + 16)
    p_t, p_h = psnr(rec_t), psnr(rec_h)
    assert p_t > mid + 4.0 and p_h > mid + 4.0, (mid, p_t, p_h)
    assert abs(p_t - p_h) <= 0.05, (p_t, p_h)
    best_t = min(m for _, m, _ in r["epoch_mse"])
    assert abs(float(fit.mse_log[:, 0].min().item()) - best_t) <= 5e-3 * best_t


def test_fits_in_flight_together_equal_fits_one_by_one(dev):
    """codec.fit_many: several images progressing at once on their own streams and host threads give, bit for
    bit, what fitting them one after another gives (each fit seeds the shared generator itself, inside one
    critical section with all its draws)."""
    shapes = [(4, 70, 90), (8, 64, 64), (3, 100, 41), (8, 96, 80), (4, 70, 90)]
    imgs = [ops.to_device_u16(synthetic_tile(20 + i, *s), dev) for i, s in enumerate(shapes)]
    args = (5, 2, 64, 2, 1e-3, 256, 4)
    one_by_one = []
    for img_d in imgs:
        torch.manual_seed(19920517)
        one_by_one.append(codec.fit_device(img_d, *args))
    for in_flight in (2, 3):
        together = codec.fit_many(imgs, *args, seed=19920517, in_flight=in_flight)
        torch.cuda.synchronize()
        for a, b in zip(one_by_one, together):
            assert torch.equal(a.best_params.view(torch.int32), b.best_params.view(torch.int32))
            assert torch.equal(a.mse_log, b.mse_log)
    # `then` runs on the worker's stream; results come back in input order
    recs = codec.fit_many(imgs, *args, seed=19920517, in_flight=2,
                          then=lambda fit: codec.apply_device(fit.geom, fit.net, fit.msb,
                                                              codec.truncate_device(fit.best_params, 16)))
    torch.cuda.synchronize()
    for img_d, rec in zip(imgs, recs):
        assert rec.shape == img_d.shape
        assert torch.equal((rec.to(torch.int32) & 0xFFFF) >> 5, (img_d.to(torch.int32) & 0xFFFF) >> 5)
    with pytest.raises(ValueError):
        codec.fit_many(imgs, *args, seed=None, in_flight=2)


def test_fits_of_a_group_step_side_by_side_and_equal_fits_one_by_one(dev):
    """codec.fit_group / lbdrn_train_epoch_group: independent fits of one raster shape whose minibatches run in ONE
    launch (blockIdx.y = fit) give, bit for bit, what each gives alone -- also with a ragged last minibatch, with three
    fits in the group, through fit_many's automatic grouping (in_flight = 4), and with the per-step losses kept."""
    shape = (8, 70, 90)                              # 6300 pixels: 24 full minibatches of 256 + one of 156
    imgs = [ops.to_device_u16(synthetic_tile(40 + i, *shape), dev) for i in range(5)]
    other = ops.to_device_u16(synthetic_tile(50, 4, 64, 64), dev)
    args = (5, 2, 64, 2, 1e-3, 256, 3)
    alone = []
    for img_d in imgs + [other]:
        torch.manual_seed(19920517)
        alone.append(codec.fit_device(img_d, *args, keep_losses=True))
    same = lambda a, b: (torch.equal(a.best_params.view(torch.int32), b.best_params.view(torch.int32)) and
                         torch.equal(a.mse_log, b.mse_log))
    for n in (2, 3):
        grouped = codec.fit_group(imgs[:n], *args, seed=19920517, keep_losses=True)
        torch.cuda.synchronize()
        for a, b in zip(alone, grouped):
            assert same(a, b)
            assert torch.equal(a.losses, b.losses)
    # fit_many: pairs of equal shape step together, the odd one out (another shape) and the leftover run alone
    together = codec.fit_many(imgs[:3] + [other] + imgs[3:], *args, seed=19920517, in_flight=4)
    torch.cuda.synchronize()
    for a, b in zip(alone[:3] + [alone[5]] + alone[3:5], together):
        assert same(a, b)
    with pytest.raises(ValueError):
        codec.fit_group([imgs[0], other], *args, seed=19920517)


def test_bench_prints_one_json_line_with_the_contract_keys(dev):
    """bench.py on a small tile (same code path as the full-size run): exactly one JSON line on stdout with the
    driver's keys, the roofline and cpu_baseline objects, and values of the right kind."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--height", "192", "--width", "256", "-bs", "2048", "-e", "3", "--cpu-sample", "48"],
                         check=True, env=env, capture_output=True, text=True, timeout=900).stdout
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "single_tile_ms", "records"):
        assert key in d, key
    assert len(d["records"]) == d["n_gpus"] * d["steps"] and d["single_tile_ms"] > 0
    assert 0 < d["single_tile_decode_ms"] < d["single_tile_encode_ms"] < d["single_tile_ms"]
    assert d["unit"] == "Mpixels/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 192 * 256 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * d["value"] + 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert r["kernel_us"] > 0 and r["reduce_adam_us"] > 0 and "unaccounted_us" in r
    # round 4: the kernel's own duration and the marginal (doubling-probe) figure side by side; the timed launch sequence
    # checked against the same tile fitted alone; the legs of the other configurations only at the headline size
    assert r["marginal_us"] > 0 and abs(r["frac_live"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] <= r["frac_live"] + 1e-9
    assert d["timed_equals_lone"] is True and "other_configs" not in d
    assert d["repeats"] == 3 and len(d["ms_per_step_all_repeats"]) == 3 and d["ranks_seen"] == 1 and len(d["rank_elapsed_ms"]) == 1
    assert min(d["ms_per_step_all_repeats"]) <= d["ms_per_step"] <= max(d["ms_per_step_all_repeats"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert str(c["cores"]) in c["form_B_by_threads"] and c["vectorised_form_B"] == max(c["form_B_by_threads"].values())
    # round 6: the tile as an equation -- its terms come from the line's own figures and add up to the measured tile
    a = d["accounting"]
    launches = 3 * ((192 * 256 + 2047) // 2048) / d["config"]["tiles_per_launch"]
    assert a["training_launches_per_tile"] == launches and abs(a["training_ms"] - launches * r["kernel_us"] * 1e-3) < 0.02
    assert abs(a["evaluation_passes_ms"] - 3 * r["apply_pass_ms"]) < 0.02 and abs(a["measured_ms_per_tile"] - d["ms_per_step"]) < 0.02
    assert abs(a["training_ms"] + a["evaluation_passes_ms"] + a["decode_ms"] + a["rest_ms"] - a["measured_ms_per_tile"]) < 0.05
    # ... and the opt-in evaluation arithmetic is timed beside the one the fits ran, never instead of it
    assert r["apply_pass_x16_ms_opt_in"] > 0 and "LBDRN_EVAL_X16" not in r["apply_pass"]


def test_bench_four_band_run_has_its_own_roofline(dev):
    """`bench.py --bands 4` (the reference's majority shape) on a small tile: the line names the workload, the fits of the timed
    region equal the lone fit bit for bit, FLOPs are counted over the 96 features the step multiplies."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bands", "4", "--steps", "4", "--warmup", "2", "--height", "192",
                          "--width", "256", "-bs", "2048", "-e", "3", "--no-cpu-baseline", "--repeats", "1"],
                         check=True, env=env, capture_output=True, text=True, timeout=900).stdout
    d = json.loads([ln for ln in out.splitlines() if ln.strip()][-1])
    assert "4-band" in d["config"]["workload"] and d["timed_equals_lone"] is True and d["config"]["tiles_per_launch"] == 2
    r = d["roofline"]
    assert r["features_multiplied"] == 96 and r["fits_per_launch"] == 2 and r["kernel_us"] > 0 and 0 < r["frac"] <= r["frac_live"] + 1e-9
    assert r["flop_per_launch"] == 2 * 2048 * (3 * 2 * (96 * 64 + 64 * 64 + 64 * 4) - 2 * 96 * 64)


def test_the_reference_scene_size_through_the_clis(dev, tmp_path):
    """the reference's real image shape, once -- an 8-band 6000 x 6000 uint16 scene (the GF6-WFI
    scenes of its tables; ref encode.py:228-262, LBDRNdataset.py:46-68) through encode.py /
    decode.py with -sr 1 (one 36 M-pixel fit: 30 GB of rows, lbdrn_randperm beyond its partitioned path's 2048^2) and
    -sr 3 (nine 2000 x 2000 tiles, fits in flight sized against the free memory).  The high bits come back exact, the
    payload sizes agree with each other, the device high-water mark stays under codec.fit_bytes' estimate, and the
    permutation of 36 M elements is a permutation.  Three epochs keep the test short; scripts/scene_timing.py is the
    ten-epoch record (profiles/r05_scene_6000x6000x8.jsonl: 1.7 s to encode, 0.5 s to decode, 31.8 GiB)."""
    import encode as enc_mod
    import decode as dec_mod
    free, total = torch.cuda.mem_get_info(dev)
    side, C, K, epochs = 6000, 8, 5, 3
    need = codec.fit_bytes(C, side, side, K, 2, 64, 2, 8192, epochs)
    if free < 1.2 * need:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of free device memory")
    img = synthetic_tile(7, C, side, side)
    src = str(tmp_path / "scene.npy")
    raster_io.write_raster(src, img)
    n = side * side
    perm = ops.randperm([19920517], n, dev)[0]
    assert int(perm.min()) == 0 and int(perm.max()) == n - 1 and torch.equal(torch.sort(perm).values, torch.arange(n, device=dev))
    del perm
    sizes = {}
    for sr in (1, 3):
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        out = tmp_path / f"out{sr}"
        assert enc_mod.main(["-i", src, "-o", str(out), "-sr", str(sr), "-e", str(epochs)]) in (0, None)
        sub = out / f"scene_r{sr}_K{K}_bc64_nl2_D2_prec16_lr0.001_bs8192_e{epochs}"
        assert dec_mod.main(["-i", str(sub / "scene.bin")]) in (0, None)
        peak = torch.cuda.max_memory_allocated(dev)
        tile = side // sr + side % sr
        in_flight = 1 if sr == 1 else 4
        assert peak <= in_flight * codec.fit_bytes(C, tile, tile, K, 2, 64, 2, 8192, epochs) + 3 * img.nbytes, (sr, peak)
        rec = raster_io.read_raster(str(sub / "scene_recon.tif"))
        assert rec.shape == img.shape and np.array_equal(rec >> K, img >> K), sr
        mse = float(np.mean((img.astype(np.float32) - rec.astype(np.float32)) ** 2))
        assert mse < 31.0 ** 2 / 3, (sr, mse)                      # (better than predicting mid-range; a fit that learned: ~81)
        sizes[sr] = os.path.getsize(sub / "scene.bin")
        os.remove(sub / "scene_recon.tif")
        del rec
    assert abs(sizes[1] - sizes[3]) < 0.02 * sizes[1], sizes       # (the MSB planes dominate; nine small networks instead of one)


@pytest.mark.parametrize("bands", (8, 4))
def test_lone_fit_schedule_guess_is_close_to_its_measurement(dev, bands):
    """A lone fit steps the head of an epoch on the half-chip launch beside the previous epoch's background evaluation pass
    and the rest on every CU (codec.fit_device).  Where the line is drawn is a matter of time only (no bit depends on it:
    test_alone_hint_changes_no_number) -- first by a model with two measured constants (codec.background_steps), from the
    second fit of a shape on by the fit's own events (codec._calibrated_head).  This holds the constants to the measurement,
    so that they cannot go stale silently when a kernel changes: full-size tile, both band counts."""
    if os.environ.get("LBDRN_LONE_HEAD_FRAC") is not None or codec.device_shared():
        pytest.skip("the schedule is overridden by the environment")
    img = ops.to_device_u16(synthetic_tile(0, bands, 2048, 2048), dev)
    K, D, bc, nl, bs, epochs = 5, 2, 64, 2, 8192, 3
    fits = []
    for _ in range(3):
        torch.manual_seed(19920517)
        fits.append(codec.fit_device(img, K, D, bc, nl, 1e-3, bs, epochs))
        torch.cuda.synchronize()
    net = fits[0].net
    steps = (2048 * 2048 + bs - 1) // bs
    guess = codec.background_steps(steps, net, 2048 * 2048, bs)
    got = codec.head_calibration().get(codec._head_key(dev, net, 2048 * 2048, bs))
    assert got is not None, "no fit of this shape left a measurement"
    assert 0.1 * steps < guess < 0.5 * steps
    assert 0.6 * guess <= got <= 1.6 * guess, (guess, got)
    # ... and wherever the line was drawn, the fits are one fit
    for f in fits[1:]:
        assert torch.equal(f.best_params.view(torch.int32), fits[0].best_params.view(torch.int32))


def test_epoch_calls_queued_back_to_back_equal_synchronised_ones(dev):
    """What the shipped library shares with round 5's dropped overlap experiment -- whose variants all aborted when two epoch
    calls were queued back to back -- is exactly that: calls queued behind one another on one stream, each
    opening with its packing launches while the previous call's last reduce launch may still run (codec.fit_device queues two
    per epoch: head and rest).  Twenty-four calls in a row, the lone-fit hint alternating (k_train_stream / k_train_split), no
    host synchronisation in between, against the same calls with a synchronisation after each: identical bits; 8- and 4-band."""
    rng = np.random.default_rng(23)
    for C, H, W, bs in ((8, 70, 90, 512), (4, 61, 47, 300)):
        img = synthetic_tile(9, C, H, W)
        msb, _, mx = O.split_bits(img, 5)
        cfg = FeatCfg()
        F = cfg.feature_dim(C, 2)
        geom = ops.FeatureGeometry(C, H, W, 5, 2, mx, cfg, dev)
        net = ops.make_net(F, 64, C, 2)
        p0 = _params(rng, F, 64, C, 2)
        perms = [torch.from_numpy(rng.permutation(H * W).astype(np.int64)).to(dev) for _ in range(3)]
        img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
        steps = (H * W + bs - 1) // bs
        got = []
        for sync in (True, False):
            p = torch.from_numpy(p0.copy()).to(dev)
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            losses = torch.zeros((24, steps), dtype=torch.float32, device=dev)
            ws = ops.TrainWorkspace(geom, net, bs, dev).prepare(img_d, msb_d, MFMA)
            torch.cuda.synchronize()
            for k in range(24):
                ops.train_epoch(geom, net, img_d, msb_d, perms[k % 3], bs, p, m, v, k * steps, 1e-3, losses[k], path=MFMA, ws=ws,
                                alone=bool(k & 1))
                if sync:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            got.append([t.cpu().numpy().view(np.int32) for t in (p, m, v, losses)])
        for a, b in zip(*got):
            assert np.array_equal(a, b), C
        assert np.isfinite(got[0][3].view(np.float32)).all()


def test_full_size_properties_four_bands(dev):
    """The reference's majority shape at full size (4 x 2048^2, K5 D2 bc64 nl2, F = 100; run.sh:14-28): the generic and the
    fused apply kernels agree bit for bit on the 16.8 M decoded sub-pixels and to 1e-12 on the whole-image SSE, a background
    pass gives the same double, the high bits survive, the error is bounded by the dropped bits; then the fit itself, two
    epochs: two runs identical bit for bit, the MSE it selected on equals the MSE recomputed from the decode kernel's outputs
    (independent kernels), the reconstruction beats "predict mid-range"; and a PAIR of such fits in flight (the launch sequence
    of the bench line's bands4 leg: k_train_stream<24,2,3,6>) equals the same fits alone (mostly k_train_split<24,6>)."""
    C, H, W, K, D = 4, 2048, 2048, 5, 2
    tiles_np = [synthetic_tile(21 + k, C, H, W) for k in range(2)]
    img = tiles_np[0]
    img_d = ops.to_device_u16(img, dev)
    msb_d, mx = ops.split_bits(img_d, K)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, FeatCfg(), dev)
    net = ops.make_net(100, 64, C, 2)
    p = torch.from_numpy(_params(np.random.default_rng(4), 100, 64, C, 2) * 2.0).to(dev)
    a = ops.decode_fused(geom, net, msb_d, p, path=MFMA)
    assert torch.equal(a, ops.decode_fused(geom, net, msb_d, p, path=GEN))
    rec = ops.from_device_u16(a)
    assert np.array_equal(rec >> K, img >> K) and np.abs(rec.astype(np.int32) - img.astype(np.int32)).max() <= 31
    s1 = float(ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA).item())
    s2 = float(ops.eval_sse(geom, net, img_d, msb_d, p, path=GEN).item())
    assert abs(s1 - s2) <= 1e-12 * s2
    assert float(ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA, background=True).item()) == s1
    fast = float(ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA, fast=True).item())
    assert abs(fast - s1) <= 1e-6 * s1
    # the fit
    fits = []
    for _ in range(2):
        torch.manual_seed(19920517)
        fits.append(codec.fit_device(img_d, K, D, 64, 2, 1e-3, 8192, 2, keep_losses=True))
    f0, f1 = fits
    assert torch.equal(f0.best_params.view(torch.int32), f1.best_params.view(torch.int32))
    assert torch.equal(f0.losses, f1.losses) and torch.equal(f0.mse_log, f1.mse_log)
    assert ops.train_step_features(f0.geom, f0.net) == 96
    losses = f0.losses.cpu().numpy()
    assert np.isfinite(losses).all() and losses[1].mean() <= losses[0].mean()
    mse_log = f0.mse_log.cpu().numpy()
    best = int(np.argmin(mse_log[:, 0]))
    out, y = ops.decode_fused(f0.geom, f0.net, f0.msb, f0.best_params, want_y=True)
    lab = ops.labels(img_d, K)
    mse_indep = float(((y.double() - lab.double()) ** 2).mean().item())
    assert abs(mse_indep - float(mse_log[best, 0])) <= 2e-6 * mse_indep
    rec = ops.from_device_u16(out)
    assert np.array_equal(rec >> K, img >> K)
    err = float(np.mean((rec.astype(np.float32) - img.astype(np.float32)) ** 2))
    base = float(np.mean(((((img >> K) << K) + 16).astype(np.float32) - img.astype(np.float32)) ** 2))
    assert err < base
    # a pair in flight against the same fits alone
    tiles = [img_d, ops.to_device_u16(tiles_np[1], dev)]
    assert ops.train_group_size(C, H, W, K, D, FeatCfg(), 64, 2) >= 2
    together = codec.fit_many(tiles, K, D, 64, 2, 1e-3, 8192, 2, cfg=FeatCfg(), seed=19920517, in_flight=4)
    torch.cuda.synchronize()
    for k, t in enumerate(tiles):
        alone = codec.fit_many([t], K, D, 64, 2, 1e-3, 8192, 2, cfg=FeatCfg(), seed=19920517, in_flight=1)[0]
        assert torch.equal(alone.best_params.view(torch.int32), together[k].best_params.view(torch.int32)), k
        assert torch.equal(alone.mse_log, together[k].mse_log), k
    assert torch.equal(together[0].best_params.view(torch.int32), f0.best_params.view(torch.int32))
    assert not torch.equal(together[0].best_params, together[1].best_params)


def test_minibatches_whose_slabs_pass_2_gib_take_the_generic_step(dev):
    """The reference takes any -bs.  The fused bc = 64 step reads one step's gradient slabs through a buffer resource with 32-bit
    offsets (k_reduce_adam): a minibatch size whose slabs would pass 2 GiB (here 2.1 M rows: 65,625 slabs of 73 KB) is refused by
    LBDRN_PATH_MFMA with LBDRN_E_UNSUPPORTED instead of wrapping silently, and LBDRN_PATH_AUTO runs it on the
    generic kernels -- the same numbers as the generic path asked for by name."""
    C, H, W, K, D, bs = 8, 40, 52, 5, 2, 2_100_000
    img = synthetic_tile(5, C, H, W)
    msb, _, mx = O.split_bits(img, K)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, FeatCfg(), dev)
    net = ops.make_net(200, 64, C, 2)
    img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
    with pytest.raises(ops._lib.LbdrnError) as e:
        ops.TrainWorkspace(geom, net, bs, dev).prepare(img_d, msb_d, MFMA)
    assert e.value.code == ops._lib.E_UNSUPPORTED
    ops.TrainWorkspace(geom, net, 900_000, dev).prepare(img_d, msb_d, MFMA)        # (28,125 slabs = 2.10 GB: addressable)
    rng = np.random.default_rng(3)
    p0 = _params(rng, 200, 64, C, 2)
    perm = torch.from_numpy(rng.permutation(H * W).astype(np.int64)).to(dev)
    got = []
    for path in (ops._lib.PATH_AUTO, GEN):
        p = torch.from_numpy(p0.copy()).to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        losses = torch.zeros(1, dtype=torch.float32, device=dev)
        ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 0, 1e-3, losses, path=path)
        got.append([t.cpu().numpy().view(np.int32) for t in (p, m, v, losses)])
    for a, b in zip(*got):
        assert np.array_equal(a, b)
    assert np.isfinite(got[0][3].view(np.float32)).all()


def test_exact_operand_f16_layer0_of_the_evaluation_pass(golden, dev):
    """LBDRN_EVAL_X16 (opt-in): with the flag the fast evaluation pass multiplies layer 0's colour features on the
    f16 matrix pipe -- integer window differences x W_0 in three fp16 pieces, every product exact, float32 sums.  Its float64
    sum must sit as close to the canonical pass as the fast pass does (1e-6 relative; measured ~1e-9), be reproducible and
    the same on a background launch; where the shape or the image does not qualify (MSB values above 2047, positional features,
    absolute colours, bc = 256) the flag is ignored and the sum is the fast pass's, bit for bit; and a fit that ranks its epochs
    with it picks the same epoch and ends on the same weights."""
    rng = np.random.default_rng(8)
    def sums(img, K, D, bc, nl, cfg, params):
        C, H, W = img.shape
        img_d = ops.to_device_u16(img, dev)
        msb_d, mx = ops.split_bits(img_d, K)
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        net = ops.make_net(cfg.feature_dim(C, D), bc, C, nl, cfg.act)
        p = torch.from_numpy(params(cfg.feature_dim(C, D))).to(dev)
        canon = float(ops.eval_sse(geom, net, img_d, msb_d, p).item())
        fast = float(ops.eval_sse(geom, net, img_d, msb_d, p, fast=True).item())
        x = [float(ops.eval_sse(geom, net, img_d, msb_d, p, fast=True, x16=True, background=b).item()) for b in (False, False, True)]
        assert x[0] == x[1] == x[2]
        return canon, fast, x[0], mx
    # qualifying shapes: the trained weights of the learnable rasters (8 and 4 bands), random and huge weights, D = 1 / 3, ReLU
    for tag in ("bc64", "bands4"):
        G = golden["rasters_learn_" + tag]
        canon, fast, x16, mx = sums(G["img"], 5, 2, 64, 2, FeatCfg(), lambda F: G["params"])
        assert mx <= 2047 and x16 != fast and abs(x16 - canon) <= 1e-6 * canon, (tag, x16, fast, canon)
    for (C, H, W, K, D, bc, nl, act, gain) in [(8, 70, 150, 5, 2, 64, 2, "sine", 1.0), (4, 33, 65, 5, 1, 64, 1, "sine", 40.0),
                                                (2, 40, 70, 6, 3, 32, 2, "sine", 3.0), (6, 31, 64, 5, 2, 64, 2, "relu", 30.0),
                                                (8, 48, 64, 9, 2, 64, 2, "sine", 1e-4)]:
        img = rng.integers(0, 65536, (C, H, W)).astype(np.uint16)
        cfg = FeatCfg(activation=act)
        canon, fast, x16, mx = sums(img, K, D, bc, nl, cfg, lambda F: _params(rng, F, bc, C, nl) * np.float32(gain))
        assert mx <= 2047 and (x16 != fast or gain < 1.0), (C, D, bc)      # (weights of 1e-4: differences below a float32 ulp of y vanish)
        assert abs(x16 - canon) <= 1e-6 * canon, (C, D, bc, act, x16, canon)
    # not qualifying: the flag changes nothing
    for (img, K, bc, cfg) in [(rng.integers(0, 65536, (8, 40, 70)).astype(np.uint16), 3, 64, FeatCfg()),                       # msb up to 8191
                              (synthetic_tile(2, 8, 40, 70), 5, 64, FeatCfg(True, True, 1.4, 12, True, True)),                # positional features
                              (synthetic_tile(2, 8, 40, 70), 5, 64, FeatCfg(False, False, 1.4, 12, True, False)),             # absolute colours
                              (synthetic_tile(2, 8, 40, 70), 5, 256, FeatCfg())]:                                              # the streaming kernel
        canon, fast, x16, mx = sums(img, K, 2, bc, 2, cfg, lambda F: _params(rng, F, bc, 8, 2))
        assert x16 == fast, (K, bc, vars(cfg))
    # whole fits ranked with it
    for img, epochs, bs in ((golden["rasters_learn_bc64"]["img"], 12, 8192), (golden["rasters_learn_bands4"]["img"], 6, 8192),
                            (synthetic_tile(31, 8, 96, 128), 10, 1024)):
        img_d = ops.to_device_u16(img, dev)
        fits = []
        for flag in ("0", "1"):
            os.environ["LBDRN_EVAL_X16"] = flag
            try:
                torch.manual_seed(19920517)
                fits.append(codec.fit_device(img_d, 5, 2, 64, 2, 1e-3, bs, epochs))
            finally:
                os.environ.pop("LBDRN_EVAL_X16", None)
        a, b = fits
        assert torch.equal(a.best_params.view(torch.int32), b.best_params.view(torch.int32)), epochs
        assert torch.equal(a.mse_log[:, 1], b.mse_log[:, 1]), epochs
        np.testing.assert_allclose(a.mse_log[:, 0].cpu().numpy(), b.mse_log[:, 0].cpu().numpy(), rtol=1e-6)
        # (measured: the float32 MSEs of the epochs come out IDENTICAL -- the two passes' float64 sums differ by parts in 1e10)
