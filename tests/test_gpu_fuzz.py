"""Randomised shapes: the fused MFMA kernels against the generic kernels (bit-exact for apply, training
tolerance for the step) and, on a subsample, against the oracle.  Seeds are fixed: the cases are the same
every run."""
import os

import numpy as np
import pytest
import torch

import oracle as O
from lbdrn_hip import ops
from lbdrn_hip.features import FeatCfg

pytestmark = pytest.mark.gpu
GEN, MFMA = ops._lib.PATH_GENERIC, ops._lib.PATH_MFMA
SOAK = int(os.environ.get("LBDRN_FUZZ_SOAK", "1"))   # multiply the case count for a longer soak


def _params(rng, F, bc, C, nl, gain):
    parts = []
    for l in range(nl):
        nin = F if l == 0 else bc
        b = (1.0 / nin if l == 0 else np.sqrt(6.0 / nin) / 30.0) * gain
        parts += [rng.uniform(-b, b, bc * nin), rng.uniform(-b, b, bc)]
    b = np.sqrt(6.0 / bc) / 30.0 * gain
    parts += [rng.uniform(-b, b, C * bc), rng.uniform(-b, b, C)]
    return np.concatenate(parts).astype(np.float32)


def _random_case(rng, train):
    C = int(rng.integers(1, 17 if train else 33))
    D = int(rng.integers(0, 4))
    K = int(rng.integers(1, 9))
    H = int(rng.integers(max(2, D + 1), 70))
    W = int(rng.integers(max(2, D + 1), 150))
    coords = bool(rng.integers(0, 2))
    embed = coords and bool(rng.integers(0, 2))
    colors = True if not coords else bool(rng.integers(0, 4) > 0)
    rel = bool(rng.integers(0, 2))
    bc = 64 if train else int(rng.choice([32, 64, 128]))
    nl = int(rng.integers(1, 4))
    act = "relu" if rng.integers(0, 3) == 0 else "sine"      # (round 6: the fused kernels take either hidden activation)
    cfg = FeatCfg(coords, embed, 1.4, 12, colors, rel, act)
    hi = int(rng.choice([255, 4000, 10000, 65535]))
    img = rng.integers(0, hi + 1, (C, H, W)).astype(np.uint16)
    return C, H, W, K, D, bc, nl, cfg, img


def test_apply_fuzz_mfma_equals_generic_and_oracle(dev):
    rng = np.random.default_rng(20240101)
    checked_oracle = 0
    ran = relus = 0
    for it in range(24 * SOAK):
        C, H, W, K, D, bc, nl, cfg, img = _random_case(rng, train=False)
        F = cfg.feature_dim(C, D)
        if F > 400:
            continue
        msb = img >> K
        mx = int(msb.max())
        if mx == 0:
            continue
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        net = ops.make_net(F, bc, C, nl, cfg.act)
        p = torch.from_numpy(_params(rng, F, bc, C, nl, 2.5 * (30.0 if cfg.act else 1.0))).to(dev)   # (no w0 = 30 in front of a ReLU)
        msb_d, img_d = ops.to_device_u16(msb, dev), ops.to_device_u16(img, dev)
        try:
            a, ya = ops.decode_fused(geom, net, msb_d, p, want_y=True, path=MFMA)
        except ops._lib.LbdrnError:
            continue  # shape outside the fused kernel's LDS budget
        b, yb = ops.decode_fused(geom, net, msb_d, p, want_y=True, path=GEN)
        tag = (it, C, H, W, K, D, bc, nl, vars(cfg))
        assert torch.equal(a, b), tag
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), tag
        s1 = float(ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA).item())
        s2 = float(ops.eval_sse(geom, net, img_d, msb_d, p, path=GEN).item())
        assert abs(s1 - s2) <= 1e-11 * max(s2, 1e-30), tag
        ran += 1
        relus += cfg.act
        if H * W <= 2500 and checked_oracle < 8 + 4 * cfg.act:
            ocfg = O.FeatCfg(cfg.use_coordinates, cfg.embedding, 1.4, 12, cfg.use_colors, cfg.relative)
            with O.hidden_activation(cfg.activation):
                ro = O.decode(msb, K, D, ocfg, p.cpu().numpy(), bc, nl, mx)
            assert np.array_equal(ops.from_device_u16(a), ro), tag
            checked_oracle += 1
    assert ran >= 10 and checked_oracle >= 3 and relus >= 2, (ran, checked_oracle, relus)


def test_apply_fuzz_streaming_kernel_equals_generic_and_oracle(dev):
    """bc = 256: k_apply_wide (weights streamed, 16-pixel waves) against the per-layer generic kernels, bit for bit, on
    random geometries -- every constants.py switch, 1..16 bands, D 0..3, ragged tiles, one and two hidden layers."""
    rng = np.random.default_rng(20261004)
    checked_oracle = ran = 0
    for it in range(20 * SOAK):
        C, H, W, K, D, _, _, cfg, img = _random_case(rng, train=True)
        nl = int(rng.integers(1, 3))
        F = cfg.feature_dim(C, D)
        msb = img >> K
        mx = int(msb.max())
        if mx == 0 or F > 400 or not cfg.use_colors:
            continue
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        net = ops.make_net(F, 256, C, nl)
        p = torch.from_numpy(_params(rng, F, 256, C, nl, 2.5)).to(dev)
        msb_d, img_d = ops.to_device_u16(msb, dev), ops.to_device_u16(img, dev)
        a, ya = ops.decode_fused(geom, net, msb_d, p, want_y=True, path=MFMA)
        b, yb = ops.decode_fused(geom, net, msb_d, p, want_y=True, path=GEN)
        tag = (it, C, H, W, K, D, nl, vars(cfg))
        assert torch.equal(a, b), tag
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), tag
        s1 = ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA).item()
        s2 = ops.eval_sse(geom, net, img_d, msb_d, p, path=GEN).item()
        s3 = ops.eval_sse(geom, net, img_d, msb_d, p, path=MFMA, background=True).item()
        assert abs(s1 - s2) <= 1e-11 * max(s2, 1e-30) and s3 == s1, tag
        ran += 1
        if H * W <= 1500 and checked_oracle < 4:
            ocfg = O.FeatCfg(cfg.use_coordinates, cfg.embedding, 1.4, 12, cfg.use_colors, cfg.relative)
            ro = O.decode(msb, K, D, ocfg, p.cpu().numpy(), 256, nl, mx)
            assert np.array_equal(ops.from_device_u16(a), ro), tag
            checked_oracle += 1
    assert ran >= 10 and checked_oracle >= 2, (ran, checked_oracle)


def test_train_fuzz_mfma_matches_generic(dev):
    rng = np.random.default_rng(77)
    done = 0
    for it in range(16 * SOAK):
        C, H, W, K, D, bc, nl, cfg, img = _random_case(rng, train=True)
        F = cfg.feature_dim(C, D)
        msb = img >> K
        mx = int(msb.max())
        if mx == 0 or F > 256:
            continue
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        net = ops.make_net(F, 64, C, nl, cfg.act)
        p0 = _params(rng, F, 64, C, nl, 10.0 if cfg.act else 1.0)
        bs = int(rng.choice([64, 100, 257, 1000]))
        img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
        order = rng.permutation(H * W).astype(np.int64)
        tag = (it, C, H, W, K, D, nl, bs, vars(cfg))

        def run(perm_np):
            perm = torch.from_numpy(perm_np).to(dev)
            nsteps = (len(perm_np) + bs - 1) // bs
            out = {}
            for path in (MFMA, GEN):
                p = torch.from_numpy(p0.copy()).to(dev)
                m, v = torch.zeros_like(p), torch.zeros_like(p)
                losses = torch.zeros(nsteps, dtype=torch.float32, device=dev)
                ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 0, 1e-3, losses, path=path)
                out[path] = [t.cpu().numpy() for t in (p, losses, m, v)]
            return out[MFMA], out[GEN], nsteps

        try:
            # one (usually ragged) step: exp_avg = 0.1 * gradient exactly, the sharp check of the backward pass
            (pa, la, ma, va), (pb, lb, mb, vb), _ = run(order[:int(rng.integers(1, min(bs, H * W) + 1))])
            np.testing.assert_allclose(la, lb, rtol=2e-5, err_msg=str(tag))
            assert np.abs(ma - mb).max() <= 2e-5 * np.abs(mb).max(), tag
            assert np.abs(va - vb).max() <= 5e-5 * np.abs(vb).max(), tag
            # a few steps, the last one ragged.  Adam turns rounding-level differences into lr-sized ones
            # wherever a gradient is near zero (step 1 moves every parameter by lr * g/(|g|+eps)), so
            # parameters are compared (in rms) against the distance they can travel, not to rounding
            (pa, la, ma, va), (pb, lb, mb, vb), nsteps = run(order[:5 * bs + int(rng.integers(1, bs + 1))])
            np.testing.assert_allclose(la, lb, rtol=5e-5, err_msg=str(tag))
            assert np.linalg.norm(pa - pb) <= 0.01 * 1e-3 * nsteps * np.sqrt(len(pa)), tag
            assert np.isfinite(pa).all(), tag
        except ops._lib.LbdrnError:
            continue
        done += 1
    assert done >= 6, done


def test_two_training_kernels_one_set_of_bits_fuzz(dev):
    """k_train_stream (64-row workgroups, half the chip: the step with company) and k_train_split (32-row workgroups, units
    halved between two waves, every CU: the step of a fit alone, LBDRN_TRAIN_ALONE) share one summation tree
    (csrc/train_split.inc): on random images of the shapes that have both -- F = 200, the embedding's F = 250 and (round 6) the
    reference's 4-band F = 100 at bc = 64, two hidden layers; either hidden activation --, random K, batch sizes that leave ragged groups, one-row tails and minibatches shorter than a
    workgroup, two epochs of each leave the same parameters, Adam moments and losses bit for bit, and they train."""
    rng = np.random.default_rng(20260105)
    MFMA = ops._lib.PATH_MFMA
    shapes = [FeatCfg(False, False, 1.4, 12, True, True), FeatCfg(True, True, 1.4, 12, True, True)]
    tails = 0
    for it in range(12 * SOAK):
        cfg = shapes[it % 2]
        C, D, bc, nl = (8, 2, 64, 2) if it % 4 < 2 else (4, 3, 64, 2)   # (4 bands x 7 x 7: the same 192 / 242 features, half the channels)
        if it % 6 == 4:                                # the 4-band shape of the reference's own images: 96 features multiplied
            cfg, C, D = shapes[0], 4, 2
        act = "relu" if it % 3 == 2 else "sine"
        cfg = FeatCfg(cfg.use_coordinates, cfg.embedding, 1.4, 12, cfg.use_colors, cfg.relative, act)
        H, W, K = int(rng.integers(5, 40)), int(rng.integers(5, 60)), int(rng.integers(1, 8))
        bs = int(rng.choice([33, 64, 65, 100, 257, 1000, H * W - 1, H * W + 7]))
        bs = max(bs, 2)
        if it < 2:                                     # (a one-row tail for either shape, whatever the generator drew)
            H, W, bs = (5, 13, 64) if it == 0 else (7, 9, 31)
        img = rng.integers(0, 1 << int(rng.integers(K + 2, 15)), (C, H, W)).astype(np.uint16)
        msb = img >> K
        mx = int(msb.max())
        if mx == 0:
            continue
        F = cfg.feature_dim(C, D)
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        net = ops.make_net(F, bc, C, nl, cfg.act)
        assert ops.train_step_features(geom, net) in (192, 242, 96)      # (the shapes that have both kernels)
        p0 = _params(rng, F, bc, C, nl, 10.0 if cfg.act else 1.0)
        perm = torch.from_numpy(rng.permutation(H * W).astype(np.int64)).to(dev)
        img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
        steps = (H * W + bs - 1) // bs
        tails += (H * W) % bs == 1
        got = []
        for alone in (False, True):
            p = torch.from_numpy(p0.copy()).to(dev)
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            losses = torch.zeros(steps, dtype=torch.float32, device=dev)
            ws = ops.TrainWorkspace(geom, net, bs, dev).prepare(img_d, msb_d, MFMA)
            for e in range(2):
                ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, e * steps, 1e-3, losses, path=MFMA, ws=ws, alone=alone)
            got.append([t.cpu().numpy() for t in (p, m, v, losses)])
        tag = (it, F, H, W, K, bs)
        for a, b in zip(*got):
            assert np.array_equal(a.view(np.int32), b.view(np.int32)), tag
        assert np.isfinite(got[0][0]).all() and np.abs(got[0][1]).max() > 0, tag
    assert tails >= 2     # (a minibatch of ONE row: that step goes out on k_train_stream even for a fit alone)


def test_wide_train_fuzz_split_step_matches_generic(dev):
    """The three launches of the bc >= 128 step (k_train_half -> k_dw_wide -> k_reduce_adam) on random shapes: bands 1..16,
    D 0..3, every constants.py switch, bc 128 / 256, one and two hidden layers, minibatch sizes that leave half-filled
    32-row workgroups, odd workgroup counts (the zero-filled half block of k_dw_wide), slices of 1024 samples cut short
    and more than one slice -- losses and first-step Adam moments against the generic path (window gather, one GEMM launch
    per layer), and bitwise reproducibility of the fused path."""
    rng = np.random.default_rng(4242)
    done = ragged = odd = 0
    for it in range(14 * SOAK):
        C, H, W, K, D, _, _, cfg, img = _random_case(rng, train=True)
        bc, nl = int(rng.choice([128, 256])), int(rng.integers(1, 3))
        F = cfg.feature_dim(C, D)
        msb = img >> K
        mx = int(msb.max())
        if mx == 0 or F > 256:
            continue
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        net = ops.make_net(F, bc, C, nl)
        p0 = _params(rng, F, bc, C, nl, 1.0)
        bs = int(rng.choice([33, 96, 257, 1100, 2100]))
        img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
        order = rng.permutation(H * W).astype(np.int64)
        tag = (it, C, H, W, K, D, bc, nl, bs, vars(cfg))

        def run(perm_np, path):
            perm = torch.from_numpy(perm_np).to(dev)
            nsteps = (len(perm_np) + bs - 1) // bs
            p = torch.from_numpy(p0.copy()).to(dev)
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            losses = torch.zeros(nsteps, dtype=torch.float32, device=dev)
            ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 0, 1e-3, losses, path=path)
            return [t.cpu().numpy() for t in (p, losses, m, v)]

        one = order[:int(rng.integers(1, min(bs, H * W) + 1))]       # one (usually ragged) step: exp_avg = 0.1 * gradient
        try:
            pa, la, ma, va = run(one, MFMA)
        except ops._lib.LbdrnError as e:   # only "this shape has no fused step" skips a case; any other failure is one
            if e.code != ops._lib.E_UNSUPPORTED:
                raise
            continue
        pb, lb, mb, vb = run(one, GEN)
        ragged += len(one) % 32 != 0
        odd += ((len(one) + 31) // 32) % 2 == 1
        np.testing.assert_allclose(la, lb, rtol=2e-5, err_msg=str(tag))
        assert np.abs(ma - mb).max() <= 4e-5 * np.abs(mb).max(), tag
        assert np.abs(va - vb).max() <= 1e-4 * np.abs(vb).max(), tag
        several = order[:min(len(order), 3 * bs + int(rng.integers(1, bs + 1)))]
        pa, la, ma, va = run(several, MFMA)
        pb, lb, mb, vb = run(several, GEN)
        np.testing.assert_allclose(la, lb, rtol=1e-4, err_msg=str(tag))
        assert np.isfinite(pa).all(), tag
        pc, lc, mc, vc = run(several, MFMA)
        for x, y in ((pa, pc), (la, lc), (ma, mc), (va, vc)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), tag
        done += 1
    assert done >= 6 and ragged >= 3 and odd >= 2, (done, ragged, odd)   # half-filled workgroups and odd workgroup counts were met


def test_split_labels_features_fuzz_equal_oracle(dev):
    rng = np.random.default_rng(5)
    for it in range(20 * SOAK):
        C, H, W, K, D, _, _, cfg, img = _random_case(rng, train=False)
        if cfg.feature_dim(C, D) * H * W > 4_000_000:
            continue
        ocfg = O.FeatCfg(cfg.use_coordinates, cfg.embedding, 1.4, 12, cfg.use_colors, cfg.relative)
        msb, lab, mx = O.split_bits(img, K)
        tag = (it, C, H, W, K, D, vars(cfg))
        img_d = ops.to_device_u16(img, dev)
        msb_d, mx_d = ops.split_bits(img_d, K)
        assert mx_d == mx and np.array_equal(ops.from_device_u16(msb_d), msb), tag
        if mx == 0:
            continue
        geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
        idx = None
        if rng.integers(0, 2):
            idx_np = rng.integers(0, H * W, int(rng.integers(1, 300))).astype(np.int64)
            idx = torch.from_numpy(idx_np).to(dev)
        f = ops.features(geom, msb_d, idx).cpu().numpy()
        l = ops.labels(img_d, K, idx).cpu().numpy()
        fo = O.features(msb, D, ocfg, mx)
        lo = lab
        if idx is not None:
            fo, lo = fo[idx_np], lo[idx_np]
        assert np.array_equal(f.view(np.int32), fo.view(np.int32)), tag
        assert np.array_equal(l.view(np.int32), lo.view(np.int32)), tag


@pytest.mark.parametrize("act", ["sine", "relu"])
def test_forward_and_step_fuzz_any_width_equal_oracle(dev, act):
    """lbdrn_forward / lbdrn_train_step take any F, bc, C, nl, B (the generic MFMA GEMM with ragged tiles
    and split-K): forward bit-exact, the step within the training tolerance -- under both hidden activations the
    library knows (lbdrn_net.act: Sine(30), and the reference's named alternative torch.nn.ReLU())."""
    rng = np.random.default_rng(11)
    with O.hidden_activation(act):
        _forward_and_step_fuzz(dev, rng, ops.ACT_RELU if act == "relu" else ops.ACT_SINE, 30.0 if act == "relu" else 1.5)


def _forward_and_step_fuzz(dev, rng, hip_act, gain):
    for it in range(16 * SOAK):
        F = int(rng.integers(1, 300))
        bc = int(rng.choice([1, 7, 32, 64, 65, 100, 128, 200, 256, 300]))
        C = int(rng.integers(1, 40))
        nl = int(rng.integers(1, 5))
        B = int(rng.choice([1, 3, 63, 64, 65, 255, 1000, 4097]))
        tag = (it, F, bc, C, nl, B)
        pn = _params(rng, F, bc, C, nl, gain)   # (ReLU: weights of the size Sine's w0 = 30 would have made them)
        x = rng.uniform(-1, 1, (B, F)).astype(np.float32)
        t = rng.uniform(0, 1, (B, C)).astype(np.float32)
        net = ops.make_net(F, bc, C, nl, hip_act)
        p = torch.from_numpy(pn).to(dev)
        xd, td = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
        y = ops.forward(net, p, xd).cpu().numpy()
        yo = O.forward(pn, F, bc, C, nl, x)
        assert np.array_equal(y.view(np.int32), yo.view(np.int32)), tag
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        loss, g = ops.train_step(net, xd, td, p, m, v, 1, 1e-3)
        po, mo, vo = pn.copy(), np.zeros_like(pn), np.zeros_like(pn)
        lo, go = O.train_step(po, mo, vo, F, bc, C, nl, x, t, 1e-3, 1)
        assert abs(float(loss.item()) - lo) <= 1e-5 * abs(lo), tag
        assert np.abs(g.cpu().numpy() - go).max() <= 2e-5 * np.abs(go).max() + 1e-12, tag
        assert np.abs(m.cpu().numpy() - mo).max() <= 2e-5 * np.abs(mo).max() + 1e-12, tag


def test_activation_special_values_equal_oracle(dev):
    """sin(30 z) and the sigmoid for huge, infinite and NaN pre-activations: y = sigmoid(sin(30 x)) through a
    one-feature network with unit weights, compared bit for bit (NaNs as NaNs)."""
    rng = np.random.default_rng(3)
    F, bc, C, nl = 1, 32, 1, 1
    pn = np.zeros(O.param_count(F, bc, C, nl), np.float32)
    pn[0] = 1.0                      # W0[0,0]: z_0 = x
    pn[bc * F + bc + 0] = 1.0        # W1[0,0]: out = h_0
    special = np.array([0.0, -0.0, 1e-30, -1e-30, 0.5, 1e3, -1e3, 1e6, 7e7, 2.5e9, -2.5e9, 1e20, 3e38, -3e38,
                        np.inf, -np.inf, np.nan], np.float32)
    wide = (rng.choice([-1, 1], 4000) * np.exp(rng.uniform(np.log(1e-6), np.log(3e38), 4000))).astype(np.float32)
    x = np.concatenate([special, wide]).reshape(-1, 1)
    net = ops.make_net(F, bc, C, nl)
    y = ops.forward(net, torch.from_numpy(pn).to(dev), torch.from_numpy(x).to(dev)).cpu().numpy()
    yo = O.forward(pn, F, bc, C, nl, x)
    both_nan = np.isnan(y) & np.isnan(yo)
    assert np.array_equal(y.view(np.int32)[~both_nan], yo.view(np.int32)[~both_nan])
    assert np.array_equal(np.isnan(y), np.isnan(yo))
    # the sigmoid alone: out = W1 . sin(0) + b1 = b1
    for b1 in (0.0, -0.0, 1e-8, 20.0, -20.0, 86.5, -86.5, 200.0, -200.0, 1e30, -1e30, np.inf, -np.inf):
        p2 = np.zeros_like(pn)
        p2[-1] = b1
        y = ops.forward(net, torch.from_numpy(p2).to(dev), torch.zeros(3, 1, device=dev)).cpu().numpy()
        yo = O.forward(p2, F, bc, C, nl, np.zeros((3, 1), np.float32))
        assert np.array_equal(y.view(np.int32), yo.view(np.int32)), b1


def test_cli_fuzz_encode_decode_in_process(dev, tmp_path, monkeypatch):
    """encode.main -> .bin -> decode.main over random small rasters and flag combinations (bands, odd sizes,
    K that make the MSB plane uint8 or uint16, D = 0, tiles): the written raster keeps the high bits, equals
    the oracle's decode of the payloads in the file, and the logged MSE is the raster's."""
    import re
    import decode
    import encode
    from lbdrn_hip import container, raster_io
    from LBDRNdataset import tile_windows
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(99)
    for it in range(6 * SOAK):
        C = int(rng.integers(1, 7))
        H, W = int(rng.integers(9, 80)), int(rng.integers(9, 90))
        K = int(rng.choice([1, 3, 5, 8, 11]))
        D = int(rng.integers(0, 3))
        sr = int(rng.choice([1, 1, 2, 3]))
        if min(H, W) // sr <= D + 1:
            sr = 1
        hi = int(rng.choice([1023, 10000, 65535]))
        img = rng.integers(0, hi + 1, (C, H, W)).astype(np.uint16)
        src = str(tmp_path / f"img{it}.npy")
        np.save(src, img)
        out = str(tmp_path / f"o{it}")
        flags = ["-K", str(K), "-D", str(D), "-bs", "128", "-e", "2", "-sr", str(sr), "-nl", str(int(rng.integers(1, 4)))]
        tag = (it, C, H, W, flags)
        if int((img >> K).max()) == 0:
            # degenerate input (found by the soak: hi = 1023 under K = 11): an all-zero MSB plane makes every feature 0/0;
            # the reference would die at its missing model.pt, this package refuses to write a bitstream
            with pytest.raises(ops._lib.LbdrnError):
                encode.main(["-i", src, "-o", out] + flags)
            continue
        assert encode.main(["-i", src, "-o", out] + flags) == 0, tag
        sub = [d for d in os.listdir(out)][0]
        binp = os.path.join(out, sub, f"img{it}.bin")
        raw = open(binp, "rb").read()
        n, sr_, w_, h_, K_, bc_, nl_, D_, nn, base = container.unpack_header(raw)
        assert (sr_, w_, h_, K_, D_) == (sr, W, H, K, D), tag
        assert decode.main(["-i", binp]) == 0, tag
        rec = raster_io.read_raster(os.path.join(out, sub, f"img{it}_recon.tif")).reshape(C, H, W)
        assert np.array_equal(rec >> K, img >> K), tag
        # the oracle decodes every tile's payloads to the same pixels
        ocfg = O.FeatCfg()
        off = n
        wins = list(tile_windows(W, H, sr)) if sr > 1 else [(0, 0, 0, 0, W, H)]
        for t, (_, _, x0, y0, w, h) in enumerate(wins):
            params = container.decode_weights(raw[off:off + nn[t]])
            msb = container.decode_base(raw[off + nn[t]:off + nn[t] + base[t]])
            off += nn[t] + base[t]
            assert np.array_equal(msb, img[:, y0:y0 + h, x0:x0 + w] >> K), tag
            assert msb.dtype == (np.uint8 if int(msb.max()) <= 255 else np.uint16), tag
            if int(msb.max()) > 0:
                ro = O.decode(msb.astype(np.uint16), K, D, ocfg, params, bc_, nl_)
                assert np.array_equal(rec[:, y0:y0 + h, x0:x0 + w], ro), tag
        assert off == len(raw), tag
        assert decode.main(["-i", binp, "-org", src]) == 0    # decode.txt has no bpsp yet: runs again with metrics
        log = open(os.path.join(out, sub, "decode.txt")).read()
        mse = float(re.search(r"MSE: (\S+)", log).group(1))
        true = float(np.mean((img.astype(np.float32) - rec.astype(np.float32)) ** 2))
        assert abs(mse - true) <= 1e-4 * max(true, 1e-9), tag
