"""The hidden activation the reference names as its alternative -- `activation=torch.nn.ReLU()`, the commented-out argument
at ref encode.py:75 and decode.py:108 -- through the C ABI (lbdrn_net.act = LBDRN_ACT_RELU: since round 6 a template
parameter of the fused kernels k_apply_mfma / k_train_stream / k_train_split at bc <= 128 / bc = 64, the generic LDS-tiled
kernels elsewhere): bit-exact against the oracle where the result is integers or canonical float32, within the training
tolerance against the reference's own model / loss / Adam (tests/golden/make_golden_relu.py)."""
import os

import numpy as np
import pytest
import torch

import oracle as O
from lbdrn_hip import codec, ops, sampler
from lbdrn_hip.features import FeatCfg
from lbdrn_hip.model import LBDRNModel

pytestmark = pytest.mark.gpu

RTOL_TRAIN = 1e-5
RELU = FeatCfg(activation="relu")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_forward_bit_exact_vs_oracle_and_within_tolerance_of_the_reference(golden, dev):
    G = golden["relu_net"]
    x = torch.from_numpy(G["x"]).to(dev)
    for tag, bc, nl in (("bc64_nl2", 64, 2), ("bc32_nl3", 32, 3), ("bc256_nl1", 256, 1)):
        p = G[f"fwd/{tag}/params"]
        y = ops.forward(ops.make_net(200, bc, 8, nl, ops.ACT_RELU), torch.from_numpy(p).to(dev), x).cpu().numpy()
        with O.hidden_activation("relu"):
            yo = O.forward(p, 200, bc, 8, nl, G["x"])
        assert np.array_equal(_bits(y), _bits(yo)), tag
        np.testing.assert_allclose(y, G[f"fwd/{tag}/y"], rtol=3e-6, atol=3e-7)
        # and the same parameters under the default activation are another function
        ys = ops.forward(ops.make_net(200, bc, 8, nl), torch.from_numpy(p).to(dev), x).cpu().numpy()
        assert np.abs(ys - y).max() > 0.05


def test_drop_in_model_with_relu_runs_the_hip_forward(golden, dev):
    """LBDRNModel(activation=torch.nn.ReLU()) -- the reference's constructor call with the commented-out argument put
    back -- answers from the HIP kernels; any other custom module still raises."""
    G = golden["relu_net"]
    m = LBDRNModel(dim_in=200, dim_hidden=64, dim_out=8, num_layers=2, activation=torch.nn.ReLU())
    p, k, sd = G["fwd/bc64_nl2/params"], 0, {}
    for name, val in m.state_dict().items():
        sd[name] = torch.from_numpy(p[k:k + val.numel()].reshape(val.shape).copy())
        k += val.numel()
    m.load_state_dict(sd)
    y = m(torch.from_numpy(G["x"]).to(dev)).cpu().numpy()
    np.testing.assert_allclose(y, G["fwd/bc64_nl2/y"], rtol=3e-6, atol=3e-7)
    with pytest.raises(NotImplementedError):
        LBDRNModel(200, 64, 8, 2, activation=torch.nn.Tanh())(torch.from_numpy(G["x"]).to(dev))


def test_three_updates_vs_reference_and_oracle(golden, dev):
    G = golden["relu_net"]
    x, t = torch.from_numpy(G["x"]).to(dev), torch.from_numpy(G["t"]).to(dev)
    net = ops.make_net(200, 64, 8, 2, ops.ACT_RELU)
    p = torch.from_numpy(G["train/params0"].copy()).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    po = G["train/params0"].copy()
    mo, vo = np.zeros_like(po), np.zeros_like(po)
    for s in range(3):
        b = G["train/batches"][s]
        bd = torch.from_numpy(b).to(dev)
        loss, g = ops.train_step(net, x[bd], t[bd], p, m, v, s + 1, 1e-3)
        with O.hidden_activation("relu"):
            lo, go = O.train_step(po, mo, vo, 200, 64, 8, 2, G["x"][b], G["t"][b], 1e-3, s + 1)
        ref = float(G[f"train/step{s}/loss"])
        assert abs(float(loss.item()) - ref) <= RTOL_TRAIN * ref
        assert abs(float(loss.item()) - lo) <= RTOL_TRAIN * lo
        assert np.linalg.norm(g.cpu().numpy() - go) <= 2e-5 * np.linalg.norm(go)
        if s == 0:
            gr = G["train/step0/grads"]
            assert np.linalg.norm(g.cpu().numpy() - gr) <= 1e-5 * np.linalg.norm(gr)
    pr = G["train/params_final"]
    assert np.linalg.norm(p.cpu().numpy() - pr) <= 2e-5 * np.linalg.norm(pr)


def test_reference_decoded_raster(golden, dev):
    """decode.py:122-134 replayed by the reference on a ReLU network it fitted: the integers of apply_image agree with the
    oracle's bit for bit, and with the reference's except at listed near-boundary sub-pixels."""
    G = golden["relu_net"]
    img = G["raster/img"]
    K, D, bc, nl, _ = (int(v) for v in G["raster/cfg"])
    msb, _, mx = O.split_bits(img, K)
    with O.hidden_activation("relu"):
        oo = O.decode(msb, K, D, O.FeatCfg(), G["raster/params"], bc, nl, mx)
    out = codec.apply_image(img >> K, G["raster/params"], K, D, bc, nl, cfg=RELU, device=str(dev))
    out = np.asarray(out[0] if isinstance(out, tuple) else out).astype(np.uint16)
    assert np.array_equal(out, oo)
    ref = ((img >> K) << K) + G["raster/residual"]
    bad = np.flatnonzero((out != ref).transpose(1, 2, 0).reshape(-1))
    assert np.isin(bad, G["raster/near_idx"]).all() and bad.size <= 2
    # ... and by name on every path: the fused kernel (k_apply_mfma<.., RELU>) gives the oracle's integers too
    C, H, W = img.shape
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, RELU, dev)
    net = ops.make_net(geom.F, bc, C, nl, ops.ACT_RELU)
    p_d, msb_d = torch.from_numpy(G["raster/params"]).to(dev), ops.to_device_u16(msb, dev)
    for path in (ops._lib.PATH_GENERIC, ops._lib.PATH_AUTO, ops._lib.PATH_MFMA):
        assert np.array_equal(ops.from_device_u16(ops.decode_fused(geom, net, msb_d, p_d, path=path)), oo), path


def _rand_params(rng, F, bc, C, nl, gain=1.0):
    parts = []
    for l in range(nl):
        nin = F if l == 0 else bc
        b = (1.0 / nin if l == 0 else np.sqrt(6.0 / nin)) * gain      # (no w0 = 30 in front of a ReLU: weights of that size again)
        parts += [rng.uniform(-b, b, bc * nin), rng.uniform(-b, b, bc)]
    b = np.sqrt(6.0 / bc) * gain
    parts += [rng.uniform(-b, b, C * bc), rng.uniform(-b, b, C)]
    return np.concatenate(parts).astype(np.float32)


@pytest.mark.parametrize("case", [(8, 37, 150, 5, 2, 64, 2, False), (4, 19, 70, 5, 2, 64, 2, False), (3, 21, 33, 3, 1, 32, 1, False),
                                  (5, 9, 129, 5, 2, 128, 2, False), (8, 20, 66, 5, 2, 64, 2, True), (8, 24, 20, 5, 2, 256, 2, False)])
def test_fused_apply_with_relu_bit_exact_vs_oracle(dev, case):
    """k_apply_mfma<NT, MODE, RELU> (bc <= 128): the decode raster, y and the canonical whole-image sum equal the oracle's and
    the generic kernels' bit for bit; the epoch-ranking pass (LBDRN_EVAL_FAST) stays within the 1e-6 its flag promises.
    bc = 256 (k_apply_wide is the Sine network's): LBDRN_PATH_MFMA answers UNSUPPORTED, AUTO takes the generic kernels."""
    C, H, W, K, D, bc, nl, embed = case
    rng = np.random.default_rng(sum(case[:7]))
    cfg = FeatCfg(use_coordinates=embed, embedding=embed, activation="relu")
    ocfg = O.FeatCfg(cfg.use_coordinates, cfg.embedding, cfg.sigma, cfg.n_freq, cfg.use_colors, cfg.relative)
    img = rng.integers(0, 10000, (C, H, W)).astype(np.uint16)
    msb_o, lab_o, mx = O.split_bits(img, K)
    F = cfg.feature_dim(C, D)
    params = _rand_params(rng, F, bc, C, nl)
    with O.hidden_activation("relu"):
        out_o, y_o = O.decode(msb_o, K, D, ocfg, params, bc, nl, mx, want_y=True)
        sse_o = O.eval_sse(msb_o, lab_o, D, ocfg, params, bc, nl, mx)
    assert np.unique(out_o & ((1 << K) - 1)).size > 2          # (the outputs move: not a flat 0.5)
    img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb_o, dev)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
    net = ops.make_net(F, bc, C, nl, ops.ACT_RELU)
    p_d = torch.from_numpy(params).to(dev)
    if bc > 128:
        with pytest.raises(ops._lib.LbdrnError) as e:
            ops.decode_fused(geom, net, msb_d, p_d, path=ops._lib.PATH_MFMA)
        assert e.value.code == ops._lib.E_UNSUPPORTED
        assert torch.equal(ops.decode_fused(geom, net, msb_d, p_d, path=ops.PATH_AUTO),
                           ops.decode_fused(geom, net, msb_d, p_d, path=ops._lib.PATH_GENERIC))
        return
    for path in (ops._lib.PATH_GENERIC, ops._lib.PATH_MFMA, ops.PATH_AUTO):
        out, y = ops.decode_fused(geom, net, msb_d, p_d, want_y=True, path=path)
        assert np.array_equal(_bits(y.cpu().numpy()), _bits(y_o)), (case, path)
        assert np.array_equal(ops.from_device_u16(out), out_o), (case, path)
        sse = float(ops.eval_sse(geom, net, img_d, msb_d, p_d, path=path).item())
        assert abs(sse - sse_o) <= 1e-11 * max(1.0, abs(sse_o)), (case, path)
    fast = float(ops.eval_sse(geom, net, img_d, msb_d, p_d, path=ops._lib.PATH_MFMA, fast=True).item())
    assert abs(fast - sse_o) <= 1e-6 * sse_o
    bad = ops.make_net(F, bc, C, nl, 7)
    with pytest.raises(ops._lib.LbdrnError):
        ops.decode_fused(geom, bad, msb_d, p_d)


@pytest.mark.parametrize("alone", (False, True))
def test_fused_training_with_relu_eats_the_reference_fixture(golden, dev, alone):
    """The fused steps with the ReLU template argument (k_train_stream<48, 2, 3, 12, RELU>; alone: k_train_split<48, 12, RELU>)
    on image A_K5_D2: the fixture's three teacher-forced 160-row minibatches as one epoch through
    lbdrn_train_epoch(LBDRN_PATH_MFMA) -- every loss within 1e-5 of the reference's (LBDRNModel(activation=ReLU()),
    LBDRNLoss, torch.optim.Adam) and of the oracle's, the parameters after the third update within 2e-5."""
    G, Ft = golden["relu_net"], golden["features"]
    img = Ft["A_K5_D2/img"]
    C, H, W = img.shape
    img_d = ops.to_device_u16(img, dev)
    msb_d, mx = ops.split_bits(img_d, 5)
    geom = ops.FeatureGeometry(C, H, W, 5, 2, mx, RELU, dev)
    net = ops.make_net(200, 64, 8, 2, ops.ACT_RELU)
    p = torch.from_numpy(G["train/params0"].copy()).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    batches = G["train/batches"]
    perm = torch.from_numpy(np.concatenate(list(batches))).to(dev)
    losses = torch.zeros(3, dtype=torch.float32, device=dev)
    ws = ops.TrainWorkspace(geom, net, 160, dev).prepare(img_d, msb_d, ops._lib.PATH_MFMA)
    ops.train_epoch(geom, net, img_d, msb_d, perm, 160, p, m, v, 0, 1e-3, losses, ops._lib.PATH_MFMA, ws, alone=alone)
    po = G["train/params0"].copy()
    mo, vo = np.zeros_like(po), np.zeros_like(po)
    for s in range(3):
        with O.hidden_activation("relu"):
            lo, _ = O.train_step(po, mo, vo, 200, 64, 8, 2, G["x"][batches[s]], G["t"][batches[s]], 1e-3, s + 1)
        ref = float(G[f"train/step{s}/loss"])
        got = float(losses[s].item())
        assert abs(got - ref) <= RTOL_TRAIN * ref and abs(got - lo) <= RTOL_TRAIN * lo, (s, got, ref, lo)
    pr = G["train/params_final"]
    assert np.linalg.norm(p.cpu().numpy() - pr) <= 2e-5 * np.linalg.norm(pr)
    assert np.linalg.norm(p.cpu().numpy() - po) <= 2e-5 * np.linalg.norm(po)


@pytest.mark.parametrize("case", [(8, 40, 52, 2, 512, False), (4, 30, 41, 2, 300, False), (4, 30, 41, 1, 300, False),
                                  (8, 24, 36, 2, 400, True), (5, 30, 41, 2, 77, False)])
def test_fused_relu_epoch_matches_generic_and_oracle_whichever_kernel_steps(dev, case):
    """An epoch of the ReLU network on the fused path against the generic kernels and the oracle step by step (loss 1e-5), and
    the same bits with and without the lone-fit hint: k_train_split<.., RELU> and k_train_stream<.., RELU> share one
    summation tree like their Sine instances.  Shapes: F = 200, the 4-band F = 100 (two and one hidden layers), the
    embedding's F = 250, and a loop-schedule shape (F = 125)."""
    C, H, W, nl, bs, embed = case
    K, D, bc = 5, 2, 64
    rng = np.random.default_rng(sum(case[:5]))
    cfg = FeatCfg(use_coordinates=embed, embedding=embed, activation="relu")
    ocfg = O.FeatCfg(cfg.use_coordinates, cfg.embedding, cfg.sigma, cfg.n_freq, cfg.use_colors, cfg.relative)
    from lbdrn_hip.synth import synthetic_tile
    img = synthetic_tile(int(rng.integers(100)), C, H, W)
    msb, lab, mx = O.split_bits(img, K)
    F = cfg.feature_dim(C, D)
    p0 = _rand_params(rng, F, bc, C, nl, gain=0.5)
    perm_np = rng.permutation(H * W).astype(np.int64)
    perm = torch.from_numpy(perm_np).to(dev)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, cfg, dev)
    net = ops.make_net(F, bc, C, nl, ops.ACT_RELU)
    img_d, msb_d = ops.to_device_u16(img, dev), ops.to_device_u16(msb, dev)
    nsteps = (H * W + bs - 1) // bs
    res = {}
    for key, path, alone in (("gen", ops._lib.PATH_GENERIC, False), ("mfma", ops._lib.PATH_MFMA, False), ("lone", ops._lib.PATH_MFMA, True)):
        p = torch.from_numpy(p0.copy()).to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        losses = torch.zeros(nsteps, dtype=torch.float32, device=dev)
        ops.train_epoch(geom, net, img_d, msb_d, perm, bs, p, m, v, 3, 1e-3, losses, path=path, alone=alone)
        res[key] = [t.cpu().numpy() for t in (p, m, v, losses)]
    for a, b in zip(res["mfma"], res["lone"]):
        assert np.array_equal(a.view(np.int32), b.view(np.int32))
    feats = O.features(msb, D, ocfg, mx)
    po, mo, vo = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    lo = []
    with O.hidden_activation("relu"):
        for s in range(nsteps):
            b = perm_np[s * bs:(s + 1) * bs]
            l, _ = O.train_step(po, mo, vo, F, bc, C, nl, feats[b], lab[b], 1e-3, 3 + s + 1)
            lo.append(l)
    for key in ("gen", "mfma"):
        p, m, v, losses = res[key]
        np.testing.assert_allclose(losses, np.array(lo), rtol=1e-5), key
        assert np.linalg.norm(p - po) <= 2e-5 * np.linalg.norm(po), key
        assert np.abs(m - mo).max() <= 2e-5 * np.abs(mo).max(), key
        assert np.abs(v - vo).max() <= 5e-5 * np.abs(vo).max(), key


@pytest.mark.parametrize("bc", (32, 64))
def test_a_whole_fit_and_the_clis_with_relu(dev, tmp_path, monkeypatch, bc):
    """codec.fit_device end to end (bc = 32: the generic kernels; bc = 64: the fused ones at the 4-band shape -- row build,
    epochs, evaluation passes, best epoch) equals the oracle run step by step on the same permutations; then encode.py /
    decode.py with constants.HIDDEN_ACTIVATION = "relu"."""
    import constants
    import decode
    import encode
    from lbdrn_hip import raster_io
    from lbdrn_hip.synth import synthetic_tile
    C, H, W, K, D, nl, bs, epochs = 4, 48, 40, 5, 2, 2, 512, 3
    img = synthetic_tile(11, C, H, W)
    torch.manual_seed(77)
    fit = codec.fit_device(ops.to_device_u16(img, dev), K, D, bc, nl, 1e-3, bs, epochs, cfg=RELU, keep_losses=True)
    torch.cuda.synchronize()
    losses = fit.losses.cpu().numpy()
    assert np.isfinite(losses).all() and fit.net.act == ops.ACT_RELU
    # the same fit by the oracle: same model draw, same permutations (the sampler's, from the same generator state)
    torch.manual_seed(77)
    draws = codec.draw_fit(fit.geom.F, bc, C, nl, epochs, 1)
    msb, lab, mx = O.split_bits(img, K)
    feats = O.features(msb, D, O.FeatCfg(), mx)
    p = draws.params.numpy().copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    lrs = codec.lr_schedule(1e-3, epochs)
    step, best, best_p = 0, None, None
    with O.hidden_activation("relu"):
        for e in range(epochs):
            perm = sampler.permutation(int(draws.train_seeds[e]), H * W).numpy()
            for s in range(0, H * W, bs):
                idx = perm[s:s + bs]
                step += 1
                lo, _ = O.train_step(p, m, v, fit.geom.F, bc, C, nl, feats[idx], lab[idx], lrs[e], step)
                assert abs(float(losses[e, s // bs]) - lo) <= 1e-4 * lo, (e, s)
            sse = O.eval_sse(msb, lab, D, O.FeatCfg(), p, bc, nl, mx)
            if best is None or sse < best:
                best, best_p = sse, p.copy()
    got = fit.best_params.cpu().numpy()
    assert np.linalg.norm(got - best_p) <= 1e-4 * np.linalg.norm(best_p)
    # CLIs
    monkeypatch.setattr(constants, "HIDDEN_ACTIVATION", "relu")
    src = str(tmp_path / "t.npy")
    raster_io.write_raster(src, img)
    out = str(tmp_path / "o")
    assert encode.main(["-i", src, "-o", out, "-e", "2", "-bs", "256", "-bc", str(bc), "-sr", "2"]) in (0, None)   # (four tiles: fit_many)
    sub = [d for d in (tmp_path / "o").iterdir() if d.is_dir()][0]
    assert decode.main(["-i", str(sub / "t.bin")]) in (0, None)      # (without -org: the raster stays for the checks)
    rec = raster_io.read_raster(str(sub / "t_recon.tif"))
    assert np.array_equal(rec >> K, img >> K)
    # the same stream under the DEFAULT constants decodes to the same raster: since round 6 the file says which network it
    # holds (one extension byte behind the reference's header fields, container.pack_header) -- in round 5 this gave other
    # low bits without an error, as it does in the reference where both sides edit the same source line
    from lbdrn_hip import container
    raw = (sub / "t.bin").read_bytes()
    assert container.header_activation(raw) == "relu" and raw[0] == 8 + 7 * 4 + 1
    monkeypatch.setattr(constants, "HIDDEN_ACTIVATION", "sine")
    os.remove(str(sub / "decode.txt"))
    assert decode.main(["-i", str(sub / "t.bin")]) in (0, None)
    rec2 = raster_io.read_raster(str(sub / "t_recon.tif"))
    assert np.array_equal(rec2, rec)
    # a header without the byte (every reference-written file, every default-network file) leaves the choice to constants.py
    legacy = bytes([raw[0] - 1]) + raw[1:raw[0] - 1] + raw[raw[0]:]
    assert container.header_activation(legacy) is None and container.unpack_header(legacy)[1:] == container.unpack_header(raw)[1:]
