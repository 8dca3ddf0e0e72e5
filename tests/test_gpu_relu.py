"""The hidden activation the reference names as its alternative -- `activation=torch.nn.ReLU()`, the commented-out argument
at ref encode.py:75 and decode.py:108 -- through the C ABI (lbdrn_net.act = LBDRN_ACT_RELU: the generic LDS-tiled
kernels): bit-exact against the oracle where the result is integers or canonical float32, within the training tolerance
against the reference's own model / loss / Adam (tests/golden/make_golden_relu.py)."""
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


def test_the_fused_kernels_refuse_and_auto_takes_the_generic_path(dev):
    rng = np.random.default_rng(4)
    C, H, W, K, D = 8, 40, 56, 5, 2
    img = (rng.integers(0, 10000, (C, H, W))).astype(np.uint16)
    img_d = ops.to_device_u16(img, dev)
    msb_d, mx = ops.split_bits(img_d, K)
    geom = ops.FeatureGeometry(C, H, W, K, D, mx, RELU, dev)
    net = ops.make_net(geom.F, 64, C, 2, ops.ACT_RELU)
    p = torch.from_numpy((rng.standard_normal(ops.param_count(net)) * 0.05).astype(np.float32)).to(dev)
    with pytest.raises(ops._lib.LbdrnError) as e:
        ops.decode_fused(geom, net, msb_d, p, path=ops._lib.PATH_MFMA)
    assert e.value.code == ops._lib.E_UNSUPPORTED
    a = ops.decode_fused(geom, net, msb_d, p, path=ops.PATH_AUTO)
    b = ops.decode_fused(geom, net, msb_d, p, path=ops._lib.PATH_GENERIC)
    assert torch.equal(a, b)
    bad = ops.make_net(geom.F, 64, C, 2, 7)
    with pytest.raises(ops._lib.LbdrnError):
        ops.decode_fused(geom, bad, msb_d, p)


def test_a_whole_fit_and_the_clis_with_relu(dev, tmp_path, monkeypatch):
    """codec.fit_device end to end (generic kernels: row build, epochs, evaluation passes, best epoch) equals the oracle run
    step by step on the same permutations; then encode.py / decode.py with constants.HIDDEN_ACTIVATION = "relu"."""
    import constants
    import decode
    import encode
    from lbdrn_hip import raster_io
    from lbdrn_hip.synth import synthetic_tile
    C, H, W, K, D, bc, nl, bs, epochs = 4, 48, 40, 5, 2, 32, 2, 512, 3
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
    assert encode.main(["-i", src, "-o", out, "-e", "2", "-bs", "256", "-bc", "32", "-sr", "2"]) in (0, None)   # (four tiles: fit_many)
    sub = [d for d in (tmp_path / "o").iterdir() if d.is_dir()][0]
    assert decode.main(["-i", str(sub / "t.bin")]) in (0, None)      # (without -org: the raster stays for the checks)
    rec = raster_io.read_raster(str(sub / "t_recon.tif"))
    assert np.array_equal(rec >> K, img >> K)
    # the same stream under the default activation decodes to other low bits: the switch is not in the bitstream (as in
    # the reference, where both sides edit the same source line)
    monkeypatch.setattr(constants, "HIDDEN_ACTIVATION", "sine")
    assert decode.main(["-i", str(sub / "t.bin")]) in (0, None)
    rec2 = raster_io.read_raster(str(sub / "t_recon.tif"))
    assert np.array_equal(rec2 >> K, img >> K) and not np.array_equal(rec2, rec)
