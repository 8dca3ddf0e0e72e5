"""Generate tests/golden/bands4.npz and tests/golden/rasters_learn_bands4.npz by RUNNING the reference's own modules
(authoring container only; needs /root/reference).  Round-6 additions; the older fixtures are untouched.

The reference's majority shape: 9 of the 13 images of run.sh:14-28 (five GF-2, four GF6-PMS) have FOUR bands, so
K5 D2 bc64 nl2 gives F = 4 * 25 = 100 features (96 that can differ from zero: the window centres are exact zeros,
LBDRNdataset.py:126-128).  Until round 6 no fixture held that shape to the reference.

bands4.npz
  small/*    a 4 x 24 x 20 image: features / labels from the reference's LBDRNdataset.process() (C = 4, D = 2, relative
             colours), the reference LBDRNModel(100, 64, 4, 2) under seed 19920517 -- initial parameters, forward on all
             480 rows, forward of a "wide" copy whose sin arguments wrap many times --, and six teacher-forced updates of
             96-row minibatches with the reference's LBDRNLoss, torch.optim.Adam and StepLR exactly as encode.py:84-86
             builds them (modified_ignite_engine.py:18-27 replayed): gradient of the first step, loss, learning rate and
             parameters after every step, the Adam moments at the end;
  ragged/*   a 4 x 40 x 52 image (2080 rows), six teacher-forced updates of 200-row minibatches (three whole 64-row
             groups and one of eight rows: a parity of k_train_split's pair half empty), same optimiser chain; the
             feature matrix is not stored (the oracle's is bit-identical to process(): test_oracle_golden).

rasters_learn_bands4.npz -- as rasters_learn_bc64.npz (make_golden_round3.py) at the 4-band shape: a smooth 4 x 256 x 256
  image whose low bits can be learnt, 304 real torch Adam steps (38 epochs x 8 minibatches of 8192, encode.py:84-98
  schedule), weights truncated to 16 bits, decode.py:122-134 replayed; the residual plane and the list of sub-pixels
  within 31e-5 of a rounding boundary.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import make_golden_round3 as R3  # noqa: E402  (smooth(), process())

SEED = MG.SEED
NEAR = 31e-5


def teacher_forced(LBDRNModel, loss_fn, f, l, C, rows, gseed, out, tag, keep_grads):
    F = f.shape[1]
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=F, dim_hidden=64, dim_out=C, num_layers=2)            # encode.py:71-77
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)                               # encode.py:84
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(3 / 3)), gamma=0.1)
    g = torch.Generator().manual_seed(gseed)
    out[tag + "/params0"] = MG._flat(m.state_dict())
    nsteps = 6
    batches = np.stack([torch.randperm(f.shape[0], generator=g)[:rows].numpy() for _ in range(nsteps)])
    out[tag + "/batches"] = batches.astype(np.int64)
    for s in range(nsteps):
        x, t = torch.from_numpy(f[batches[s]]), torch.from_numpy(l[batches[s]])
        opt.zero_grad()
        m.train()
        loss = loss_fn(m(x), t)                                                   # modified_ignite_engine.py:18-27
        loss.backward()
        if keep_grads and s == 0:
            out[tag + "/step0/grads"] = np.concatenate([p.grad.numpy().reshape(-1) for p in m.parameters()])
        out[f"{tag}/step{s}/lr"] = np.float64(opt.param_groups[0]["lr"])
        opt.step()
        out[f"{tag}/step{s}/loss"] = np.float32(loss.item())
        out[f"{tag}/step{s}/params"] = MG._flat(m.state_dict())
        if s % 2 == 1:
            sched.step()   # an "epoch" of two iterations: encode.py:98
    out[tag + "/exp_avg"] = np.concatenate([opt.state[p]["exp_avg"].numpy().reshape(-1) for p in m.parameters()])
    out[tag + "/exp_avg_sq"] = np.concatenate([opt.state[p]["exp_avg_sq"].numpy().reshape(-1) for p in m.parameters()])
    return m


def main():
    sys.path.insert(0, MG.REF)
    MG._install_standins()
    import LBDRNdataset as RD
    from LBDRNloss import LBDRNLoss
    from LBDRNmodel import LBDRNModel
    loss_fn = LBDRNLoss()
    K, D, C = 5, 2, 4
    out = {}

    # ------------------------------------------------------------ small: features, forward, six steps
    img = MG._img(41, C, 24, 20)
    f, l = R3.process(RD, img, K, D, {})
    assert f.shape == (480, 100) and l.shape == (480, C)
    out["small/img"] = img
    out["small/features"] = f
    out["small/labels"] = l
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=100, dim_hidden=64, dim_out=C, num_layers=2)
    with torch.no_grad():
        out["small/init_params"] = MG._flat(m.state_dict())
        out["small/init_y"] = m(torch.from_numpy(f)).numpy()
        g = torch.Generator().manual_seed(7)
        for p in m.parameters():
            p.mul_(1.0 + 4.0 * torch.rand(p.shape, generator=g))
        m.net[0].linear.weight.mul_(6.0)
        out["small/wide_params"] = MG._flat(m.state_dict())
        out["small/wide_y"] = m(torch.from_numpy(f)).numpy()
    teacher_forced(LBDRNModel, loss_fn, f, l, C, 96, 43, out, "small", True)

    # ------------------------------------------------------------ ragged: 200-row minibatches of a 2080-row image
    img2 = MG._img(42, C, 40, 52)
    f2, l2 = R3.process(RD, img2, K, D, {})
    out["ragged/img"] = img2
    teacher_forced(LBDRNModel, loss_fn, f2, l2, C, 200, 44, out, "ragged", False)
    path = os.path.join(HERE, "bands4.npz")
    np.savez_compressed(path, **out)
    print("bands4.npz", os.path.getsize(path), "bytes")

    # ------------------------------------------------------------ a learnable 4 x 256 x 256 raster
    H = W = 256
    img = R3.smooth(4001, C, H, W)
    f, l = R3.process(RD, img, K, D, {})
    N, F = f.shape
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=F, dim_hidden=64, dim_out=C, num_layers=2)            # encode.py:71-77
    epochs, bs = 38, 8192
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)                               # encode.py:84
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(epochs / 3)), gamma=0.1)
    g = torch.Generator().manual_seed(100 + 4001)
    ft, lt = torch.from_numpy(f), torch.from_numpy(l)
    steps, epoch_loss = 0, []
    for e in range(epochs):
        perm = torch.randperm(N, generator=g)
        acc = 0.0
        for s in range(0, N, bs):
            idx = perm[s:s + bs]
            opt.zero_grad()
            m.train()
            loss = loss_fn(m(ft[idx]), lt[idx])                                   # modified_ignite_engine.py:18-27
            loss.backward()
            opt.step()
            steps += 1
            acc += float(loss.item())
        epoch_loss.append(acc / ((N + bs - 1) // bs))
        sched.step()                                                              # encode.py:98
    flat = MG._flat(m.state_dict())
    flat = (flat.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)         # "precision=16" model
    sd, k = {}, 0
    for name, val in m.state_dict().items():                                      # decode.py:114-120
        sd[name] = torch.from_numpy(flat[k:k + val.numel()].reshape(val.shape).copy())
        k += val.numel()
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():                                                         # decode.py:122-134
        z0 = ft @ m.net[0].linear.weight.T + m.net[0].linear.bias
        h0 = torch.sin(30.0 * z0)
        z1 = h0 @ m.net[1].linear.weight.T + m.net[1].linear.bias
        y_pred = m(ft)
        residual = torch.round(y_pred * (2 ** K - 1)).numpy()
    base = (img >> K).astype(np.uint16)
    res_chw = np.transpose(residual.reshape(H, W, C), (2, 0, 1))
    image = np.round((base << K).astype(np.float32) + res_chw).astype(np.uint16)
    assert np.array_equal(image >> K, base)
    t = y_pred.numpy().astype(np.float64) * (2 ** K - 1)
    dist = np.abs(t - (np.floor(t) + 0.5))
    near = np.flatnonzero(dist.reshape(-1) < NEAR)
    mse = float(np.mean((img.astype(np.float32) - image.astype(np.float32)) ** 2))
    mid = float(np.mean((img.astype(np.float32) - ((base << K) + 16).astype(np.float32)) ** 2))
    res = (image - (base << K)).astype(np.uint8)
    o = dict(img=img, K=np.int64(K), D=np.int64(D), bc=np.int64(64), nl=np.int64(2),
             flags=np.array([0, 0, 1, 1], np.int64), params=flat, residual=res, adam_steps=np.int64(steps),
             epochs=np.int64(epochs), bs=np.int64(bs), epoch_loss=np.array(epoch_loss, np.float32),
             near_idx=near.astype(np.int64), near_dist=dist.reshape(-1)[near].astype(np.float32),
             near_threshold=np.float64(NEAR), mse=np.float32(mse), mse_midrange=np.float32(mid),
             psnr=np.float64(10 * np.log10(10000 ** 2 / mse)), psnr_midrange=np.float64(10 * np.log10(10000 ** 2 / mid)),
             z_absmax30=np.array([30 * float(z0.abs().max()), 30 * float(z1.abs().max())], np.float32),
             y_sample=y_pred.numpy()[::499].copy())
    path = os.path.join(HERE, "rasters_learn_bands4.npz")
    np.savez_compressed(path, **o)
    hist = np.bincount(res.reshape(-1), minlength=32)
    print("bands4 raster: F", F, "steps", steps, "loss", epoch_loss[0], "->", epoch_loss[-1], "PSNR", o["psnr"],
          "vs mid-range", o["psnr_midrange"], "|30z| max", o["z_absmax30"], "residual values used", int((hist > 0).sum()),
          "near", near.size, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
