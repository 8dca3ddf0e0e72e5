"""Generate tests/golden/train2.npz and tests/golden/rasters_learn_*.npz by RUNNING the reference's own modules
(authoring container only; needs /root/reference).  Round-3 additions; the older fixtures are untouched.

train2.npz -- training fixtures for the kernels that run BASELINE.json configs[2] and configs[4]:
  wide256/*   the three teacher-forced bc = 256 Adam steps of wide_net.npz (same seeds, same batches: the generator
              asserts the final parameters equal that file's) with what it did not keep: the losses per step, the
              Adam moments after the last step and the first step's gradient, so that the fused bc = 256 kernel can be
              held to moment bounds too;
  embed/*     an 8 x 24 x 20 image under USE_COORDINATES + EMBEDDING (F = 250): features / labels from the
              reference's LBDRNdataset.process(), six teacher-forced updates of 96-row minibatches with the
              reference's LBDRNModel / LBDRNLoss, torch.optim.Adam and StepLR exactly as encode.py:84-86 builds them
              (modified_ignite_engine.py:18-27 replayed): loss, learning rate and parameters after every step, the
              moments at the end.

rasters_learn_{bc64,bc256,embed}.npz -- reference-made rasters on images whose low bits CAN be learnt: a smooth
8 x 128 x 128 uint16 image (low-frequency sinusoids, slopes of a few counts per pixel, noise 1.5 counts), 304 real
torch Adam steps (152 epochs x 2 minibatches of 8192, encode.py:84-98 schedule with step_size = epochs / 3), weights
truncated to 16 bits, decode.py:122-134 replayed.  The residual planes span 0..31 and |30 z| of the first layer
reaches ~10, where the older rasters sit at y ~ 0.5; a fourth case, "scaled", decodes the bc = 64 network with its
first layer multiplied by 400 (|30 z| in the hundreds: the range reduction of sin at raster scale, on reference-made
output).  Stored as in rasters_*.npz (image, truncated weights,
residual plane, near-boundary sub-pixels), plus the per-epoch training loss, the PSNR of the reconstruction and of
"predict mid-range" (the gain a fit must reproduce), and max |30 z| per layer.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

SEED = MG.SEED
NEAR = 31e-5
DEFAULT = dict(USE_COORDINATES=False, EMBEDDING=False, USE_COLORS=True, RELATIVE=True)


def smooth(seed, C, H, W, sigma=1.5):
    """Smooth bands: the five low bits of a pixel follow from where the high bits of its neighbours step."""
    rng = np.random.default_rng(seed)
    yy = np.arange(H)[:, None] / H
    xx = np.arange(W)[None, :] / W
    out = np.empty((C, H, W), np.uint16)
    for c in range(C):
        a = np.zeros((H, W))
        for _ in range(3):
            a += rng.uniform(0.4, 1.0) * np.sin(2 * np.pi * (rng.uniform(0.2, 1.2) * yy + rng.uniform(0.2, 1.2) * xx)
                                               + rng.uniform(0, 6.28))
        a = (a - a.min()) / (a.max() - a.min())
        a = 1000 + a * 1800 + rng.normal(0, sigma, (H, W))     # ~14 counts per pixel at most: MSB steps every 2-8 px
        out[c] = np.clip(np.rint(a), 0, 10000).astype(np.uint16)
    return out


def process(RD, img, K, D, flags):
    for k, v in {**DEFAULT, **flags}.items():
        setattr(RD, k, v)
    MG._RASTERS["mem.tif"] = img
    f, l = RD.process("mem.tif", K, D, "mem_base.tif")
    for k, v in DEFAULT.items():
        setattr(RD, k, v)
    return np.ascontiguousarray(f, np.float32), np.ascontiguousarray(l, np.float32)


def main():
    sys.path.insert(0, MG.REF)
    MG._install_standins()
    import LBDRNdataset as RD
    from LBDRNloss import LBDRNLoss
    from LBDRNmodel import LBDRNModel
    loss_fn = LBDRNLoss()
    out = {}

    # ------------------------------------------------------------ wide256: moments of wide_net.npz's three steps
    feats = np.load(os.path.join(HERE, "features.npz"))
    wide = np.load(os.path.join(HERE, "wide_net.npz"))
    fA, lA = feats["A_K5_D2/features"], feats["A_K5_D2/labels"]
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=200, dim_hidden=256, dim_out=8, num_layers=2)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    assert np.array_equal(MG._flat(m.state_dict()), wide["train256/params0"])
    batches = wide["train256/batches"]
    for s in range(3):
        x, t = torch.from_numpy(fA[batches[s]]), torch.from_numpy(lA[batches[s]])
        opt.zero_grad()
        m.train()
        loss = loss_fn(m(x), t)
        loss.backward()
        opt.step()
        assert np.float32(loss.item()) == wide[f"train256/step{s}/loss"]
    assert np.array_equal(MG._flat(m.state_dict()), wide["train256/params_final"])
    out["wide256/exp_avg"] = np.concatenate([opt.state[p]["exp_avg"].numpy().reshape(-1) for p in m.parameters()])
    out["wide256/exp_avg_sq"] = np.concatenate([opt.state[p]["exp_avg_sq"].numpy().reshape(-1) for p in m.parameters()])

    # ------------------------------------------------------------ embed: six teacher-forced steps at F = 250
    img = MG._img(21, 8, 24, 20)
    flags = dict(USE_COORDINATES=True, EMBEDDING=True)
    fE, lE = process(RD, img, 5, 2, flags)
    assert fE.shape[1] == 250
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=250, dim_hidden=64, dim_out=8, num_layers=2)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(3 / 3)), gamma=0.1)
    g = torch.Generator().manual_seed(31)
    out["embed/img"] = img
    out["embed/flags"] = np.array([1, 1, 1, 1], np.int64)
    out["embed/params0"] = MG._flat(m.state_dict())
    nsteps = 6
    batches = np.stack([torch.randperm(fE.shape[0], generator=g)[:96].numpy() for _ in range(nsteps)])
    out["embed/batches"] = batches.astype(np.int64)
    for s in range(nsteps):
        x, t = torch.from_numpy(fE[batches[s]]), torch.from_numpy(lE[batches[s]])
        opt.zero_grad()
        m.train()
        loss = loss_fn(m(x), t)
        loss.backward()
        out[f"embed/step{s}/lr"] = np.float64(opt.param_groups[0]["lr"])
        opt.step()
        out[f"embed/step{s}/loss"] = np.float32(loss.item())
        out[f"embed/step{s}/params"] = MG._flat(m.state_dict())
        if s % 2 == 1:
            sched.step()   # an "epoch" of two iterations: encode.py:98
    out["embed/exp_avg"] = np.concatenate([opt.state[p]["exp_avg"].numpy().reshape(-1) for p in m.parameters()])
    out["embed/exp_avg_sq"] = np.concatenate([opt.state[p]["exp_avg_sq"].numpy().reshape(-1) for p in m.parameters()])
    np.savez_compressed(os.path.join(HERE, "train2.npz"), **out)
    print("train2.npz", os.path.getsize(os.path.join(HERE, "train2.npz")), "bytes")

    # ------------------------------------------------------------ learnable rasters
    K, D, C, H, W = 5, 2, 8, 128, 128
    cases = {"bc64": (64, 2, {}, 3001), "bc256": (256, 2, {}, 3002),
             "embed": (64, 2, dict(USE_COORDINATES=True, EMBEDDING=True), 3003)}
    for tag, (bc, nl, flags, iseed) in cases.items():
        img = smooth(iseed, C, H, W)
        f, l = process(RD, img, K, D, flags)
        N, F = f.shape
        torch.manual_seed(SEED)
        m = LBDRNModel(dim_in=F, dim_hidden=bc, dim_out=C, num_layers=nl)      # encode.py:71-77
        epochs, bs = 152, 8192
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)                         # encode.py:84
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(epochs / 3)), gamma=0.1)
        g = torch.Generator().manual_seed(100 + iseed)
        ft, lt = torch.from_numpy(f), torch.from_numpy(l)
        steps, epoch_loss = 0, []
        for e in range(epochs):
            perm = torch.randperm(N, generator=g)
            acc = 0.0
            for s in range(0, N, bs):
                idx = perm[s:s + bs]
                opt.zero_grad()
                m.train()
                loss = loss_fn(m(ft[idx]), lt[idx])                             # modified_ignite_engine.py:18-27
                loss.backward()
                opt.step()
                steps += 1
                acc += float(loss.item())
            epoch_loss.append(acc / ((N + bs - 1) // bs))
            sched.step()                                                        # encode.py:98
        flat = MG._flat(m.state_dict())
        flat = (flat.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)   # "precision=16" model
        sd, k = {}, 0
        for name, val in m.state_dict().items():                                # decode.py:114-120
            sd[name] = torch.from_numpy(flat[k:k + val.numel()].reshape(val.shape).copy())
            k += val.numel()
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():                                                   # decode.py:122-134
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
        o = dict(img=img, K=np.int64(K), D=np.int64(D), bc=np.int64(bc), nl=np.int64(nl),
                 flags=np.array([int({**DEFAULT, **flags}[k]) for k in
                                 ("USE_COORDINATES", "EMBEDDING", "USE_COLORS", "RELATIVE")], np.int64),
                 params=flat, residual=res, adam_steps=np.int64(steps), epochs=np.int64(epochs), bs=np.int64(bs),
                 epoch_loss=np.array(epoch_loss, np.float32), near_idx=near.astype(np.int64),
                 near_dist=dist.reshape(-1)[near].astype(np.float32), near_threshold=np.float64(NEAR),
                 mse=np.float32(mse), mse_midrange=np.float32(mid),
                 psnr=np.float64(10 * np.log10(10000 ** 2 / mse)), psnr_midrange=np.float64(10 * np.log10(10000 ** 2 / mid)),
                 z_absmax30=np.array([30 * float(z0.abs().max()), 30 * float(z1.abs().max())], np.float32),
                 y_sample=y_pred.numpy()[::499].copy())
        path = os.path.join(HERE, f"rasters_learn_{tag}.npz")
        np.savez_compressed(path, **o)
        if tag == "bc64":   # the same network with a 400 x first layer: a decode-only case far out in sin's argument range
            with torch.no_grad():
                m.net[0].linear.weight.mul_(400.0)
                m.net[0].linear.bias.mul_(400.0)
                flat2 = MG._flat(m.state_dict())
                flat2 = (flat2.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
                sd, k = {}, 0
                for name, val in m.state_dict().items():
                    sd[name] = torch.from_numpy(flat2[k:k + val.numel()].reshape(val.shape).copy())
                    k += val.numel()
                m.load_state_dict(sd)
                z0 = ft @ m.net[0].linear.weight.T + m.net[0].linear.bias
                y2 = m(ft)
                r2 = torch.round(y2 * (2 ** K - 1)).numpy()
            img2 = np.round((base << K).astype(np.float32) + np.transpose(r2.reshape(H, W, C), (2, 0, 1))).astype(np.uint16)
            t2 = y2.numpy().astype(np.float64) * (2 ** K - 1)
            d2 = np.abs(t2 - (np.floor(t2) + 0.5))
            n2 = np.flatnonzero(d2.reshape(-1) < NEAR)
            o2 = dict(o, params=flat2, residual=(img2 - (base << K)).astype(np.uint8), near_idx=n2.astype(np.int64),
                      near_dist=d2.reshape(-1)[n2].astype(np.float32), z_absmax30=np.array([30 * float(z0.abs().max()), 0], np.float32),
                      y_sample=y2.numpy()[::499].copy())
            np.savez_compressed(os.path.join(HERE, "rasters_learn_scaled.npz"), **o2)
            print("scaled: |30 z0| max", 30 * float(z0.abs().max()), "near", n2.size, "residual min/max",
                  int(o2["residual"].min()), int(o2["residual"].max()))
        hist = np.bincount(res.reshape(-1), minlength=32)
        print(tag, "F", F, "steps", steps, "loss", epoch_loss[0], "->", epoch_loss[-1], "PSNR", o["psnr"], "vs mid-range",
              o["psnr_midrange"], "|30z| max", o["z_absmax30"], "residual values used", int((hist > 0).sum()),
              "min/max", int(res.min()), int(res.max()), "near", near.size, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
