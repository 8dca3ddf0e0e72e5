"""Generate tests/golden/rasters_*.npz by RUNNING the reference's own modules (authoring container only; needs
/root/reference): reference-made reconstructed rasters big enough for "bit-exact" to mean something.

For three BASELINE.json shapes -- configs[1] (bc=64 nl=2), configs[2] (bc=256 nl=2) and configs[4]
(USE_COORDINATES + EMBEDDING, F=250) -- on an 8 x 256 x 256 uint16 image (524,288 sub-pixels):
  * features / labels from the reference's LBDRNdataset.process() (in-memory gdal stand-in, make_golden.py);
  * a reference LBDRNModel fitted by 56 real torch Adam steps (7 epochs x 8 shuffled minibatches of 8192 rows:
    modified_ignite_engine.py:18-27 replayed with the reference's model and LBDRNLoss, Adam/StepLR as
    encode.py:84-86 builds them);
  * the weights with their low 16 bits cleared (this repo's model of fpzip precision=16);
  * decode.py:122-134 replayed with the reference model on those weights -> the integer raster.
Stored (data only): the image, K, D, the constants flags, the truncated weights, the reference's low-bit residual
plane (raster = (img >> K << K) + residual), and the flat indices + distances of every sub-pixel whose y*(2^K-1)
lies within 31e-5 of a .5 rounding boundary (= 1e-5 in y): the only places where an implementation with a
different float32 summation order / sin implementation may legitimately round the other way.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stand-ins + image generator)

SEED = MG.SEED
NEAR = 31e-5    # in units of y*(2^K-1) with K=5: 1e-5 in y


def synth(seed, C, H, W, sigma=40.0):
    """bench.py's synthetic tile recipe at fixture size: low-frequency sinusoids in [500, 9500] + N(0, 40^2)."""
    rng = np.random.default_rng(seed)
    yy = np.arange(H)[:, None] / H
    xx = np.arange(W)[None, :] / W
    out = np.empty((C, H, W), np.uint16)
    for c in range(C):
        a = np.zeros((H, W))
        for _ in range(6):
            a += rng.uniform(0.2, 1.0) * np.sin(2 * np.pi * (rng.uniform(0.3, 4) * yy + rng.uniform(0.3, 4) * xx)
                                               + rng.uniform(0, 6.28))
        a = (a - a.min()) / (a.max() - a.min())
        a = 500 + a * 9000 + rng.normal(0, sigma, (H, W))
        out[c] = np.clip(np.rint(a), 0, 10000).astype(np.uint16)
    return out


def main():
    sys.path.insert(0, MG.REF)
    MG._install_standins()
    import LBDRNdataset as RD
    from LBDRNloss import LBDRNLoss
    from LBDRNmodel import LBDRNModel

    DEFAULT = dict(USE_COORDINATES=False, EMBEDDING=False, USE_COLORS=True, RELATIVE=True)
    K, D, C, H, W = 5, 2, 8, 256, 256
    cases = {
        "bc64": (64, 2, {}, 2001, 40.0),      # bench.py's tile statistics: the low bits are noise, y clusters at 0.5
        "bc256": (256, 2, {}, 2002, 2.0),      # smooth image: the low bits are learnable, the fit moves the weights
        "embed": (64, 2, dict(USE_COORDINATES=True, EMBEDDING=True), 2003, 6.0),
    }
    for tag, (bc, nl, flags, iseed, sigma) in cases.items():
        img = synth(iseed, C, H, W, sigma)
        for k, v in {**DEFAULT, **flags}.items():
            setattr(RD, k, v)
        MG._RASTERS["mem.tif"] = img
        f, l = RD.process("mem.tif", K, D, "mem_base.tif")
        for k, v in DEFAULT.items():
            setattr(RD, k, v)
        f, l = np.ascontiguousarray(f, np.float32), np.ascontiguousarray(l, np.float32)
        N, F = f.shape
        torch.manual_seed(SEED)
        m = LBDRNModel(dim_in=F, dim_hidden=bc, dim_out=C, num_layers=nl)      # encode.py:71-77
        epochs, bs = 7, 8192
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)                         # encode.py:84
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(epochs / 3)), gamma=0.1)
        loss_fn = LBDRNLoss()
        g = torch.Generator().manual_seed(100 + iseed)
        ft, lt = torch.from_numpy(f), torch.from_numpy(l)
        steps = 0
        for e in range(epochs):
            perm = torch.randperm(N, generator=g)
            for s in range(0, N, bs):
                idx = perm[s:s + bs]
                opt.zero_grad()
                m.train()
                loss = loss_fn(m(ft[idx]), lt[idx])                             # modified_ignite_engine.py:18-27
                loss.backward()
                opt.step()
                steps += 1
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
            y_pred = m(ft)
            residual = torch.round(y_pred * (2 ** K - 1)).numpy()
        base = (img >> K).astype(np.uint16)
        res_chw = np.transpose(residual.reshape(H, W, C), (2, 0, 1))
        image = np.round((base << K).astype(np.float32) + res_chw).astype(np.uint16)
        assert np.array_equal(image >> K, base)
        # distance of y*(2^K-1) to the nearest .5 boundary, in float64 from the reference's float32 y
        t = y_pred.numpy().astype(np.float64) * (2 ** K - 1)
        dist = np.abs(t - (np.floor(t) + 0.5))                                  # [N, C]
        near = np.flatnonzero(dist.reshape(-1) < NEAR)                          # flat index n*C + c
        mse = float(np.mean((img.astype(np.float32) - image.astype(np.float32)) ** 2))
        out = dict(img=img, K=np.int64(K), D=np.int64(D), bc=np.int64(bc), nl=np.int64(nl),
                   flags=np.array([int({**DEFAULT, **flags}[k]) for k in
                                   ("USE_COORDINATES", "EMBEDDING", "USE_COLORS", "RELATIVE")], np.int64),
                   params=flat, residual=(image - (base << K)).astype(np.uint8), adam_steps=np.int64(steps),
                   final_loss=np.float32(loss.item()), near_idx=near.astype(np.int64),
                   near_dist=dist.reshape(-1)[near].astype(np.float32), near_threshold=np.float64(NEAR),
                   mse=np.float32(mse), y_sample=y_pred.numpy()[::997].copy())
        np.savez_compressed(os.path.join(HERE, f"rasters_{tag}.npz"), **out)
        print(tag, "F", F, "steps", steps, "loss", float(loss.item()), "mse", mse, "near-boundary sub-pixels", near.size,
              "of", dist.size, os.path.getsize(os.path.join(HERE, f"rasters_{tag}.npz")), "bytes")


if __name__ == "__main__":
    main()
