"""Generate tests/golden/relu_net.npz by RUNNING the reference's LBDRNModel / LBDRNLoss / LBDRNdataset.process with the
hidden activation the reference names as its alternative -- `activation=torch.nn.ReLU()`, the commented-out argument at
ref encode.py:75 and decode.py:108 -- (authoring container only; needs /root/reference).  Data only:
  * forward: the reference-made feature matrix of fixture A (features.npz), "trained-like" parameters, the outputs;
  * train: three teacher-forced Adam steps of LBDRNModel(200, 64, 8, 2, activation=ReLU) built as encode.py:71-86 builds
    it (modified_ignite_engine.py:18-27 replayed with the reference's own objects): losses, the first step's gradients,
    the parameters after the third step;
  * raster: an 8 x 128 x 128 uint16 image, features / labels from the reference's process(), 240 real Adam steps (lr 1e-2: at the SIREN initialisation a ReLU network moves slowly), the
    weights with their low 16 bits cleared, decode.py:122-134 replayed -> the low-bit residual plane, and the flat
    indices of the sub-pixels within 31e-5 of a rounding boundary (as make_golden_rasters.py).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from make_golden_rasters import NEAR, synth  # noqa: E402

SEED = MG.SEED


def main():
    sys.path.insert(0, MG.REF)
    MG._install_standins()
    import LBDRNdataset as RD
    from LBDRNloss import LBDRNLoss
    from LBDRNmodel import LBDRNModel

    feats = np.load(os.path.join(HERE, "features.npz"))
    fA, lA = feats["A_K5_D2/features"], feats["A_K5_D2/labels"]
    out = {"x": fA, "t": lA}
    # forward
    for tag, bc, nl in (("bc64_nl2", 64, 2), ("bc32_nl3", 32, 3), ("bc256_nl1", 256, 1)):
        torch.manual_seed(SEED)
        m = LBDRNModel(dim_in=200, dim_hidden=bc, dim_out=8, num_layers=nl, activation=torch.nn.ReLU())
        with torch.no_grad():
            g = torch.Generator().manual_seed(5)
            for p in m.parameters():
                p.mul_(1.0 + 40.0 * torch.rand(p.shape, generator=g))   # (Sine's w0 = 30 is gone: weights of that size again)
            out[f"fwd/{tag}/params"] = MG._flat(m.state_dict())
            out[f"fwd/{tag}/y"] = m(torch.from_numpy(fA)).numpy()
    # train
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=200, dim_hidden=64, dim_out=8, num_layers=2, activation=torch.nn.ReLU())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    loss_fn = LBDRNLoss()
    g = torch.Generator().manual_seed(21)
    out["train/params0"] = MG._flat(m.state_dict())
    batches = np.stack([torch.randperm(fA.shape[0], generator=g)[:160].numpy() for _ in range(3)])
    out["train/batches"] = batches.astype(np.int64)
    for s in range(3):
        x, t = torch.from_numpy(fA[batches[s]]), torch.from_numpy(lA[batches[s]])
        opt.zero_grad()
        m.train()
        loss = loss_fn(m(x), t)
        loss.backward()
        if s == 0:
            out["train/step0/grads"] = np.concatenate([p.grad.numpy().reshape(-1) for p in m.parameters()])
        opt.step()
        out[f"train/step{s}/loss"] = np.float32(loss.item())
    out["train/params_final"] = MG._flat(m.state_dict())
    # raster
    K, D, C, H, W, bc, nl = 5, 2, 8, 128, 128, 64, 2
    img = synth(2105, C, H, W, 2.0)
    MG._RASTERS["mem.tif"] = img
    f, l = RD.process("mem.tif", K, D, "mem_base.tif")
    f, l = np.ascontiguousarray(f, np.float32), np.ascontiguousarray(l, np.float32)
    N, F = f.shape
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=F, dim_hidden=bc, dim_out=C, num_layers=nl, activation=torch.nn.ReLU())
    epochs, bs = 30, 2048
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)       # (at the SIREN initialisation a ReLU network moves slowly: 1e-2, 240 steps)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(epochs / 3)), gamma=0.1)
    gg = torch.Generator().manual_seed(2205)
    ft, lt = torch.from_numpy(f), torch.from_numpy(l)
    steps = 0
    for e in range(epochs):
        perm = torch.randperm(N, generator=gg)
        for s in range(0, N, bs):
            idx = perm[s:s + bs]
            opt.zero_grad()
            m.train()
            loss = loss_fn(m(ft[idx]), lt[idx])
            loss.backward()
            opt.step()
            steps += 1
        sched.step()
    flat = MG._flat(m.state_dict())
    flat = (flat.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    sd, k = {}, 0
    for name, val in m.state_dict().items():
        sd[name] = torch.from_numpy(flat[k:k + val.numel()].reshape(val.shape).copy())
        k += val.numel()
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        y_pred = m(ft)
        residual = torch.round(y_pred * (2 ** K - 1)).numpy()
    base = (img >> K).astype(np.uint16)
    res_chw = np.transpose(residual.reshape(H, W, C), (2, 0, 1))
    image = np.round((base << K).astype(np.float32) + res_chw).astype(np.uint16)
    t = y_pred.numpy().astype(np.float64) * (2 ** K - 1)
    dist = np.abs(t - (np.floor(t) + 0.5))
    near = np.flatnonzero(dist.reshape(-1) < NEAR)
    out.update({"raster/img": img, "raster/params": flat, "raster/residual": (image - (base << K)).astype(np.uint8),
                "raster/near_idx": near.astype(np.int64), "raster/final_loss": np.float32(loss.item()),
                "raster/mse": np.float32(np.mean((img.astype(np.float32) - image.astype(np.float32)) ** 2)),
                "raster/cfg": np.array([K, D, bc, nl, steps], np.int64)})
    np.savez_compressed(os.path.join(HERE, "relu_net.npz"), **out)
    print("wrote relu_net.npz", os.path.getsize(os.path.join(HERE, "relu_net.npz")), "bytes; raster loss",
          float(loss.item()), "near", near.size, "of", dist.size, "losses", [float(out[f"train/step{s}/loss"]) for s in range(3)])


if __name__ == "__main__":
    main()
