"""Generate tests/golden/*.npz by RUNNING the reference's own modules (authoring container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; the GPU box never runs this)

What is imported from /root/reference and executed as-is:
  constants, LBDRNmodel.LBDRNModel, LBDRNloss.LBDRNLoss, LBDRNdataset.process,
  encode.write_image_header, decode.read_image_header.
LBDRNdataset/encode/decode import packages that are not installed here (osgeo, fpzip, ignite,
tensorboard).  The generator pre-seeds sys.modules with inert stand-ins so that the imports
succeed; the only stand-in that carries data is an in-memory `gdal.Open(path).ReadAsArray()` /
`GetDriverByName().Create()` pair, which hands process() the numpy array this script made and
swallows the GeoTIFF it writes.  All arithmetic in the fixtures is the reference's numpy/torch.
Lines that cannot be called (decode.test() body, the ignite trainer) are replayed here from the
reference's text with the reference's own model/loss objects; each such block cites its lines.

The fixtures hold data only (inputs + expected outputs); no reference source is stored.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 19920517  # encode.py:169


# ---------------------------------------------------------------- stand-ins for absent packages

_RASTERS = {}


def _install_standins():
    gdal = types.ModuleType("osgeo.gdal")
    gdal.GDT_Byte, gdal.GDT_UInt16, gdal.GDT_Float32, gdal.GDT_Float64 = 1, 2, 6, 7
    gdal.UseExceptions = lambda: None

    class _DS:
        def __init__(self, arr):
            self._a = arr
            self.RasterXSize = arr.shape[-1]
            self.RasterYSize = arr.shape[-2]

        def ReadAsArray(self):
            return self._a.copy()

    class _Band:
        def WriteArray(self, a):
            pass

    class _Out:
        def GetRasterBand(self, i):
            return _Band()

        def FlushCache(self):
            pass

    class _Drv:
        def Create(self, *a):
            return _Out()

    gdal.Open = lambda path: _DS(_RASTERS[path])
    gdal.GetDriverByName = lambda name: _Drv()
    osgeo = types.ModuleType("osgeo")
    osgeo.gdal = gdal
    sys.modules["osgeo"] = osgeo
    sys.modules["osgeo.gdal"] = gdal
    sys.modules["fpzip"] = types.ModuleType("fpzip")
    # ignite: names only, never executed
    ig = types.ModuleType("ignite")
    ige = types.ModuleType("ignite.engine")
    ige.Events = type("Events", (), {})
    igee = types.ModuleType("ignite.engine.engine")
    igee.Engine = type("Engine", (), {})
    igu = types.ModuleType("ignite.utils")
    igu.convert_tensor = lambda x, **k: x
    igm = types.ModuleType("ignite.metrics")
    igmm = types.ModuleType("ignite.metrics.metric")
    igmm.Metric = type("Metric", (), {})
    for name, mod in [("ignite", ig), ("ignite.engine", ige), ("ignite.engine.engine", igee),
                      ("ignite.utils", igu), ("ignite.metrics", igm),
                      ("ignite.metrics.metric", igmm)]:
        sys.modules[name] = mod
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {})
    sys.modules["torch.utils.tensorboard"] = tb


def _img(seed, C, H, W, lo=0, hi=10000):
    rng = np.random.default_rng(seed)
    yy = np.arange(H)[:, None] / H
    xx = np.arange(W)[None, :] / W
    out = np.empty((C, H, W), np.uint16)
    for c in range(C):
        a = np.sin(2 * np.pi * (rng.uniform(0.5, 3) * yy + rng.uniform(0.5, 3) * xx)
                   + rng.uniform(0, 6))
        a = lo + (a + 1) / 2 * (hi - lo) * 0.9 + rng.normal(0, 0.01 * (hi - lo), (H, W))
        out[c] = np.clip(np.rint(a), lo, hi).astype(np.uint16)
    return out


def _flat(sd):
    """encode.py:123-128"""
    return np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in sd.values()])


def main():
    sys.path.insert(0, REF)
    _install_standins()
    import constants  # noqa: F401
    import LBDRNdataset as RD
    from LBDRNloss import LBDRNLoss
    from LBDRNmodel import LBDRNModel
    import encode as RE
    import decode as RDEC

    DEFAULT = dict(USE_COORDINATES=False, EMBEDDING=False, USE_COLORS=True, RELATIVE=True)

    def run_process(img, K, D, **flags):
        for k, v in {**DEFAULT, **flags}.items():
            setattr(RD, k, v)
        _RASTERS["mem.tif"] = img
        f, l = RD.process("mem.tif", K, D, "mem_base.tif")
        for k, v in DEFAULT.items():
            setattr(RD, k, v)
        return f, l

    # ------------------------------------------------------------ features / labels (a1-a3)
    cases = {
        "A_K5_D2": (_img(1, 8, 24, 20), 5, 2, {}),
        "B_K3_D1": (_img(2, 4, 33, 17), 3, 1, {}),
        "C_K5_D0": (_img(3, 3, 9, 11), 5, 0, {}),
        "D_u8_K5_D2": (_img(4, 2, 12, 13, 0, 6000), 5, 2, {}),  # msb max <= 255 -> uint8 branch
        "E_K1_D3": (_img(5, 1, 10, 9, 0, 65535), 1, 3, {}),     # HW-only band count 1, deep msb
        "F_coords": (_img(6, 8, 12, 10), 5, 2, dict(USE_COORDINATES=True)),
        "G_embed": (_img(7, 8, 12, 10), 5, 2, dict(USE_COORDINATES=True, EMBEDDING=True)),
        "H_embed_nocolor": (_img(8, 4, 8, 14), 4, 2,
                            dict(USE_COORDINATES=True, EMBEDDING=True, USE_COLORS=False)),
        "I_absolute": (_img(9, 4, 10, 12), 5, 2, dict(RELATIVE=False)),
        "J_tiny_D2": (_img(10, 2, 3, 4), 6, 2, {}),              # D close to the image size
    }
    feats = {}
    for name, (img, K, D, flags) in cases.items():
        f, l = run_process(img, K, D, **flags)
        feats[name + "/img"] = img
        feats[name + "/K"] = np.int64(K)
        feats[name + "/D"] = np.int64(D)
        feats[name + "/flags"] = np.array([int({**DEFAULT, **flags}[k]) for k in
                                           ("USE_COORDINATES", "EMBEDDING", "USE_COLORS",
                                            "RELATIVE")], np.int64)
        feats[name + "/features"] = f.astype(np.float32)
        feats[name + "/labels"] = l.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "features.npz"), **feats)

    # ------------------------------------------------------------ init (a6) + RNG position
    model_cases = {"bc64_nl2": (200, 64, 8, 2), "bc16_nl3": (18, 16, 3, 3), "bc32_nl1": (50, 32, 4, 1)}
    init = {}
    for name, (F, bc, C, nl) in model_cases.items():
        torch.manual_seed(SEED)
        m = LBDRNModel(dim_in=F, dim_hidden=bc, dim_out=C, num_layers=nl)  # encode.py:71-77
        init[name + "/dims"] = np.array([F, bc, C, nl], np.int64)
        init[name + "/params"] = _flat(m.state_dict())
        init[name + "/keys"] = np.array(list(m.state_dict().keys()))
        # next two draws of the global generator = what the first DataLoader iterator consumes
        init[name + "/next_draws"] = np.array(
            [torch.empty((), dtype=torch.int64).random_().item() for _ in range(2)], np.int64)
    np.savez_compressed(os.path.join(OUT, "init.npz"), **init)

    # ------------------------------------------------------------ forward (a5)
    fw = {}
    fA = feats["A_K5_D2/features"]
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=200, dim_hidden=64, dim_out=8, num_layers=2)
    with torch.no_grad():
        fw["init/x"] = fA
        fw["init/params"] = _flat(m.state_dict())
        fw["init/y"] = m(torch.from_numpy(fA)).numpy()
        # "trained-like" weights: larger pre-activations so that sin(30 z) wraps many times
        g = torch.Generator().manual_seed(7)
        for p in m.parameters():
            p.mul_(1.0 + 4.0 * torch.rand(p.shape, generator=g))
        m.net[0].linear.weight.mul_(6.0)
        fw["wide/x"] = fA
        fw["wide/params"] = _flat(m.state_dict())
        fw["wide/y"] = m(torch.from_numpy(fA)).numpy()
        z0 = torch.from_numpy(fA) @ m.net[0].linear.weight.T + m.net[0].linear.bias
        fw["wide/z0_absmax"] = np.float32(z0.abs().max().item())
    fG = feats["G_embed/features"]
    torch.manual_seed(SEED)
    m5 = LBDRNModel(dim_in=fG.shape[1], dim_hidden=64, dim_out=8, num_layers=2)
    with torch.no_grad():
        fw["embed/x"] = fG
        fw["embed/params"] = _flat(m5.state_dict())
        fw["embed/y"] = m5(torch.from_numpy(fG)).numpy()
    np.savez_compressed(os.path.join(OUT, "forward.npz"), **fw)

    # ------------------------------------------------------------ teacher-forced updates (a7, a8)
    # modified_ignite_engine.py:18-27 replayed with the reference's model and loss objects and
    # torch.optim.Adam / StepLR exactly as encode.py:84-86 builds them.
    tr = {}
    lA = feats["A_K5_D2/labels"]
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=200, dim_hidden=64, dim_out=8, num_layers=2)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(3 / 3)), gamma=0.1)
    loss_fn = LBDRNLoss()
    g = torch.Generator().manual_seed(11)
    tr["params0"] = _flat(m.state_dict())
    nsteps = 6
    batches = np.stack([torch.randperm(fA.shape[0], generator=g)[:96].numpy() for _ in range(nsteps)])
    tr["batches"] = batches.astype(np.int64)
    tr["x"] = fA
    tr["t"] = lA
    for s in range(nsteps):
        x = torch.from_numpy(fA[batches[s]])
        t = torch.from_numpy(lA[batches[s]])
        opt.zero_grad()
        m.train()
        y = m(x)
        loss = loss_fn(y, t)
        loss.backward()
        tr[f"step{s}/grads"] = np.concatenate([p.grad.numpy().reshape(-1) for p in m.parameters()])
        tr[f"step{s}/lr"] = np.float64(opt.param_groups[0]["lr"])
        opt.step()
        tr[f"step{s}/loss"] = np.float32(loss.item())
        tr[f"step{s}/params"] = _flat(m.state_dict())
        if s % 2 == 1:
            sched.step()  # an "epoch" of two iterations: encode.py:98
    tr["exp_avg"] = np.concatenate([opt.state[p]["exp_avg"].numpy().reshape(-1) for p in m.parameters()])
    tr["exp_avg_sq"] = np.concatenate([opt.state[p]["exp_avg_sq"].numpy().reshape(-1) for p in m.parameters()])
    # whole-image MSE as the evaluator computes it: LBDRNperformance.py:18-21
    with torch.no_grad():
        m.eval()
        tr["final_mse"] = np.float32(torch.nn.functional.mse_loss(
            m(torch.from_numpy(fA)), torch.from_numpy(lA)).item())
    np.savez_compressed(os.path.join(OUT, "train.npz"), **tr)

    # ------------------------------------------------------------ apply + reconstruct (a11)
    # decode.py:122-134 replayed with the reference model; weights = the trained ones above with
    # the low 16 bits cleared (this repo's model of fpzip precision=16; not checked against fpzip).
    de = {}
    img = cases["A_K5_D2"][0]
    K = 5
    base = (img >> K).astype(np.uint16)
    flat = tr[f"step{nsteps - 1}/params"].copy()
    flat = (flat.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    sd, k = {}, 0
    for name, val in m.state_dict().items():  # decode.py:114-120
        sd[name] = torch.from_numpy(flat[k:k + val.numel()].reshape(val.shape).copy())
        k += val.numel()
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        y_pred = m(torch.from_numpy(fA))
        residual = torch.round(y_pred * (2 ** K - 1)).numpy()
    C, H, W = img.shape
    res_chw = np.transpose(residual.reshape(H, W, C), (2, 0, 1))
    image = np.round((base << K).astype(np.float32) + res_chw).astype(np.uint16)
    de["img"] = img
    de["K"] = np.int64(K)
    de["D"] = np.int64(2)
    de["params"] = flat
    de["y"] = y_pred.numpy()
    de["image"] = image
    org = img
    mse = np.mean((org.astype(np.float32) - image.astype(np.float32)) ** 2)  # decode.py:216
    de["mse"] = np.float32(mse)
    de["psnr"] = np.float64(10 * np.log10(10000 ** 2 / mse))  # :218-219
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **de)

    # ------------------------------------------------------------ header (a12)
    hd = {}
    hcases = [
        (1, 2048, 2048, 5, 64, 2, 2, [35096], [7654321]),
        (2, 6000, 5999, 6, 256, 3, 1, [1, 2, 3, 16777215], [4, 5, 6, 4294967295]),
        (1, 17, 33, 1, 16, 1, 0, [0], [0]),
        (5, 65535, 1, 15, 32768, 15, 15, list(range(25)), list(range(100, 125))),
    ]
    for i, (sr, w, h, K, bc, nl, D, nn, bb) in enumerate(hcases):
        path = os.path.join(OUT, "_hdr.tmp")
        RE.write_image_header(path, sr, w, h, K, bc, nl, D, nn, bb)
        with open(path, "rb") as f:
            raw = f.read()
        os.remove(path)
        parsed = RDEC.read_image_header(raw)
        hd[f"h{i}/args"] = np.array([sr, w, h, K, bc, nl, D], np.int64)
        hd[f"h{i}/nn"] = np.array(nn, np.int64)
        hd[f"h{i}/base"] = np.array(bb, np.int64)
        hd[f"h{i}/bytes"] = np.frombuffer(raw, np.uint8)
        hd[f"h{i}/parsed"] = np.array(list(parsed[:8]) + list(parsed[8]) + list(parsed[9]), np.int64)
    np.savez_compressed(os.path.join(OUT, "header.npz"), **hd)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
