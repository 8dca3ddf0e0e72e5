"""CPU restatement of the reference's fit / apply loop with stock torch CPU ops ("port").

TEST INFRASTRUCTURE ONLY: used by tests/ as a second opinion on training numerics and by
bench.py's cpu_baseline leg (kind "port").  It mirrors the reference's cost structure: a
map-style dataset with per-sample __getitem__ (ref LBDRNdataset.py:136-155), DataLoader(shuffle)
(ref encode.py:69-70), per-step zero_grad/forward/loss/backward/Adam (ref
modified_ignite_engine.py:18-27), StepLR once per epoch and a whole-image evaluation pass with
the concatenating metric after every epoch (ref encode.py:96-117, LBDRNperformance.py:13-21),
then the chunked apply + rounding of decode.py:122-134.  GDAL / fpzip / file I/O are excluded.
pytorch-ignite is not installed here; the event order (scheduler.step() before the evaluation
pass, best-model selection on strict improvement) is restated from the reference's text.
"""
import math
import time

import numpy as np
import torch
from torch import nn
from torch.utils.data import DataLoader, Dataset

import oracle as O


class PortModel(nn.Module):
    """Same structure, state_dict keys and init draws as ref LBDRNmodel.py:46-82."""

    class _Layer(nn.Module):
        def __init__(self, din, dout, first, sigmoid=False):
            super().__init__()
            self.linear = nn.Linear(din, dout)
            b = 1 / din if first else math.sqrt(6 / din) / 30.0
            nn.init.uniform_(self.linear.weight, -b, b)
            nn.init.uniform_(self.linear.bias, -b, b)
            self.sigmoid = sigmoid

        def forward(self, x):
            z = self.linear(x)
            return torch.sigmoid(z) if self.sigmoid else torch.sin(30.0 * z)

    def __init__(self, F, bc, C, nl):
        super().__init__()
        self.net = nn.Sequential(*[self._Layer(F if i == 0 else bc, bc, i == 0) for i in range(nl)])
        self.last_layer = self._Layer(bc, C, False, sigmoid=True)

    def forward(self, x):
        return self.last_layer(self.net(x))


class _Rows(Dataset):
    def __init__(self, f, l):
        self.f, self.l = f, l

    def __len__(self):
        return len(self.f)

    def __getitem__(self, i):
        return self.f[i], self.l[i]


def fit(img, K, D, bc, nl, lr, bs, epochs, cfg=None, num_workers=0, faithful=True, val_duration=1):
    """-> dict(params best, best_epoch, epoch_mse, seconds).  faithful=True: map-style dataset +
    DataLoader (form A of BASELINE.md); False: index_select minibatches + streaming MSE (form B).
    Both consume the global generator in the reference's order."""
    cfg = cfg or O.FeatCfg()
    msb, labels, mx = O.split_bits(img, K)
    feats = O.features(msb, D, cfg, mx)
    f, l = torch.from_numpy(feats), torch.from_numpy(labels)
    C = labels.shape[1]
    model = PortModel(feats.shape[1], bc, C, nl)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, int(epochs / 3)), gamma=0.1)
    loader = DataLoader(_Rows(f, l), batch_size=bs, shuffle=True, num_workers=num_workers)
    best, best_epoch, best_state, log = 1e6, -1, None, []
    t0 = time.time()

    def batches():
        if faithful:
            yield from loader
        else:
            torch.empty((), dtype=torch.int64).random_()
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
            perm = torch.randperm(len(f), generator=g)
            for s in range(0, len(f), bs):
                idx = perm[s:s + bs]
                yield f.index_select(0, idx), l.index_select(0, idx)

    losses = []
    for e in range(1, epochs + 1):
        model.train()
        for x, t in batches():
            opt.zero_grad()
            loss = nn.functional.mse_loss(model(x), t)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        sched.step()
        if epochs == 1:
            best_state, best_epoch = {k: v.clone() for k, v in model.state_dict().items()}, e
            continue
        if e % min(val_duration, epochs) == 0:
            model.eval()
            with torch.no_grad():
                if faithful:
                    yp = yt = None
                    for x, t in batches():
                        y = model(x)
                        yp = y if yp is None else torch.cat((y, yp), dim=0)
                        yt = t if yt is None else torch.cat((t, yt), dim=0)
                    mse = nn.functional.mse_loss(yp, yt).item()
                else:
                    sse, cnt = 0.0, 0
                    for x, t in batches():
                        sse += float(((model(x) - t) ** 2).sum())
                        cnt += t.numel()
                    mse = sse / cnt
            improved = mse < best
            log.append((e, mse, improved))
            if improved:
                best, best_epoch = mse, e
                best_state = {k: v.clone() for k, v in model.state_dict().items()}
    flat = np.concatenate([v.numpy().reshape(-1) for v in best_state.values()])
    return dict(params=flat, best_epoch=best_epoch, epoch_mse=log, losses=losses, msb=msb, msb_max=mx,
                seconds=time.time() - t0, F=feats.shape[1], C=C)


def apply(msb, params, K, D, bc, nl, cfg=None):
    """decode.py:73-134 with torch CPU ops -> [C,H,W] uint16."""
    cfg = cfg or O.FeatCfg()
    C, H, W = msb.shape
    feats = O.features(msb, D, cfg)
    model = PortModel(feats.shape[1], bc, C, nl)
    sd, k = {}, 0
    for name, v in model.state_dict().items():
        sd[name] = torch.from_numpy(params[k:k + v.numel()].reshape(tuple(v.shape)).copy())
        k += v.numel()
    model.load_state_dict(sd)
    model.eval()
    x = torch.from_numpy(feats)
    with torch.no_grad():
        y = torch.zeros(x.shape[0], C)
        step = 2 ** 22
        for b in range(math.ceil(x.shape[0] / step)):
            y[step * b:step * (b + 1)] = model(x[step * b:step * (b + 1)])
        residual = torch.round(y * (2 ** K - 1)).numpy()
    residual = np.transpose(residual.reshape(H, W, C), (2, 0, 1))
    return np.round((msb.astype(np.uint16) << K).astype(np.float32) + residual).astype(np.uint16)
