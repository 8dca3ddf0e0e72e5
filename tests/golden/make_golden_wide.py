"""Generate tests/golden/wide_net.npz by RUNNING the reference's LBDRNModel / LBDRNLoss (authoring container
only; needs /root/reference): BASELINE.json configs[2] width (bc = 256, nl = 2) and a three-hidden-layer
network, on the reference-made feature matrix of fixture A (features.npz).  Forward outputs, and three
teacher-forced Adam steps of the bc = 256 model built exactly as encode.py:84-86 builds them
(modified_ignite_engine.py:18-27 replayed with the reference's own objects).  Data only."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 19920517  # encode.py:169


def _flat(sd):
    return np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in sd.values()])


def main():
    sys.path.insert(0, REF)
    from LBDRNloss import LBDRNLoss
    from LBDRNmodel import LBDRNModel
    feats = np.load(os.path.join(OUT, "features.npz"))
    fA, lA = feats["A_K5_D2/features"], feats["A_K5_D2/labels"]
    out = {"x": fA, "t": lA}
    for tag, bc, nl in (("bc256_nl2", 256, 2), ("bc64_nl3", 64, 3), ("bc128_nl1", 128, 1)):
        torch.manual_seed(SEED)
        m = LBDRNModel(dim_in=200, dim_hidden=bc, dim_out=8, num_layers=nl)
        with torch.no_grad():
            g = torch.Generator().manual_seed(3)
            for p in m.parameters():    # "trained-like": larger pre-activations, sin(30 z) wraps
                p.mul_(1.0 + 2.0 * torch.rand(p.shape, generator=g))
            out[f"{tag}/params"] = _flat(m.state_dict())
            out[f"{tag}/y"] = m(torch.from_numpy(fA)).numpy()
    torch.manual_seed(SEED)
    m = LBDRNModel(dim_in=200, dim_hidden=256, dim_out=8, num_layers=2)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    loss_fn = LBDRNLoss()
    g = torch.Generator().manual_seed(12)
    out["train256/params0"] = _flat(m.state_dict())
    batches = np.stack([torch.randperm(fA.shape[0], generator=g)[:128].numpy() for _ in range(3)])
    out["train256/batches"] = batches.astype(np.int64)
    for s in range(3):
        x, t = torch.from_numpy(fA[batches[s]]), torch.from_numpy(lA[batches[s]])
        opt.zero_grad()
        m.train()
        loss = loss_fn(m(x), t)
        loss.backward()
        if s == 0:
            out["train256/step0/grads"] = np.concatenate([p.grad.numpy().reshape(-1) for p in m.parameters()])
        opt.step()
        out[f"train256/step{s}/loss"] = np.float32(loss.item())
    out["train256/params_final"] = _flat(m.state_dict())
    np.savez_compressed(os.path.join(OUT, "wide_net.npz"), **out)
    print("wrote wide_net.npz", {k: v.shape for k, v in out.items() if k.endswith("/y")})


if __name__ == "__main__":
    main()
