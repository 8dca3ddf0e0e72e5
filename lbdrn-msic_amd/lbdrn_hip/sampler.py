"""Minibatch order of the reference's DataLoader(shuffle=True), without the DataLoader (a4).

The reference never seeds the sampler itself: every `iter(DataLoader)` draws two int64 from the
GLOBAL torch CPU generator -- `_base_seed` (torch/utils/data/dataloader.py:706-710) and the
RandomSampler seed (torch/utils/data/sampler.py:163-165) -- and the epoch's order is
`torch.randperm(n, generator=Generator().manual_seed(seed))` (sampler.py:182).  Both the trainer
(encode.py:157) and the per-epoch evaluator run (encode.py:105) create one iterator per pass,
so the global generator is consumed train, eval, train, eval, ...  This module replays exactly
those draws with the same torch calls; only the permutation is then shipped to the GPU instead
of 4.19 M Python-level __getitem__ calls per pass (ref LBDRNdataset.py:151-155).
"""
from concurrent.futures import ThreadPoolExecutor

import torch


def draw_iterator_seed():
    """RNG side effects of creating one DataLoader iterator; returns the sampler seed."""
    torch.empty((), dtype=torch.int64).random_()  # _base_seed (only used by worker processes)
    return int(torch.empty((), dtype=torch.int64).random_().item())


def permutation(seed, n):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


def epoch_plan(epochs, val_duration):
    """Sequence of passes the reference runs: list of ('train'|'eval', epoch)  (encode.py:96-117)."""
    plan = []
    for e in range(1, epochs + 1):
        plan.append(("train", e))
        if epochs != 1 and e % min(val_duration, epochs) == 0:
            plan.append(("eval", e))
    return plan


class DevicePermutationStream:
    """Same draws as PermutationStream; the permutations of all epochs are computed on the GPU by one
    lbdrn_randperm call (the exact torch.randperm sequences, ~13 ms for ten 4 M-element orders instead
    of 60-300 ms of host Fisher-Yates each)."""

    def __init__(self, n, epochs, val_duration, device):
        from . import ops
        self.plan = epoch_plan(epochs, val_duration)
        seeds, self._row = [], {}
        for kind, e in self.plan:
            seed = draw_iterator_seed()
            if kind == "train":
                self._row[e] = len(seeds)
                seeds.append(seed)
        self.seeds = dict(zip(self._row, seeds))
        self._perms = ops.randperm(seeds, n, device)

    def get(self, epoch):
        return self._perms[self._row[epoch]]

    def close(self):
        self._perms = None


GPU_RANDPERM_MAX = 0xFFFFFFFF // 20  # torch's randperm switches algorithm above this


class PermutationStream:
    """Draws every pass's seed up front (the global generator is touched by nothing else during the
    fit) and computes the train permutations on worker threads so that the host-side Fisher-Yates
    (0.1-0.3 s for 4 M indices) overlaps the GPU work of earlier epochs."""

    def __init__(self, n, epochs, val_duration, workers=4, pin=True):
        self.n = n
        self.plan = epoch_plan(epochs, val_duration)
        self.seeds = {}
        for kind, e in self.plan:
            seed = draw_iterator_seed()
            if kind == "train":
                self.seeds[e] = seed
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self._pin = pin and torch.cuda.is_available()
        self._futs = {e: self._pool.submit(self._make, s) for e, s in sorted(self.seeds.items())}

    def _make(self, seed):
        p = permutation(seed, self.n)
        return p.pin_memory() if self._pin else p

    def get(self, epoch):
        return self._futs.pop(epoch).result()

    def close(self):
        self._pool.shutdown(wait=False, cancel_futures=True)
