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


def draw_pass_seeds(plan):
    """One DataLoader iterator per pass of the plan, in order (every pass consumes two draws); returns the
    sampler seeds of the training passes."""
    seeds = []
    for kind, _ in plan:
        seed = draw_iterator_seed()
        if kind == "train":
            seeds.append(seed)
    return seeds


class DevicePermutationStream:
    """The permutations of a fit's training passes, computed on the GPU by lbdrn_randperm (the
    exact torch.randperm sequences).  The MT19937 recurrence is serial (5 ms for 4 M words, whatever the
    number of seeds generated side by side), so the work runs on a side stream in two batches -- epoch 1
    alone, then all the others -- and the training stream only waits for the batch it needs: the first
    wait hides behind the row-matrix build, the second behind epoch 1."""

    def __init__(self, n, epochs, val_duration, device, train_seeds=None):
        """train_seeds: the sampler seeds of the training passes when they were drawn earlier
        (codec.draw_fit); None = draw them now from the global generator."""
        from . import ops
        self.plan = epoch_plan(epochs, val_duration)
        order = [e for kind, e in self.plan if kind == "train"]
        seeds = list(train_seeds) if train_seeds is not None else draw_pass_seeds(self.plan)
        assert len(seeds) == len(order)
        self.seeds = dict(zip(order, seeds))
        self._row = {e: i for i, e in enumerate(order)}
        main = torch.cuda.current_stream(device)
        side = _side_stream(device)
        side.wait_stream(main)
        self._batches = []
        with torch.cuda.stream(side):
            for lo, hi in ((0, 1), (1, len(seeds))):
                if hi > lo:
                    perms = ops.randperm(seeds[lo:hi], n, device)
                    perms.record_stream(main)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    self._batches.append((lo, hi, perms, ev))

    def get(self, epoch):
        i = self._row[epoch]
        for lo, hi, perms, ev in self._batches:
            if lo <= i < hi:
                torch.cuda.current_stream(perms.device).wait_event(ev)
                return perms[i - lo]
        raise KeyError(epoch)

    def close(self):
        self._batches = []


_SIDE = {}


def _side_stream(device):
    key = torch.device(device).index or 0
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


GPU_RANDPERM_MAX = 0xFFFFFFFF // 20  # torch's randperm switches algorithm above this
