"""Image-level sharding for multi-GPU runs: every image (or split_ratio tile) is an
independent fit with its own re-seeded RNG (ref encode.py:200-205), so ranks never exchange data on
the path; the only collective is one all_gather of small per-image records at the end."""
import torch
import torch.distributed as dist


def assign(n_items, rank, world):
    """Static round-robin: item i belongs to rank i % world (64 images over 8 GPUs -> 8 each)."""
    return list(range(rank, n_items, world))


def gather_records(local, width, device=None):
    """local: list of `width`-float records of this rank.  Returns all ranks' records, ordered by
    rank then local order.  Works on any initialised backend (nccl = RCCL on ROCm, gloo on CPU);
    without a process group it returns the local records."""
    t = torch.tensor(local, dtype=torch.float64, device=device).reshape(-1, width)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t.cpu().tolist()
    world = dist.get_world_size()
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([t.shape[0]], dtype=torch.int64, device=device))
    cap = int(max(c.item() for c in counts))
    padded = torch.zeros((cap, width), dtype=torch.float64, device=device)
    padded[:t.shape[0]] = t
    bufs = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    out = []
    for b, c in zip(bufs, counts):
        out += b[:int(c.item())].cpu().tolist()
    return out


def max_over_ranks(value, device=None):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
