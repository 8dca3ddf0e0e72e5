"""Image-level sharding for multi-GPU runs: every image (or split_ratio tile) is an
independent fit with its own re-seeded RNG (ref encode.py:200-205), so ranks never exchange data on
the path; the only collective is one all_gather of small per-image records at the end (bench.py), or
one gather of the finished payloads to rank 0 (encode.py / decode.py / sweep.py under torchrun)."""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, world, local_rank) as torchrun exports them; (0, 1, 0) for a plain `python` run."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_host_group():
    """Process group for the CLIs' end-of-run exchange of host objects (payload bytes, log lines, tiles).
    gloo: what is exchanged lives in host memory and no kernel waits for it; the fits themselves never
    communicate.  Returns (rank, world, local_rank); does nothing for a single process."""
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local


def device_for(local_rank):
    """One process per GPU; ranks beyond the device count share devices round-robin (two processes per GPU
    is a useful setting for sweeps: the second fit fills the idle gaps of the first one's kernel chain)."""
    n = torch.cuda.device_count()
    return f"cuda:{local_rank % n}" if n else "cuda:0"


def bind_device(local_rank):
    """device_for() and make that GPU the process's current device, once, before any GPU call: every allocation,
    stream and kernel of this rank then lives on its own GPU (ops._call additionally takes device and stream
    from the tensors it is handed, so a stray tensor of another GPU cannot pull a launch over to it)."""
    name = device_for(local_rank)
    if torch.cuda.device_count():
        torch.cuda.set_device(torch.device(name))
    return name


def gather_to_root(obj):
    """Every rank's `obj` as a list on rank 0 (None elsewhere); [obj] without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def all_to_all_objects(obj):
    """Every rank's `obj` as a list on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def finish():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


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
