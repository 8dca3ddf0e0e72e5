"""world_size-2 gloo rehearsal of the multi-GPU plumbing (sharding + the final metrics gather)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
    from lbdrn_hip import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.assign(7, rank, world)          # ragged: 4 + 3 images
    recs = [[float(i), 100.0 + i, float(rank)] for i in mine]
    allr = shard.gather_records(recs, 3)
    tmax = shard.max_over_ranks(1.0 + rank)
    dist.barrier()
    dist.destroy_process_group()
    torch.save({"mine": mine, "all": allr, "tmax": tmax}, os.path.join(out_dir, f"r{rank}.pt"))


def test_round_robin_sharding_and_record_gather(tmp_path):
    world, port = 2, 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["mine"] == [0, 2, 4, 6] and r1["mine"] == [1, 3, 5]
    assert sorted(r0["mine"] + r1["mine"]) == list(range(7))
    assert r0["all"] == r1["all"]
    assert [r[0] for r in r0["all"]] == [0, 2, 4, 6, 1, 3, 5]
    assert all(r[1] == 100.0 + r[0] for r in r0["all"])
    assert r0["tmax"] == r1["tmax"] == 2.0


def test_single_process_fallbacks():
    sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
    from lbdrn_hip import shard
    assert shard.assign(5, 0, 1) == [0, 1, 2, 3, 4]
    assert shard.gather_records([[1.0, 2.0]], 2) == [[1.0, 2.0]]
    assert shard.max_over_ranks(3.5) == 3.5


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks before touching any GPU and
    relays rank 0's single JSON line (LBDRN_BENCH_DRYRUN: launcher + gloo exchange only, nothing is measured)."""
    import json
    import subprocess
    env = dict(os.environ, LBDRN_BENCH_DRYRUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         check=True, env=env, capture_output=True, text=True, timeout=300).stdout
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["records"]) == 2 * 3
    assert sorted(r[0] for r in d["records"]) == [2.0, 3.0, 4.0, 5.0, 6.0, 7.0]   # the timed tiles of both ranks
    assert d["max_over_ranks"] == 2.0
    # a launcher that started a different number of ranks than --gpus is an error, not a silent n_gpus=1
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "--gpus 2" in bad.stderr


def test_bench_gpus_8_dry_run_has_the_shape_of_configs_3():
    """BASELINE.json configs[3] -- 64 tiles over 8 ranks -- through the launcher and the exchange (gloo, nothing
    measured): 64 records, every rank's own clock in the line, the communicator's world size as it reports it; and a
    rank that fails is named with what it said."""
    import json
    import subprocess
    env = dict(os.environ, LBDRN_BENCH_DRYRUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "8", "--warmup", "1"],
                         check=True, env=env, capture_output=True, text=True, timeout=600).stdout
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and len(d["records"]) == 64
    assert sorted(r[0] for r in d["records"]) == [float(i) for i in range(8, 72)]      # the 64 timed tiles, each once
    assert d["rank_elapsed_ms"] == [1000.0 * (1 + r) for r in range(8)] and d["max_over_ranks"] == 8.0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"],
                         env=dict(env, LBDRN_BENCH_DRYRUN_FAIL="2"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "rank 2" in bad.stderr and "simulated failure" in bad.stderr
