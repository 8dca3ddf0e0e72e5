"""The K-sweep of the reference's run.sh (ref run.sh:29-42) as one multi-GPU job.

run.sh loops images x K = 1..6 serially on one device, starting two Python processes per point.  Every
(image, K) point is an independent encode + decode that re-seeds its RNG (ref encode.py:200-205), so here the
points are dealt round-robin to the ranks of a torchrun launch -- one process per GPU, no exchange between
points -- and each rank runs its points in-process (the HIP library and the CUDA context are loaded once).
Rank 0 gathers one small record per point at the end and, with --summary, writes the CSV of
results_summary.py.

    python -m torch.distributed.run --nproc-per-node 8 sweep.py -o outputs --images data/*.tif
    python sweep.py -o outputs --images a.tif b.tif --k 1 6            # single GPU

The positional form of run.sh is kept too:  sweep.py DEVICE_ID D BC NL LR BS EPOCH SR OUTPUT_DIR
(run.sh exports HIP_VISIBLE_DEVICES=DEVICE_ID unless NGPU is set; sweep.py itself takes its device from the launcher).
"""
import argparse
import os
import sys
import time
import traceback

import decode
import encode
import results_summary
from lbdrn_hip import shard

REFERENCE_IMAGES = (
    [f"data/GF-dataset/GF-2/{name}.tif" for name in results_summary.TRIPLESAT]
    + [f"data/GF-dataset/GF-6/GF6-{s}/GF6_{s}_Sample_{x}.tif" for s in ("WFI", "PMS") for x in "ABCD"]
)


def build_parser():
    p = argparse.ArgumentParser(description="LBDRN-MSIC K sweep (encode + decode per point)")
    p.add_argument("legacy", nargs="*", help="run.sh form: DEVICE_ID D BC NL LR BS EPOCH SR OUTPUT_DIR")
    p.add_argument("--images", nargs="*", default=None, help="input rasters (default: the reference's list)")
    p.add_argument("--k", nargs=2, type=int, default=(1, 6), metavar=("FIRST", "LAST"), help="inclusive K range")
    p.add_argument("-o", "--output_dir", default="outputs")
    p.add_argument("-sr", "--split_ratio", type=int, default=1)
    p.add_argument("-bc", "--base_channel", type=int, default=64)
    p.add_argument("-nl", "--num_layers", type=int, default=2)
    p.add_argument("-D", "--D", type=int, default=2)
    p.add_argument("-prec", "--precision", type=int, default=16)
    p.add_argument("-lr", "--lr", type=str, default="0.001", help="kept as text: it is part of the directory name")
    p.add_argument("-bs", "--batch_size", type=int, default=8192)
    p.add_argument("-e", "--epochs", type=int, default=10)
    p.add_argument("--summary", action="store_true", help="rank 0 writes the results CSV at the end")
    return p


def parse(argv=None):
    a = build_parser().parse_args(argv)
    if a.legacy:
        if len(a.legacy) != 9:
            raise SystemExit("positional form takes exactly: DEVICE_ID D BC NL LR BS EPOCH SR OUTPUT_DIR")
        _, D, BC, NL, LR, BS, EPOCH, SR, OUT = a.legacy
        a.D, a.base_channel, a.num_layers, a.lr = int(D), int(BC), int(NL), LR
        a.batch_size, a.epochs, a.split_ratio, a.output_dir = int(BS), int(EPOCH), int(SR), OUT
    a.images = a.images if a.images else list(REFERENCE_IMAGES)
    if os.environ.get("LBDRN_WEIGHTS_CODEC") != "fpzip":
        from lbdrn_hip import container
        try:   # before any fit of the sweep starts
            container.check_weight_precision(a.precision)
        except ValueError as e:
            raise SystemExit(str(e))
    return a


def points(a):
    """The sweep in run.sh's order: images outermost, K inside."""
    return [(img, K) for img in a.images for K in range(a.k[0], a.k[1] + 1)]


def run_point(a, image, K):
    stem = os.path.splitext(os.path.basename(image))[0]
    common = ["-K", str(K), "-D", str(a.D), "-bc", str(a.base_channel), "-nl", str(a.num_layers),
              "-lr", a.lr, "-bs", str(a.batch_size), "-e", str(a.epochs), "-sr", str(a.split_ratio),
              "-prec", str(a.precision)]
    encode.main(["-i", image, "-o", a.output_dir] + common, shard_tiles=False)
    # float(lr) -> the directory name encode.py derives from its parsed float (ref run.sh:38 spells it by hand)
    folder = (f"{a.output_dir}/{stem}_r{a.split_ratio}_K{K}_bc{a.base_channel}_nl{a.num_layers}_D{a.D}"
              f"_prec{a.precision}_lr{float(a.lr)}_bs{a.batch_size}_e{a.epochs}")
    decode.main(["-i", f"{folder}/{stem}.bin", "-org", image], shard_tiles=False)
    return folder


def main(argv=None):
    a = parse(argv)
    rank, world, local = shard.init_host_group()
    encode.DEVICE = decode.DEVICE = shard.bind_device(local)
    todo = points(a)
    records = []
    for idx in shard.assign(len(todo), rank, world):
        image, K = todo[idx]
        t0 = time.time()
        print(f"[rank {rank}] Processing file: {image}  K={K}", flush=True)
        try:
            folder, err = run_point(a, image, K), None
        except Exception as e:   # run.sh carries on after a failed point; so does the sweep
            traceback.print_exc()
            folder, err = None, f"{type(e).__name__}: {e}"
        records.append((idx, image, K, folder, err, time.time() - t0))
    gathered = shard.gather_to_root(records)
    failed = 0
    if rank == 0:
        for idx, image, K, folder, err, secs in sorted(rec for part in gathered for rec in part):
            status = "ok" if err is None else f"FAILED ({err})"
            print(f"{os.path.basename(image)} K={K}: {status} in {secs:.1f}s")
            failed += err is not None
        print("All files processed.")
        if a.summary:
            stems = [os.path.splitext(os.path.basename(i))[0] for i in a.images]
            results_summary.save_to_csv(
                ["-o", a.output_dir, "-sr", str(a.split_ratio), "-bc", str(a.base_channel), "-nl", str(a.num_layers),
                 "-D", str(a.D), "-prec", str(a.precision), "-lr", a.lr, "-bs", str(a.batch_size),
                 "-e", str(a.epochs), "--k", str(a.k[0]), str(a.k[1]), "--files"] + stems)
    if world > 1:
        failed = sum(shard.all_to_all_objects(failed))
        shard.finish()
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
