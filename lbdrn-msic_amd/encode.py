"""LBDRN-MSIC encoder, MI355X build: same command line, output directory naming, .bin container
and log records as the reference's encode.py (ref encode.py:167-289); the per-image fit runs as
fused HIP kernels (lbdrn_hip.codec.fit_image) instead of DataLoader + ignite + autograd.

Bitstream: header | for each tile (row-major): network payload | MSB payload   (ref encode.py:29-36)

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N encode.py ... -sr S` fits the S*S tiles of the
image round-robin on N GPUs (one process per GPU, no exchange during the fits) and rank 0 assembles the
.bin; the result is byte-identical to the single-GPU run because every rank replays the random draws of
the tiles it does not fit (lbdrn_hip.codec.skip_fit_rng).
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

import logger
from lbdrn_hip import codec, container, raster_io, shard
from lbdrn_hip.features import FeatCfg
from LBDRNdataset import tile_windows

DEVICE = "cuda:0"
BASE_CODEC = os.environ.get("LBDRN_BASE_CODEC", "LBB2")   # "jp2": JPEG 2000 through OpenJPEG, what the reference writes
                                                          # (encode.py:137); "LBB1": the portable host payload of older files
REPORT_BOTH = os.environ.get("LBDRN_REPORT_BOTH_BPSP", "") not in ("", "0")   # also log the size of the payload format NOT
# written (JPEG 2000 costs 2-3 s of one host core per 8 x 2048^2 tile: off by default)
IN_FLIGHT = int(os.environ["LBDRN_IN_FLIGHT"]) if "LBDRN_IN_FLIGHT" in os.environ else None   # tiles of one image
# progressing at a time on a GPU (None: codec.fit_many's default, 4)


def write_image_header(header_path, split_ratio, width, height, K, bc, nl, D, nn_bytes_list,
                       base_bytes_list):
    data = container.pack_header(split_ratio, width, height, K, bc, nl, D, nn_bytes_list, base_bytes_list)
    with open(header_path, "wb") as f:
        f.write(data)
    if os.path.getsize(header_path) != data[0]:
        raise ValueError(f"Invalid number of bytes in header! expected {data[0]}, "
                         f"got {os.path.getsize(header_path)}")


def train(args, img=None):
    """Fit one image or tile and return (nn_payload, base_payload); logs what the reference logs
    (ref encode.py:67-157)."""
    if img is None:
        img = raster_io.read_raster(args.path)
    res = codec.fit_image(img, args.K, args.D, args.base_channel, args.num_layers, args.lr,
                          args.batch_size, args.epochs, args.val_duration,
                          cfg=FeatCfg.from_constants(), device=DEVICE, host_msb=False)
    return report_and_pack(args, res)


class BasePayloadsAhead:
    """The JPEG 2000 MSB payloads of an image's tiles, coded WHILE the GPU fits (round 6).  The MSB planes are `tile >> K`
    (ref LBDRNdataset.py:95): known the moment the raster is read, long before the fit ends -- and the reference's own format
    (LBDRN_BASE_CODEC=jp2: what `gdal_translate -of JP2OpenJPEG` writes, ref encode.py:137) costs 0.8-1.3 s of eight host
    threads per 8 x 2048^2 tile against a 0.1 s fit.  One background thread codes the tiles in order, OpenJPEG's own worker
    threads inside each call (lbdrn_jp2_set_threads; the ctypes call releases the interpreter); report_and_pack takes the
    finished payload -- the same bytes as coding it afterwards -- or waits for it."""

    def __init__(self, tiles, K):
        import threading
        self.payloads, self.errors, self.seconds = {}, {}, {}
        self._done = {k: threading.Event() for k in range(len(tiles))}

        def work():
            for k, tile in enumerate(tiles):
                t0 = time.time()
                try:
                    msb = np.asarray(tile).reshape((-1,) + tile.shape[-2:]) >> K
                    msb = msb.astype(np.uint8) if int(msb.max()) <= 255 else msb.astype(np.uint16)      # LBDRNdataset.py:100
                    self.payloads[k] = container.encode_base(msb, codec="jp2")
                except Exception as e:   # (reported where the payload is asked for)
                    self.errors[k] = e
                self.seconds[k] = time.time() - t0
                self._done[k].set()
        self._thread = threading.Thread(target=work, name="lbdrn-jp2-ahead", daemon=True)
        self._thread.start()

    def take(self, k):
        self._done[k].wait()
        if k in self.errors:
            raise self.errors[k]
        return self.payloads.pop(k)


def report_and_pack(args, res, base_ahead=None):
    """The records train() logs for one finished fit, and its two payloads.  base_ahead: () -> the MSB payload coded meanwhile."""
    filename = os.path.splitext(os.path.basename(args.path))[0]
    logger.log.info("total_params: {}".format(res.params.size))
    for epoch, mse, improved in res.epoch_mse:
        if improved:
            logger.log.info("Save current best val model (MSE: {:.5f}) @epoch {}".format(mse, epoch))
        else:
            logger.log.info("Model is not updated (MSE: {:.5f}) @epoch: {}".format(mse, epoch))
    logger.log.info("best epoch: {}".format(res.best_epoch))
    nn_payload = container.encode_weights(res.params, args.precision)      # ref encode.py:129
    logger.log.info(f"nn: {len(nn_payload)} bytes, bpsp={len(nn_payload) * 8 / res.n_subpixels}")
    # ref encode.py:137 (gdal_translate to JP2): here the plane is coded where the fit left it, in HBM;
    # uint8 when it fits, like the reference's MSB raster (LBDRNdataset.py:100)
    if base_ahead is not None:
        t0 = time.time()
        base_payload = base_ahead()
        logger.log.info(f"MSB payload ({BASE_CODEC}) coded beside the fit; waited {time.time() - t0:.3f}s more for it")
    elif BASE_CODEC == "LBB2":
        base_payload = container.encode_base(res.msb_device, device=DEVICE, as_uint8=res.msb_max <= 255)
    else:
        base_payload = container.encode_base(_host_msb(res), codec=BASE_CODEC)
    logger.log.info(f"MSB: {len(base_payload)} bytes: bpsp={len(base_payload) * 8 / res.n_subpixels}")
    if REPORT_BOTH:   # the other payload's size beside the one written: LBB2 is this package's private format, jp2 the
        # reference's (the bpsp the published tables are in); the record does not match results_summary.py's patterns
        other = "jp2" if BASE_CODEC == "LBB2" else "LBB2"
        try:
            alt = (container.encode_base(_host_msb(res), codec="jp2") if other == "jp2" else
                   container.encode_base(res.msb_device, device=DEVICE, as_uint8=res.msb_max <= 255))
            logger.log.info(f"MSB as {other}: {len(alt)} bytes: bpsp={len(alt) * 8 / res.n_subpixels} (written: {BASE_CODEC})")
        except Exception as e:   # (liblbdrn_jp2.so absent: say so, the run goes on)
            logger.log.info(f"MSB as {other}: unavailable ({e})")
    logger.log.info(f"{filename}: fit {res.seconds['fit']:.3f}s on {DEVICE}")
    return nn_payload, base_payload


def train_tiles(args, tiles, draws):
    """Several tiles of one image: fitted with IN_FLIGHT of them progressing at a time (their random draws
    were made beforehand, in tile order), then reported one after another like train() would.
    tiles: [(path, array)]; returns [(nn_payload, base_payload, log records or None)]."""
    results = codec.fit_images([arr for _, arr in tiles], args.K, args.D, args.base_channel, args.num_layers,
                               args.lr, args.batch_size, args.epochs, args.val_duration,
                               cfg=FeatCfg.from_constants(), device=DEVICE, host_msb=False, draws=draws,
                               in_flight=IN_FLIGHT)
    return results


def _host_msb(res):
    from lbdrn_hip import ops
    msb = ops.from_device_u16(res.msb_device)
    return msb.astype(np.uint8) if res.msb_max <= 255 else msb


def build_parser():
    p = argparse.ArgumentParser(description="LBDRN-MSIC")
    p.add_argument("--seed", type=int, default=19920517)
    p.add_argument("-rn", "--randomness", action="store_true", help="Allow randomness during training?")
    p.add_argument("-i", "--path", type=str, help="path of input tif or img file")
    p.add_argument("-o", "--output_dir", default="outputs", type=str, help="output dir")
    p.add_argument("-sr", "--split_ratio", type=int, default=1, help="tile size (default: 1)")
    p.add_argument("-K", "--K", type=int, default=5, help=" (default: 5)")
    p.add_argument("-bc", "--base_channel", type=int, default=64, help="base channel (default: 64)")
    p.add_argument("-nl", "--num_layers", type=int, default=2, help="Number of layers (default: 2)")
    p.add_argument("-D", "--D", type=int, default=2, help="#neighbors (2D+1)^2")
    p.add_argument("-prec", "--precision", type=int, default=16, help=" (default: 16)")
    p.add_argument("-lr", "--lr", type=float, default=1e-3, help="learning rate (default: 1e-3)")
    p.add_argument("-bs", "--batch_size", type=int, default=8192, help="batch size (default: 8192)")
    p.add_argument("-e", "--epochs", type=int, default=10, help="number of epochs to train (default: 10)")
    p.add_argument("-vd", "--val_duration", type=int, default=1,
                   help="number of epoch duration for val (default: 1)")
    return p


def main(argv=None, shard_tiles=None):
    """shard_tiles: spread the split_ratio tiles over the ranks of a torchrun launch (default: whenever
    WORLD_SIZE > 1; sweep.py passes False because it shards whole jobs instead)."""
    global DEVICE
    parser = build_parser()
    args = parser.parse_args(argv)
    if os.environ.get("LBDRN_WEIGHTS_CODEC") != "fpzip":
        try:   # a precision the payload coder refuses is refused here, not after the fit
            container.check_weight_precision(args.precision)
        except ValueError as e:
            parser.error(str(e))
    rank, world = 0, 1
    if shard_tiles is None:
        shard_tiles = shard.env_world()[1] > 1
    if shard_tiles:
        rank, world, local = shard.init_host_group()
        DEVICE = shard.bind_device(local)
    if not args.randomness:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
        random.seed(args.seed)
    org_path = args.path
    filename = os.path.splitext(os.path.basename(org_path))[0]
    args.output_dir = "{}/{}_r{}_K{}_bc{}_nl{}_D{}_prec{}_lr{}_bs{}_e{}".format(
        args.output_dir, filename, args.split_ratio, args.K, args.base_channel, args.num_layers,
        args.D, args.precision, args.lr, args.batch_size, args.epochs)
    os.makedirs(args.output_dir, exist_ok=True)
    bitstream_path = f"{args.output_dir}/{filename}.bin"
    log_path = f"{args.output_dir}/encode.txt"
    done = False
    if os.path.exists(log_path) and os.path.exists(bitstream_path):
        with open(log_path) as f:
            done = "Time elapsed" in f.read()
    if world > 1:   # one decision for all ranks, taken before anyone touches the log file
        done = all(shard.all_to_all_objects(done))
    if done:
        if rank == 0:
            print("Bitstream already created!")
        if world > 1:
            shard.finish()
        return 0
    if rank == 0:
        logger.create_logger(args.output_dir, "encode.txt")
    else:
        logger.create_logger(args.output_dir, "", log_file_only=True)
    start_time = time.time()
    img = raster_io.read_raster(org_path)
    img = img.reshape((-1,) + img.shape[-2:])
    height, width = img.shape[-2:]
    windows = list(tile_windows(width, height, args.split_ratio)) if args.split_ratio > 1 else [None]
    n_feature = FeatCfg.from_constants().feature_dim(img.shape[0], args.D)
    # every tile's random draws, in tile order, whoever fits it: the reference runs the tiles one after
    # another on one generator (ref encode.py:231-262)
    C = img.shape[0]
    draws = [codec.draw_fit(n_feature, args.base_channel, C, args.num_layers, args.epochs, args.val_duration)
             for _ in windows]
    mine, jobs = [t for t in range(len(windows)) if t % world == rank], []
    for t in mine:
        path, tile = org_path, img
        if windows[t] is not None:
            i, j, x0, y0, w, h = windows[t]
            path = f"{args.output_dir}/tile_{i}_{j}.tif"
            tile = np.ascontiguousarray(img[:, y0:y0 + h, x0:x0 + w])
        jobs.append((path, tile))
    # the reference's MSB format is host work that needs nothing from the fit: it starts now (LBDRN_JP2_AHEAD=0: afterwards, as before)
    ahead = None
    if jobs and BASE_CODEC.lower() in ("jp2", "jpeg2000", "jp2openjpeg") and os.environ.get("LBDRN_JP2_AHEAD", "1") != "0":
        ahead = BasePayloadsAhead([tile for _, tile in jobs], args.K)
    results = train_tiles(args, jobs, [draws[t] for t in mine]) if jobs else []
    fitted = []   # (tile index, nn payload, MSB payload, captured log records or None)
    for k, (t, (path, _), res) in enumerate(zip(mine, jobs, results)):
        args.path = path
        take = (lambda k=k: ahead.take(k)) if ahead is not None else None
        if world > 1:
            with logger.capture() as lines:
                logger.log.info(args)
                nn, base = report_and_pack(args, res, take)
        else:
            lines = None
            logger.log.info(args)
            nn, base = report_and_pack(args, res, take)
        fitted.append((t, nn, base, lines))
    gathered = shard.gather_to_root(fitted) if world > 1 else [fitted]
    if rank == 0:
        tiles = sorted((rec for part in gathered for rec in part), key=lambda rec: rec[0])
        assert [rec[0] for rec in tiles] == list(range(len(windows)))
        for _, _, _, lines in tiles:
            if lines is not None:
                logger.replay(lines)
        header = container.pack_header(args.split_ratio, width, height, args.K, args.base_channel,
                                       args.num_layers, args.D, [len(rec[1]) for rec in tiles],
                                       [len(rec[2]) for rec in tiles], activation=FeatCfg.from_constants().activation)
        with open(bitstream_path, "wb") as f:
            f.write(header)
            for _, nn, base, _ in tiles:
                f.write(nn)
                f.write(base)
        logger.log.info(f"Time elapsed: {time.time() - start_time}")
    if world > 1:
        shard.finish()
    return 0


if __name__ == "__main__":
    sys.exit(main())
