"""LBDRN-MSIC decoder, MI355X build: same command line, log records and reconstruction as the
reference's decode.py (ref decode.py:151-224); feature rebuild, network forward, rounding and
integer reconstruction run as one fused HIP kernel (lbdrn_hip.codec.apply_image).

Under `python -m torch.distributed.run --nproc-per-node N decode.py ...` the split_ratio tiles of the
bitstream are decoded round-robin on N GPUs and merged on rank 0 (tiles are independent: no exchange
while decoding)."""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

import logger
from lbdrn_hip import codec, container, ops, raster_io, shard
from lbdrn_hip.features import FeatCfg
from LBDRNdataset import tile_windows, write_tiff_with_gdal

DEVICE = "cuda:0"
K = D = bc = nl = None  # set from the header; test() reads them like the reference's does
ACTIVATION = None       # "relu" / "sine" where the header names the hidden activation (container.header_activation), else None


def read_image_header(bitstream):
    return container.unpack_header(bitstream)


def test(bitstream, dirname, filename, nn_bytes, base_bytes, write=True):
    """Decode one image or tile from the front of `bitstream`; returns the remaining bytes
    (ref decode.py:56-141)."""
    nn_payload, bitstream = bitstream[:nn_bytes], bitstream[nn_bytes:]
    base_payload, bitstream = bitstream[:base_bytes], bitstream[base_bytes:]
    base = container.decode_base(base_payload, device=DEVICE, keep_on_device=True)   # ref decode.py:69-73
    cfg = FeatCfg.from_constants()
    if ACTIVATION is not None and ACTIVATION != cfg.activation:   # the file says which network it holds: it wins over constants.py
        cfg.activation = ACTIVATION
    # the parameter count the header's network shape needs: a payload that holds another number is refused before
    # anything is sized by it (ref decode.py:114-120 slices the vector by state_dict shapes)
    need = ops.param_count(ops.make_net(cfg.feature_dim(int(base.shape[0]), D), bc, int(base.shape[0]), nl))
    params = container.decode_weights(nn_payload, expected=need)
    image = codec.apply_image(base, params, K, D, bc, nl, cfg=cfg, device=DEVICE)
    recon_path = f"{dirname}/{filename}_recon.tif"
    test.last_image = image
    if write:
        write_tiff_with_gdal(recon_path, image)
        logger.log.info(f"Recon: {recon_path}")
    return bitstream


def main(argv=None, shard_tiles=None):
    global K, D, bc, nl, DEVICE, ACTIVATION
    p = argparse.ArgumentParser(description="LBDRN-RSIC")
    p.add_argument("--seed", type=int, default=19920517)
    p.add_argument("-i", "--bin_path", type=str, help="binstream path")
    p.add_argument("-org", "--org_path", type=str, default=None, help="org path")
    args = p.parse_args(argv)
    rank, world = 0, 1
    if shard_tiles is None:
        shard_tiles = shard.env_world()[1] > 1
    if shard_tiles:
        rank, world, local = shard.init_host_group()
        DEVICE = shard.bind_device(local)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    random.seed(args.seed)
    dirname, basename = os.path.split(args.bin_path)
    dirname = dirname or "."
    filename = os.path.splitext(basename)[0]
    done = False
    if os.path.exists(f"{dirname}/decode.txt"):
        with open(f"{dirname}/decode.txt") as f:
            done = "bpsp" in f.read()
    if world > 1:
        done = all(shard.all_to_all_objects(done))
    if done:
        if rank == 0:
            print("Bitstream already decoded!")
        if world > 1:
            shard.finish()
        return 0
    if rank == 0:
        logger.create_logger(dirname, "decode.txt")
    else:
        logger.create_logger(dirname, "", log_file_only=True)
    logger.log.info(f"Binstream: {args.bin_path}")
    start_time = time.time()
    with open(args.bin_path, "rb") as fin:
        bitstream = fin.read()
    n_hdr, split_ratio, width, height, K, bc, nl, D, nn_list, base_list = read_image_header(bitstream)
    ACTIVATION = container.header_activation(bitstream)
    if ACTIVATION is not None and ACTIVATION != FeatCfg.from_constants().activation:
        logger.log.info(f"hidden activation {ACTIVATION} (from the header; constants.HIDDEN_ACTIVATION says otherwise)")
    bitstream = bitstream[n_hdr:]
    recon_path = f"{dirname}/{basename[:-4]}_recon.tif"
    if split_ratio > 1:
        decoded, offset = [], 0
        for t, (i, j, x0, y0, w, h) in enumerate(tile_windows(width, height, split_ratio)):
            if t % world == rank:
                test(bitstream[offset:], dirname, f"tile_{i}_{j}", nn_list[t], base_list[t], write=False)
                decoded.append((t, test.last_image))
            offset += nn_list[t] + base_list[t]
        gathered = shard.gather_to_root(decoded) if world > 1 else [decoded]
        if rank == 0:
            tiles = dict(rec for part in gathered for rec in part)
            merged = None
            for t, (i, j, x0, y0, w, h) in enumerate(tile_windows(width, height, split_ratio)):
                if merged is None:
                    merged = np.zeros((tiles[t].shape[0], height, width), tiles[t].dtype)
                merged[:, y0:y0 + h, x0:x0 + w] = tiles[t]
            write_tiff_with_gdal(recon_path, merged)
    elif rank == 0:
        test(bitstream, dirname, filename, nn_list[0], base_list[0])
    if rank == 0:
        logger.log.info(f"Time elapsed: {time.time() - start_time}")
        if args.org_path is not None:
            org_img = raster_io.read_raster(args.org_path)
            rec_img = raster_io.read_raster(recon_path)
            nbytes = os.path.getsize(args.bin_path)
            mse_value = np.mean((org_img.astype(np.float32) - rec_img.astype(np.float32)) ** 2)
            logger.log.info(f"MSE: {mse_value}")
            psnr = 10 * np.log10(10000 ** 2 / mse_value)    # peak fixed at 10000 (ref decode.py:218)
            logger.log.info(f"PSNR: {psnr}")
            logger.log.info(f"Total size: {nbytes} bytes, bpsp={nbytes * 8 / np.prod(org_img.shape)}")
            os.remove(recon_path)                             # ref decode.py:223-224
    if world > 1:
        shard.finish()
    return 0


if __name__ == "__main__":
    sys.exit(main())
