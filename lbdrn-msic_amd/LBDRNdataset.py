"""Drop-in for the reference's LBDRNdataset module: process(), LBDRNDataset, tile split / merge and
the GeoTIFF writer, with the bit split and the feature / label matrices computed by liblbdrn_hip
(ref LBDRNdataset.py:92-155).  The fused fit (encode.py) never materialises these matrices; this
module exists for callers that want them, and for parity tests."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from lbdrn_hip import ops, raster_io
from lbdrn_hip.features import FeatCfg

DEVICE = "cuda:0"


def write_tiff_with_gdal(output_path, array):
    raster_io.write_raster(output_path, array)


def tile_windows(width, height, split_ratio):
    """(i, j, x0, y0, w, h) per tile; the last row / column absorbs the remainder
    (ref LBDRNdataset.py:46-68)."""
    tw, th = width // split_ratio, height // split_ratio
    for i in range(split_ratio):
        for j in range(split_ratio):
            w = tw if j + 1 < split_ratio else width - tw * j
            h = th if i + 1 < split_ratio else height - th * i
            yield i, j, j * tw, i * th, w, h


def split_image(input_file, output_dir, split_ratio):
    img = raster_io.read_raster(input_file)
    img = img.reshape((-1,) + img.shape[-2:])
    for i, j, x0, y0, w, h in tile_windows(img.shape[2], img.shape[1], split_ratio):
        write_tiff_with_gdal(os.path.join(output_dir, f"tile_{i}_{j}.tif"),
                             np.ascontiguousarray(img[:, y0:y0 + h, x0:x0 + w]))
        print(f"Tile {i}_{j} created, shape: {w}x{h}")


def merge_tiles(input_dir, output_file, split_ratio, width, height):
    out = None
    for i, j, x0, y0, w, h in tile_windows(width, height, split_ratio):
        tile = raster_io.read_raster(os.path.join(input_dir, f"tile_{i}_{j}_recon.tif"))
        tile = tile.reshape((-1,) + tile.shape[-2:])
        if out is None:
            out = np.zeros((tile.shape[0], height, width), tile.dtype)
        out[:, y0:y0 + h, x0:x0 + w] = tile
        print(f"Tile {i}_{j} merged, shape: {w}x{h}")
    write_tiff_with_gdal(output_file, out)


def process(path, K, D, output_path, device=None):
    """-> (features [H*W,F] float32, labels [H*W,C] float32) as numpy arrays; writes the MSB plane
    to output_path like the reference does (ref LBDRNdataset.py:92-133)."""
    dev = torch.device(device or DEVICE)
    img = np.ascontiguousarray(raster_io.read_raster(path)).astype(np.uint16)
    img = img.reshape((-1,) + img.shape[-2:])
    C, H, W = img.shape
    img_d = ops.to_device_u16(img, dev)
    msb_d, msb_max = ops.split_bits(img_d, K)
    msb = ops.from_device_u16(msb_d)
    write_tiff_with_gdal(output_path, msb.astype(np.uint16) if msb_max > 255 else msb.astype(np.uint8))
    geom = ops.FeatureGeometry(C, H, W, K, D, msb_max, FeatCfg.from_constants(), dev)
    features = ops.features(geom, msb_d)
    labels = ops.labels(img_d, K)
    return features.cpu().numpy(), labels.cpu().numpy()


class LBDRNDataset(Dataset):
    def __init__(self, args):
        filename = os.path.splitext(os.path.basename(args.path))[0]
        features, labels = process(args.path, args.K, args.D, f"{args.output_dir}/{filename}_base.tif")
        self.features = torch.from_numpy(features)
        self.labels = torch.from_numpy(labels)
        self.n_pixels = len(self.features)
        self.n_feature = self.features.shape[-1]
        self.channels = self.labels.shape[-1]
        self.n_subpixels = self.n_pixels * self.channels

    def __len__(self):
        return self.n_pixels

    def __getitem__(self, idx):
        return self.features[idx], self.labels[idx]
