"""Generate tests/golden/tiles.npz by RUNNING the reference's LBDRNdataset.split_image / merge_tiles
(authoring container only; needs /root/reference) with a recording stand-in for the absent `osgeo.gdal`:
the windows split_image asks gdal.Translate for (srcWin) and the offsets merge_tiles writes its tiles at,
for several image sizes and split ratios.  Data only: one [sr*sr, 6] integer table per case."""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
CALLS = {"translate": [], "write": []}


class _Band:
    DataType = 2


class _DS:
    def __init__(self, w, h, data=None):
        self.RasterXSize, self.RasterYSize, self.RasterCount, self.data = w, h, 1, data

    def GetRasterBand(self, i):
        return _Band()

    def ReadAsArray(self):
        return self.data

    def WriteArray(self, data, x, y):
        CALLS["write"].append((x, y, data.shape[-1], data.shape[-2]))

    def FlushCache(self):
        pass


def main():
    gdal = types.ModuleType("osgeo.gdal")
    state = {}
    gdal.UseExceptions = lambda: None
    gdal.Open = lambda path: state["open"](path)
    gdal.Translate = lambda out, ds, srcWin=None: CALLS["translate"].append((os.path.basename(out),) + tuple(srcWin))
    gdal.GetDriverByName = lambda name: types.SimpleNamespace(Create=lambda path, w, h, n, dt: _DS(w, h))
    osgeo = types.ModuleType("osgeo")
    osgeo.gdal = gdal
    sys.modules["osgeo"], sys.modules["osgeo.gdal"] = osgeo, gdal
    sys.path.insert(0, REF)
    import LBDRNdataset as RD
    out = {}
    for (w, h, sr) in [(50, 31, 3), (64, 64, 2), (65, 130, 4), (7, 9, 5), (2048, 2048, 1), (1000, 999, 3)]:
        CALLS["translate"].clear()
        CALLS["write"].clear()
        state["open"] = lambda path: _DS(w, h)
        RD.split_image("in.tif", "tiles", sr)
        wins = {name: win for (name, *win) in CALLS["translate"]}
        # merge: every tile file reports the shape split_image cut it to
        def open_tile(path):
            name = os.path.basename(path).replace("_recon", "")
            x, y, tw, th = wins.get(name, (0, 0, w, h))
            return _DS(tw, th, np.zeros((1, th, tw), np.uint16))
        state["open"] = open_tile
        RD.merge_tiles("tiles", "out.tif", sr, w, h)
        table = []
        for i in range(sr):
            for j in range(sr):
                x, y, tw, th = wins[f"tile_{i}_{j}.tif"]
                table.append((i, j, x, y, tw, th))
        assert [(x, y, tw, th) for (_, _, x, y, tw, th) in table] == CALLS["write"]   # merge puts them back where split took them
        out[f"w{w}_h{h}_sr{sr}"] = np.array(table, np.int64)
    np.savez_compressed(os.path.join(OUT, "tiles.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
