"""Raster files in and out (the reference goes through GDAL: LBDRNdataset.py:71-89, 93).

GDAL is used when it is importable.  Without it this module reads and writes the subset the codec
itself produces and consumes: baseline TIFF, uncompressed strips, unsigned 8/16-bit or float32
samples, any band count, chunky or planar, either byte order -- and .npy arrays.  Arrays are
[C,H,W] (or [H,W] for one band), as gdal's ReadAsArray() returns them.
"""
import struct

import numpy as np

_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 16: "Q"}


def _gdal():
    try:
        from osgeo import gdal
        gdal.UseExceptions()
        return gdal
    except Exception:
        return None


def read_raster(path):
    if path.endswith(".npy"):
        return np.load(path)
    g = _gdal()
    if g is not None:
        return g.Open(path).ReadAsArray()
    return _read_tiff(path)


def write_raster(path, array):
    """array [C,H,W]: uint8 / uint16 / float32 / float64 (ref write_tiff_with_gdal)."""
    if array.dtype.type not in (np.uint8, np.uint16, np.float32, np.float64):
        raise ValueError("Unsupported data type in this function")
    if path.endswith(".npy"):
        np.save(path, array)
        return
    g = _gdal()
    if g is not None:
        code = {np.uint8: g.GDT_Byte, np.uint16: g.GDT_UInt16, np.float32: g.GDT_Float32,
                np.float64: g.GDT_Float64}[array.dtype.type]
        ds = g.GetDriverByName("GTiff").Create(path, array.shape[2], array.shape[1], array.shape[0], code)
        for i in range(array.shape[0]):
            ds.GetRasterBand(i + 1).WriteArray(array[i])
        ds.FlushCache()
        return
    _write_tiff(path, array)


def raster_size(path):
    a = read_raster(path)
    return a.shape[-1], a.shape[-2]


def _read_tiff(path):
    with open(path, "rb") as f:
        buf = f.read()
    bo = {b"II": "<", b"MM": ">"}.get(buf[:2])
    if bo is None or struct.unpack_from(bo + "H", buf, 2)[0] != 42:
        raise ValueError(f"{path}: not a classic TIFF (BigTIFF and other containers need GDAL)")
    off = struct.unpack_from(bo + "I", buf, 4)[0]
    n = struct.unpack_from(bo + "H", buf, off)[0]
    tags = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack_from(bo + "HHI4s", buf, off + 2 + 12 * i)
        size = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 16: 8}.get(typ)
        if size is None:
            continue
        raw = val if size * cnt <= 4 else buf[struct.unpack(bo + "I", val)[0]:][:size * cnt]
        if typ == 3:
            tags[tag] = list(struct.unpack(bo + f"{cnt}H", raw[:2 * cnt]))
        elif typ == 4:
            tags[tag] = list(struct.unpack(bo + f"{cnt}I", raw[:4 * cnt]))
        elif typ == 1:
            tags[tag] = list(raw[:cnt])
    W, H = tags[256][0], tags[257][0]
    bits = tags.get(258, [1])
    spp = tags.get(277, [1])[0]
    if tags.get(259, [1])[0] != 1:
        raise ValueError(f"{path}: compressed TIFF needs GDAL")
    if 322 in tags:
        raise ValueError(f"{path}: tiled TIFF needs GDAL")
    fmt = tags.get(339, [1])[0]
    planar = tags.get(284, [1])[0]
    if len(set(bits)) != 1:
        raise ValueError("mixed bit depths")
    dt = {(8, 1): "u1", (16, 1): "u2", (32, 1): "u4", (32, 3): "f4", (64, 3): "f8"}[(bits[0], fmt)]
    dt = np.dtype(bo + dt)
    data = b"".join(buf[o:o + c] for o, c in zip(tags[273], tags[279]))
    a = np.frombuffer(data, dt)
    if planar == 2:
        a = a[:spp * H * W].reshape(spp, H, W)
    else:
        a = a[:H * W * spp].reshape(H, W, spp).transpose(2, 0, 1)
    a = np.ascontiguousarray(a.astype(dt.newbyteorder("=")))
    return a[0] if spp == 1 else a


def _write_tiff(path, array):
    C, H, W = array.shape
    a = np.ascontiguousarray(array.astype(array.dtype.newbyteorder("<")))
    fmt = 3 if array.dtype.kind == "f" else 1
    bits = array.dtype.itemsize * 8
    plane = H * W * array.dtype.itemsize
    entries = []  # (tag, type, values)
    entries.append((256, 4, [W]))
    entries.append((257, 4, [H]))
    entries.append((258, 3, [bits] * C))
    entries.append((259, 3, [1]))
    entries.append((262, 3, [1]))
    entries.append((273, 4, None))  # strip offsets, patched below
    entries.append((277, 3, [C]))
    entries.append((278, 4, [H]))
    entries.append((279, 4, [plane] * C))
    entries.append((284, 3, [2]))   # planar: one strip per band
    if C > 1:
        entries.append((338, 3, [0] * (C - 1)))
    entries.append((339, 3, [fmt] * C))
    ifd_off = 8
    ifd_size = 2 + 12 * len(entries) + 4
    extra_off = ifd_off + ifd_size
    extra = b""
    data_off_holder = []

    def place(vals, typ):
        nonlocal extra
        size = 2 if typ == 3 else 4
        raw = struct.pack("<" + ("H" if typ == 3 else "I") * len(vals), *vals)
        if size * len(vals) <= 4:
            return raw.ljust(4, b"\0")
        off = extra_off + len(extra)
        extra += raw + (b"\0" if len(raw) % 2 else b"")
        return struct.pack("<I", off)

    # first pass to learn the size of the out-of-line area
    strip_slot = C * 4 if C * 4 > 4 else 0
    body = b""
    for tag, typ, vals in entries:
        if tag == 273:
            body += b"\0" * 12
            continue
        body += struct.pack("<HHI", tag, typ, len(vals)) + place(vals, typ)
    strip_tab_off = extra_off + len(extra)
    data_off = strip_tab_off + strip_slot
    data_off += data_off % 2
    offs = [data_off + i * plane for i in range(C)]
    extra2 = extra
    if strip_slot:
        strip_val = struct.pack("<I", strip_tab_off)
        extra2 += struct.pack("<%dI" % C, *offs)
    else:
        strip_val = struct.pack("<I", offs[0])
    out = struct.pack("<2sHI", b"II", 42, ifd_off) + struct.pack("<H", len(entries))
    for tag, typ, vals in entries:
        if tag == 273:
            out += struct.pack("<HHI", 273, 4, C) + strip_val
        else:
            start = 12 * [e[0] for e in entries].index(tag)
            out += body[start:start + 12]
    out += struct.pack("<I", 0) + extra2
    out = out.ljust(data_off, b"\0")
    with open(path, "wb") as f:
        f.write(out)
        f.write(a.tobytes())
