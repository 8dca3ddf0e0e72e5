""".bin container of the codec: header (a12) + per tile the network payload and the MSB payload.

Header layout (ref encode.py:37-64, decode.py:25-53), big-endian:
  hdr_len:u8  split_ratio:u8  width:u16  height:u16  (K<<4)|D:u8  (log2(bc)<<4)|nl:u8
  nn_bytes:u24 x sr^2   base_bytes:u32 x sr^2
The two payload codecs of the reference are third-party programs that are absent from this image
(fpzip 1.2.4 for the weights, GDAL/OpenJPEG lossless JP2 for the MSB plane).
  * weights: an fpzip stream, always written and read by this package's restatement of the published algorithm
    (csrc/weights_codec.hip; byte compatibility with fpzip is unverified here: parity unpinned) and cross-checked
    against the fpzip module wherever that is importable;
  * MSB plane: by default this package's own tagged format (LBB2, coded on the GPU in 3.5 ms; LBB1, the older host codec).
    LBDRN_BASE_CODEC=jp2 writes what the reference writes -- a multi-component reversible (5/3) lossless JPEG 2000 file,
    through the OpenJPEG library GDAL's JP2OpenJPEG driver wraps (lbdrn_hip/jp2.py, csrc/jp2_shim.c) -- and ANY payload
    that starts with the JP2 signature box (or a raw codestream marker) is decoded that way, so a base payload written
    by the reference decodes too.  Pixel values are pinned (the path is lossless); byte identity with a GDAL-written
    file is not (GDAL is absent: its tiling and box choices cannot be replayed here).  bpsp of an LBB2 file is not
    comparable with the reference's published figures; that of a jp2 file is, up to those choices.
DESIGN.md "Container" states what is and is not interchangeable.
"""
import lzma
import struct
import zlib

import numpy as np

NN_PRIVATE_MAGIC = b"LBW1"
BASE_PRIVATE_MAGIC = b"LBB1"
BASE_LBB2_MAGIC = b"LBB2"


# ---------------------------------------------------------------- header

HEADER_FLAG_RELU = 0x01      # bit 0 of the extension byte: the hidden activation is torch.nn.ReLU (constants.HIDDEN_ACTIVATION)


def pack_header(split_ratio, width, height, K, bc, nl, D, nn_bytes_list, base_bytes_list, activation="sine"):
    """The reference's header (ref encode.py:37-64), byte for byte -- for the default network.  activation="relu" (the
    reference's alternative, a source edit there: encode.py:75 / decode.py:108) appends ONE extension byte behind the
    reference's fields and counts it in the length byte: the reference's reader strips `n_bytes_header` bytes whatever they are
    (decode.py:26, 182), so such a file still parses there, and this package's decoder no longer depends on both sides
    having made the same edit (a file encoded with relu and decoded under the default constants gave
    garbage low bits without an error)."""
    tiles = split_ratio * split_ratio
    if len(nn_bytes_list) != tiles or len(base_bytes_list) != tiles:
        raise ValueError("one nn and one base size per tile expected")
    log2bc = int(bc).bit_length() - 1
    if 1 << log2bc != bc:
        raise ValueError(f"base_channel must be a power of two for the header nibble, got {bc}")
    for name, v, hi in (("K", K, 15), ("D", D, 15), ("nl", nl, 15), ("log2(bc)", log2bc, 15),
                        ("width", width, 65535), ("height", height, 65535), ("split_ratio", split_ratio, 255)):
        if not 0 <= v <= hi:
            raise OverflowError(f"{name}={v} does not fit its header field (max {hi})")
    if activation not in ("sine", "relu"):
        raise ValueError(f"hidden activation {activation!r}")
    ext = bytes([HEADER_FLAG_RELU]) if activation == "relu" else b""
    n = 8 + 7 * tiles + len(ext)
    if n > 255:
        raise OverflowError(f"header of {n} bytes does not fit its one-byte length field")
    out = struct.pack(">BBHHBB", n, split_ratio, width, height, (K << 4) | D, (log2bc << 4) | nl)
    for b in nn_bytes_list:
        if not 0 <= b < 1 << 24:
            raise OverflowError(f"nn payload of {b} bytes does not fit 3 bytes")
        out += int(b).to_bytes(3, "big")
    for b in base_bytes_list:
        if not 0 <= b < 1 << 32:
            raise OverflowError(f"base payload of {b} bytes does not fit 4 bytes")
        out += struct.pack(">I", b)
    out += ext
    assert len(out) == n
    return out


def header_activation(buf):
    """The hidden activation a header names: "relu" / "sine" from the extension byte, None where the header has none (every
    reference-written file and every default-network file: the decoder then takes constants.HIDDEN_ACTIVATION, as before)."""
    n, sr = buf[0], buf[1]
    base = 8 + 7 * sr * sr
    if n <= base:
        return None
    if n > base + 1 or buf[base] & ~HEADER_FLAG_RELU:
        raise ValueError(f"header extension of {n - base} byte(s) / flags {buf[base]:#x}: written by a newer version of this package")
    return "relu" if buf[base] & HEADER_FLAG_RELU else "sine"


def unpack_header(buf):
    """-> (n_bytes_header, split_ratio, width, height, K, bc, nl, D, nn_bytes_list, base_bytes_list)"""
    n, sr, width, height, kd, bcnl = struct.unpack_from(">BBHHBB", buf, 0)
    tiles = sr * sr
    pos = 8
    nn = [int.from_bytes(buf[pos + 3 * i: pos + 3 * i + 3], "big") for i in range(tiles)]
    pos += 3 * tiles
    base = [struct.unpack_from(">I", buf, pos + 4 * i)[0] for i in range(tiles)]
    return n, sr, width, height, kd >> 4, 1 << (bcnl >> 4), bcnl & 15, kd & 15, nn, base


# ---------------------------------------------------------------- weights payload (a10)

def flatten_state(state_dict):
    """state_dict tensors in key order, C order, one float32 vector (ref encode.py:123-128)."""
    return np.concatenate([v.detach().cpu().numpy().astype(np.float32).reshape(-1)
                           for v in state_dict.values()])


def unflatten_state(flat, like_state_dict):
    """Inverse walk (ref decode.py:114-120)."""
    import torch
    out, k = {}, 0
    for name, ref in like_state_dict.items():
        n = ref.numel()
        out[name] = torch.from_numpy(np.ascontiguousarray(flat[k:k + n]).reshape(tuple(ref.shape)).copy())
        k += n
    if k != flat.size:
        raise ValueError(f"parameter vector has {flat.size} values, the model takes {k}")
    return out


WEIGHT_PRECISIONS = (0,) + tuple(range(9, 33))   # what the weight payload coder takes (0 = 32 = lossless)


def check_weight_precision(precision):
    """Refuse a `-prec` the weight payload coder will refuse -- BEFORE a fit is spent on it (encode.py / sweep.py call
    this when they parse their arguments; the coder itself refuses the same set: lbdrn_weights_encode).  fpzip codes precisions 1..8 with a
    narrow residual coder that this package does not restate (csrc/weights_codec.hip)."""
    if precision not in WEIGHT_PRECISIONS:
        raise ValueError(f"-prec {precision}: the weight payload coder takes 0 (lossless) or 9..32 bits")
    return precision


def truncate_precision(flat, precision):
    """The value map of the weight payload: keep the `precision` most significant bits of each float32 pattern
    (sign, exponent, leading mantissa bits), clear the rest; precision 0 or 32 = lossless.  This IS fpzip's lossy
    mode as its published algorithm defines it -- the order-preserving integer map drops the 32-precision low bits
    of the (complemented) pattern and its inverse restores them as zeros, for every input alike: negative numbers,
    both zeros, denormals, infinities, NaNs (a NaN whose payload sits only in the dropped bits becomes an infinity)
    -- restated in csrc/weights_codec.hip and oracle/fpz_port.py.  Not checked against an fpzip build (absent here):
    parity unpinned."""
    if precision in (0, 32):
        return flat.astype(np.float32).copy()
    if not 2 <= precision < 32:
        raise ValueError(f"precision {precision} outside [2,32]")   # (the payload coder itself takes 9..32)
    mask = np.uint32((0xFFFFFFFF << (32 - precision)) & 0xFFFFFFFF)
    return (flat.astype(np.float32).view(np.uint32) & mask).view(np.float32)


def _fpzip_module():
    try:
        import fpzip
        return fpzip
    except ImportError:
        return None


MAX_WEIGHT_VALUES = 1 << 27   # what decode_weights will allocate for when the caller names no expected count


def _native_encode(flat, precision):
    import ctypes
    from . import _lib
    L = _lib.lib()
    cap = L.lbdrn_weights_bound(flat.size)
    out = (ctypes.c_uint8 * cap)()
    nbytes = ctypes.c_size_t()
    _lib.check(L.lbdrn_weights_encode(flat.ctypes.data_as(ctypes.c_void_p), flat.size, int(precision), out, cap,
                                      ctypes.byref(nbytes)))
    return bytes(out[:nbytes.value])


def _native_decode(buf, expected):
    import ctypes
    from . import _lib
    L = _lib.lib()
    n, prec = ctypes.c_int64(), ctypes.c_int32()
    _lib.check(L.lbdrn_weights_info(buf, len(buf), ctypes.byref(n), ctypes.byref(prec)))
    # the count comes from the stream: checked against what the network needs (or a fixed ceiling) BEFORE anything
    # is allocated by it (a damaged header could ask for 16 GB)
    if expected is not None and n.value != expected:
        raise _lib.LbdrnError(f"weight payload holds {n.value} values, the network needs {expected}")
    if n.value > MAX_WEIGHT_VALUES:
        raise _lib.LbdrnError(f"weight payload claims {n.value} values")
    out = np.empty(n.value, np.float32)
    _lib.check(L.lbdrn_weights_decode(buf, len(buf), out.ctypes.data_as(ctypes.c_void_p), out.size))
    return out


def encode_weights(flat, precision):
    """The float32 parameter vector as a 1-D fpzip stream (ref encode.py:129: fpzip.compress(params, precision=16,
    order='C')).

    WHO writes the stream does not depend on what happens to be installed: it is always lbdrn_weights_encode, this
    package's restatement of the published algorithm (parity unpinned), so a .bin written here is
    decoded by the same code that wrote it on every box.  Where the fpzip module is importable the stream is
    cross-checked on the spot -- fpzip must decode it to exactly the truncated values -- and a disagreement raises
    instead of shipping a payload the reference could not read.  LBDRN_WEIGHTS_CODEC=fpzip hands the job to the
    module itself (the reference's own bytes, for interchange experiments).  Precisions <= 8 are refused: fpzip
    codes them with a narrow residual coder this package does not restate."""
    import os
    flat = np.ascontiguousarray(flat, dtype=np.float32)
    fpzip = _fpzip_module()
    if os.environ.get("LBDRN_WEIGHTS_CODEC") == "fpzip":
        if fpzip is None:
            raise RuntimeError("LBDRN_WEIGHTS_CODEC=fpzip, but the fpzip module is not importable")
        return fpzip.compress(flat, precision=precision, order="C")
    stream = _native_encode(flat, precision)
    if fpzip is not None and flat.size:
        back = np.asarray(fpzip.decompress(stream, order="C"), dtype=np.float32).reshape(-1)
        want = truncate_precision(flat, precision)
        if back.size != want.size or not np.array_equal(back.view(np.uint32), want.view(np.uint32)):
            raise RuntimeError("the installed fpzip does not decode this package's weight stream to the same values: "
                               "the restated syntax differs from fpzip's (set LBDRN_WEIGHTS_CODEC=fpzip to write with "
                               "the module)")
    return stream


def decode_weights(buf, expected=None):
    """Weight payload -> float32 vector (ref decode.py:113).  `expected`: the parameter count the header's network
    shape needs; a stream that holds another number is refused before anything is allocated.

    Deterministic like encode_weights: this package's decoder reads the stream; where the fpzip module is importable
    its answer is compared and a difference raises (a stream written by real fpzip in a syntax this restatement
    gets wrong would otherwise become silently wrong weights).  A stream this package's decoder refuses is handed to
    fpzip when present.  Streams of earlier builds (tag LBW1: byte planes + zlib) still decode."""
    buf = bytes(buf)
    if buf[:4] == NN_PRIVATE_MAGIC:
        precision, count = struct.unpack_from(">BI", buf, 4)
        if expected is not None and count != expected:
            raise ValueError(f"weight payload holds {count} values, the network needs {expected}")
        nbytes = 4 if precision in (0, 32) else (precision + 7) // 8
        raw = np.frombuffer(zlib.decompress(buf[9:]), np.uint8).reshape(nbytes, count)
        q = np.zeros(count, np.uint32)
        for i in range(nbytes):
            q |= raw[i].astype(np.uint32) << np.uint32(24 - 8 * i)
        return q.view(np.float32)
    fpzip = _fpzip_module()
    from . import _lib
    try:
        out = _native_decode(buf, expected)
    except _lib.LbdrnError:
        if fpzip is None:
            raise
        out = None
    if fpzip is not None:
        theirs = np.asarray(fpzip.decompress(buf, order="C"), dtype=np.float32).reshape(-1)
        if expected is not None and theirs.size != expected:
            raise ValueError(f"weight payload holds {theirs.size} values, the network needs {expected}")
        if out is not None and (theirs.size != out.size or not np.array_equal(theirs.view(np.uint32), out.view(np.uint32))):
            raise RuntimeError("fpzip and this package's decoder read different weights from the same stream")
        out = theirs if out is None else out
    return out


# ---------------------------------------------------------------- MSB-plane payload

def encode_base(msb, codec="LBB2", device="cuda:0", as_uint8=None):
    """Lossless MSB plane [C,H,W] (uint8 when max <= 255 else uint16, ref LBDRNdataset.py:100); stands where
    the reference runs gdal_translate to JPEG 2000 (ref encode.py:137).

    LBB2 (default): coded on the GPU by lbdrn_plane_encode (csrc/plane_codec.hip) -- msb may be a numpy array
    or the device tensor the fit left in HBM (then as_uint8 says which dtype the decoder hands back).  LBB1: the portable host codec of earlier bitstreams (plane
    predictor + LZMA; a minute per 8 x 2048^2 tile), kept so that those files still decode and for hosts
    that only need to write small rasters."""
    if codec.lower() in ("jp2", "jpeg2000", "jp2openjpeg"):   # ref encode.py:137
        from . import jp2
        if not isinstance(msb, np.ndarray):
            from . import ops
            x = ops.from_device_u16(msb)
            msb = x.astype(np.uint8) if as_uint8 else x
        return jp2.encode(msb)
    if codec == "LBB2":
        from . import ops
        if isinstance(msb, np.ndarray):
            code = 1 if msb.dtype == np.uint8 else 2
            planes = ops.to_device_u16(np.ascontiguousarray(msb).astype(np.uint16), device)
        else:
            planes, code = msb, (1 if as_uint8 else 2)
        C, H, W = planes.shape
        return BASE_LBB2_MAGIC + struct.pack(">BHII", code, C, H, W) + ops.plane_encode(planes)
    if codec != "LBB1":
        raise ValueError(f"unknown MSB payload codec {codec!r}")
    msb = np.ascontiguousarray(msb)
    C, H, W = msb.shape
    code = 1 if msb.dtype == np.uint8 else 2
    x = msb.astype(np.uint16)
    d = x.copy()
    d[:, 1:, :] -= x[:, :-1, :]
    e = d.copy()
    e[:, :, 1:] -= d[:, :, :-1]
    s = e.view(np.int16).astype(np.int32)
    z = ((s << 1) ^ (s >> 31)).astype(np.uint16)  # zig-zag
    body = lzma.compress((z >> 8).astype(np.uint8).tobytes() + (z & 0xFF).astype(np.uint8).tobytes(),
                         preset=6)
    return BASE_PRIVATE_MAGIC + struct.pack(">BHII", code, C, H, W) + body


def decode_base(buf, device="cuda:0", keep_on_device=False):
    """MSB payload -> [C,H,W] numpy (uint8 / uint16 as encoded).  keep_on_device: an LBB2 payload is returned
    as the device tensor it was decoded into (uint16 bits, int16 storage) -- decode.py feeds it straight to
    the apply kernel."""
    from . import jp2
    if jp2.is_jp2(buf):   # a JPEG 2000 payload (LBDRN_BASE_CODEC=jp2, or the reference's own: decode.py:69-73)
        x = jp2.decode(buf)
        if keep_on_device:
            from . import ops
            return ops.to_device_u16(x.astype(np.uint16), device)
        return x
    if buf[:4] == BASE_LBB2_MAGIC:
        from . import ops
        code, C, H, W = struct.unpack_from(">BHII", buf, 4)
        planes = ops.plane_decode(bytes(buf[15:]), C, H, W, device)
        if keep_on_device:
            return planes
        x = ops.from_device_u16(planes)
        return x.astype(np.uint8) if code == 1 else x
    if buf[:4] != BASE_PRIVATE_MAGIC:
        raise ValueError("MSB payload is neither a JPEG 2000 stream nor one of this package's tagged formats")
    code, C, H, W = struct.unpack_from(">BHII", buf, 4)
    raw = np.frombuffer(lzma.decompress(bytes(buf[15:])), np.uint8)
    n = C * H * W
    z = (raw[:n].astype(np.uint16) << 8) | raw[n:2 * n].astype(np.uint16)
    s = ((z >> 1).astype(np.int32) ^ -(z & 1).astype(np.int32)).astype(np.int16)
    e = s.view(np.uint16).reshape(C, H, W)
    d = np.cumsum(e, axis=2, dtype=np.uint16)
    x = np.cumsum(d, axis=1, dtype=np.uint16)
    return x.astype(np.uint8) if code == 1 else x
