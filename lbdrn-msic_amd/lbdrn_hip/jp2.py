"""ctypes binding of liblbdrn_jp2.so (include/lbdrn_jp2.h): the MSB-plane payload as a real JPEG 2000 stream through the
OpenJPEG library that GDAL's JP2OpenJPEG driver wraps (ref encode.py:137, decode.py:69-73).  Host code only; built by
csrc/build.py where openjpeg.h and libopenjp2 are found, absent elsewhere (available() says which)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("LBDRN_JP2_LIB") or os.path.join(os.path.dirname(_HERE), "liblbdrn_jp2.so")
SIGNATURE = b"\x00\x00\x00\x0cjP  \r\n\x87\n"     # the JP2 signature box every .jp2 file starts with
CODESTREAM = b"\xff\x4f\xff\x51"                    # SOC + SIZ markers of a raw codestream

_lib = None


class Jp2Error(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise Jp2Error(f"{_PATH} not built: the JPEG 2000 payload needs OpenJPEG (openjpeg.h + libopenjp2) at build time "
                           "(python lbdrn-msic_amd/csrc/build.py)")
        L = ctypes.CDLL(_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.lbdrn_jp2_last_error.restype = ctypes.c_char_p
        L.lbdrn_jp2_encode.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.POINTER(u8p), ctypes.POINTER(ctypes.c_size_t)]
        L.lbdrn_jp2_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, i32p, i32p, i32p, i32p]
        L.lbdrn_jp2_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.lbdrn_jp2_free.argtypes = [u8p]
        L.lbdrn_jp2_free.restype = None
        L.lbdrn_jp2_set_threads.argtypes = [ctypes.c_int32]
        # OpenJPEG codes the code blocks of a call on this many worker threads (same bytes): LBDRN_JP2_THREADS, default
        # up to 8 of the host's cores -- a 8 x 2048 x 2048 plane set takes seconds of ONE core otherwise
        L.lbdrn_jp2_set_threads(default_threads())
        _lib = L
    return _lib


def default_threads():
    v = os.environ.get("LBDRN_JP2_THREADS")
    if v is not None:
        return max(0, int(v))
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return min(8, n)


def set_threads(n):
    """Worker threads OpenJPEG may use inside one encode / decode call (lbdrn_jp2_set_threads); returns the previous value."""
    return int(lib().lbdrn_jp2_set_threads(int(n)))


def available():
    try:
        lib()
        return True
    except (Jp2Error, OSError):
        return False


def is_jp2(buf):
    return bytes(buf[:12]) == SIGNATURE or bytes(buf[:4]) == CODESTREAM


def _check(rc, what):
    if rc != 0:
        raise Jp2Error(f"{what}: {lib().lbdrn_jp2_last_error().decode(errors='replace')}")


def encode(planes):
    """[C,H,W] uint8 / uint16 -> a complete .jp2 file in memory: multi-component, reversible 5/3, one lossless layer
    (what `gdal_translate -of JP2OpenJPEG -co QUALITY=100 -co REVERSIBLE=YES` asks OpenJPEG for)."""
    planes = np.ascontiguousarray(planes)
    if planes.ndim == 2:
        planes = planes[None]
    if planes.dtype not in (np.uint8, np.uint16):
        raise ValueError("planes must be uint8 or uint16")
    bits = 8 if planes.dtype == np.uint8 else 16
    x = np.ascontiguousarray(planes.astype(np.uint16))
    C, H, W = x.shape
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    _check(lib().lbdrn_jp2_encode(x.ctypes.data_as(ctypes.c_void_p), C, H, W, bits, ctypes.byref(out), ctypes.byref(n)),
           "lbdrn_jp2_encode")
    try:
        return ctypes.string_at(out, n.value)
    finally:
        lib().lbdrn_jp2_free(out)


def decode(buf):
    """.jp2 file or raw codestream -> [C,H,W] uint8 (precision <= 8 bits) or uint16."""
    buf = bytes(buf)
    C, H, W, bits = (ctypes.c_int32() for _ in range(4))
    _check(lib().lbdrn_jp2_info(buf, len(buf), ctypes.byref(C), ctypes.byref(H), ctypes.byref(W), ctypes.byref(bits)),
           "lbdrn_jp2_info")
    if C.value < 1 or H.value < 1 or W.value < 1 or C.value * H.value * W.value > (1 << 33):
        raise Jp2Error(f"implausible JPEG 2000 geometry {C.value} x {H.value} x {W.value}")
    out = np.empty((C.value, H.value, W.value), np.uint16)
    _check(lib().lbdrn_jp2_decode(buf, len(buf), out.ctypes.data_as(ctypes.c_void_p), C.value, H.value, W.value),
           "lbdrn_jp2_decode")
    return out.astype(np.uint8) if bits.value <= 8 else out
