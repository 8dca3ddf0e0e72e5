"""ctypes front end of the C oracle plus the numpy restatements of the small host-side pieces.

TEST INFRASTRUCTURE ONLY (see oracle/lbdrn_oracle.c header): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
Citations are relative to /root/reference.
"""
import contextlib
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build as _build  # noqa: E402

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build()
        L = ctypes.CDLL(path)
        L.orc_sin.restype = ctypes.c_float
        L.orc_sin.argtypes = [ctypes.c_float]
        L.orc_cos.restype = ctypes.c_float
        L.orc_cos.argtypes = [ctypes.c_float]
        L.orc_sigmoid.restype = ctypes.c_float
        L.orc_sigmoid.argtypes = [ctypes.c_float]
        L.orc_param_count.restype = ctypes.c_int64
        L.orc_eval_sse.restype = ctypes.c_double
        _lib = L
    return _lib


def _p(a, ctype=None):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ---------------------------------------------------------------- config


class FeatCfg:
    """constants.py:3-14 as an object instead of module globals."""

    def __init__(self, use_coordinates=False, embedding=False, sigma=1.4, n_freq=12,
                 use_colors=True, relative=True):
        self.use_coordinates = bool(use_coordinates)
        self.embedding = bool(embedding)
        self.sigma = float(sigma)
        self.n_freq = int(n_freq)
        self.use_colors = bool(use_colors)
        self.relative = bool(relative)

    @property
    def P(self):
        """positional features per axis: LBDRNdataset.py:105"""
        if not self.use_coordinates:
            return 0
        return 2 * self.n_freq + 1 if self.embedding else 1

    def feature_dim(self, C, D):
        return 2 * self.P + (C * (2 * D + 1) ** 2 if self.use_colors else 0)


def pos_tables(H, W, cfg):
    """LBDRNdataset.py:108-118: per-row and per-column positional features.

    The reference builds [H,W,2] coords -> [H,W,2,1+2*N_FREQ] -> [H,W,2*(1+2*N_FREQ)]; entry
    [h,w,0,:] depends on h only and [h,w,1,:] on w only, so two small tables carry it all."""
    P = cfg.P
    if P == 0:
        return np.zeros((H, 0), np.float32), np.zeros((W, 0), np.float32)

    def axis(n):
        pos = (2 * np.arange(n) / (n - 1) - 1).astype(np.float32)  # :110-112 (f64 then cast)
        if not cfg.embedding:
            return pos[:, None].astype(np.float32)
        freq = cfg.sigma ** np.arange(cfg.n_freq) * np.pi  # f64, :114-115
        arg = freq * pos[:, None]  # f32 * f64 -> f64
        tab = np.concatenate([pos[:, None], np.sin(arg), np.cos(arg)], axis=-1)  # :116
        return tab.astype(np.float32)  # cast on store into the f32 feature array, :118

    with np.errstate(all="ignore"):
        return axis(H), axis(W)


# ---------------------------------------------------------------- a1/a2/a3


@contextlib.contextmanager
def hidden_activation(name):
    """Run the oracle's network functions with this hidden activation: 'sine' (Sine(w0 = 30), ref LBDRNmodel.py:37) or
    'relu' (torch.nn.ReLU(), the alternative named at ref encode.py:75 / decode.py:108)."""
    if name not in ("sine", "relu"):
        raise ValueError(name)
    lib().orc_set_hidden_activation(1 if name == "relu" else 0)
    try:
        yield
    finally:
        lib().orc_set_hidden_activation(0)


def split_bits(img, K):
    """LBDRNdataset.py:95-101,131 -> (msb [C,H,W] u16, labels [H*W,C] f32, msb_max)."""
    img = _c(img, np.uint16)
    if img.ndim == 2:
        img = img[None]
    C, H, W = img.shape
    msb = np.empty_like(img)
    labels = np.empty((H * W, C), np.float32)
    mx = ctypes.c_int(0)
    rc = lib().orc_split_bits(_p(img), C, H, W, K, _p(msb), _p(labels), ctypes.byref(mx))
    assert rc == 0, rc
    return msb, labels, mx.value


def features(msb, D, cfg, msb_max=None, idx=None):
    """LBDRNdataset.py:104-130 -> [n, F] f32; idx (int64 pixel indices) or raster order."""
    msb = _c(msb, np.uint16)
    C, H, W = msb.shape
    if msb_max is None:
        msb_max = int(msb.max())
    rowtab, coltab = pos_tables(H, W, cfg)
    F = cfg.feature_dim(C, D)
    if idx is None:
        n = H * W
        ip = None
    else:
        idx = _c(idx, np.int64)
        n = idx.size
        ip = _p(idx)
    out = np.empty((n, F), np.float32)
    rc = lib().orc_features(_p(msb), C, H, W, D, msb_max, int(cfg.use_colors), int(cfg.relative),
                            cfg.P, _p(rowtab), _p(coltab), ip, ctypes.c_int64(n), _p(out))
    assert rc == 0, rc
    return out


# ---------------------------------------------------------------- a5/a10


def param_count(F, bc, C, nl):
    return int(lib().orc_param_count(F, bc, C, nl))


def forward(params, F, bc, C, nl, x):
    params = _c(params, np.float32)
    x = _c(x, np.float32)
    assert params.size == param_count(F, bc, C, nl) and x.shape[1] == F
    y = np.empty((x.shape[0], C), np.float32)
    rc = lib().orc_forward(_p(params), F, bc, C, nl, _p(x), ctypes.c_int64(x.shape[0]), _p(y))
    assert rc == 0
    return y


def decode(msb, K, D, cfg, params, bc, nl, msb_max=None, want_y=False):
    """decode.py:73-134 -> reconstructed [C,H,W] u16 (and y [H*W,C] if want_y)."""
    msb = _c(msb, np.uint16)
    C, H, W = msb.shape
    if msb_max is None:
        msb_max = int(msb.max())
    rowtab, coltab = pos_tables(H, W, cfg)
    params = _c(params, np.float32)
    out = np.empty_like(msb)
    y = np.empty((H * W, C), np.float32) if want_y else None
    rc = lib().orc_decode(_p(msb), C, H, W, K, D, msb_max, int(cfg.use_colors), int(cfg.relative),
                          cfg.P, _p(rowtab), _p(coltab), _p(params), bc, nl, _p(out), _p(y))
    assert rc == 0
    return (out, y) if want_y else out


def eval_sse(msb, labels, D, cfg, params, bc, nl, msb_max=None):
    """LBDRNperformance.py:18-21 numerator: sum of squared error over the whole image (f64)."""
    msb = _c(msb, np.uint16)
    C, H, W = msb.shape
    if msb_max is None:
        msb_max = int(msb.max())
    rowtab, coltab = pos_tables(H, W, cfg)
    labels = _c(labels, np.float32)
    params = _c(params, np.float32)
    return float(lib().orc_eval_sse(_p(msb), _p(labels), C, H, W, D, msb_max,
                                    int(cfg.use_colors), int(cfg.relative), cfg.P, _p(rowtab),
                                    _p(coltab), _p(params), bc, nl))


def train_step(params, m, v, F, bc, C, nl, x, t, lr, step, apply_adam=True):
    """modified_ignite_engine.py:18-27 + Adam.  params/m/v updated in place (float32 arrays).
    Returns (loss, grads)."""
    for a in (params, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    x = _c(x, np.float32)
    t = _c(t, np.float32)
    grads = np.empty_like(params)
    loss = ctypes.c_double(0)
    rc = lib().orc_train_step(_p(params), _p(m), _p(v), F, bc, C, nl, _p(x), _p(t), x.shape[0],
                              ctypes.c_double(lr), int(step), int(apply_adam),
                              ctypes.byref(loss), _p(grads))
    assert rc == 0
    return loss.value, grads


def lr_schedule(lr, epochs):
    """StepLR(step_size=max(1,int(e/3)), gamma=0.1) stepped once per epoch (encode.py:85,98):
    learning rate in force during epoch e (0-based), chained float64 multiplies."""
    step = max(1, int(epochs / 3))
    out = []
    cur = float(lr)
    for e in range(epochs):
        out.append(cur)
        if (e + 1) % step == 0:
            cur = cur * 0.1
    return out


# ---------------------------------------------------------------- a12: header


def write_header(split_ratio, width, height, K, bc, nl, D, nn_bytes_list, base_bytes_list):
    """encode.py:37-64 as a byte string."""
    n = 8 + 3 * len(nn_bytes_list) + 4 * len(base_bytes_list)
    out = bytearray()
    out += n.to_bytes(1, "big")
    out += int(split_ratio).to_bytes(1, "big")
    out += int(width).to_bytes(2, "big")
    out += int(height).to_bytes(2, "big")
    out += (K * 16 + D).to_bytes(1, "big")
    out += (int(np.log2(bc)) * 16 + nl).to_bytes(1, "big")
    for b in nn_bytes_list:
        out += int(b).to_bytes(3, "big")
    for b in base_bytes_list:
        out += int(b).to_bytes(4, "big")
    assert len(out) == n
    return bytes(out)


def read_header(buf):
    """decode.py:25-53."""
    n = buf[0]
    sr = buf[1]
    width = int.from_bytes(buf[2:4], "big")
    height = int.from_bytes(buf[4:6], "big")
    K, D = buf[6] >> 4, buf[6] & 15
    bc, nl = 2 ** (buf[7] >> 4), buf[7] & 15
    p = 8
    nn, base = [], []
    for _ in range(sr * sr):
        nn.append(int.from_bytes(buf[p:p + 3], "big"))
        p += 3
    for _ in range(sr * sr):
        base.append(int.from_bytes(buf[p:p + 4], "big"))
        p += 4
    return n, sr, width, height, K, bc, nl, D, nn, base


# ---------------------------------------------------------------- a13: metrics


def decode_metrics(org, rec, total_bytes):
    """decode.py:215-222 -> (MSE f32, PSNR, bpsp)."""
    mse = np.mean((org.astype(np.float32) - rec.astype(np.float32)) ** 2)
    with np.errstate(divide="ignore"):
        psnr = 10 * np.log10(10000 ** 2 / mse)
    bpsp = total_bytes * 8 / np.prod(org.shape)
    return mse, psnr, bpsp


def sin_ref(x):
    return np.array([lib().orc_sin(float(v)) for v in np.asarray(x, np.float32).ravel()],
                    np.float32).reshape(np.shape(x))


def sigmoid_ref(x):
    return np.array([lib().orc_sigmoid(float(v)) for v in np.asarray(x, np.float32).ravel()],
                    np.float32).reshape(np.shape(x))


# ---------------------------------------------------------------- LBB2 MSB-plane payload (oracle/plane_codec.c)

def plane_encode(planes):
    """[C,H,W] uint16 -> (counts uint32[nstrips], words uint32[nw]) of the LBB2 body."""
    x = _c(planes, np.uint16)
    C, H, W = x.shape
    L = lib()
    L.orc_plane_nstrips.restype = ctypes.c_int64
    L.orc_plane_encode.restype = ctypes.c_int64
    ns = int(L.orc_plane_nstrips(C, H, W))
    counts = np.zeros(ns, np.uint32)
    cap = ns * (2 + (H + 63) // 64 * (1 + 64 * 64) + 64)
    words = np.zeros(cap, np.uint32)
    nw = int(L.orc_plane_encode(_p(x), C, H, W, _p(counts), _p(words), ctypes.c_int64(cap)))
    assert nw >= 0
    return counts, words[:nw].copy()


def plane_decode(counts, words, C, H, W):
    out = np.zeros((C, H, W), np.uint16)
    rc = lib().orc_plane_decode(_p(_c(counts, np.uint32)), _p(_c(words, np.uint32)), C, H, W, _p(out))
    if rc:
        raise ValueError(f"LBB2 stream rejected by the oracle decoder ({rc})")
    return out
