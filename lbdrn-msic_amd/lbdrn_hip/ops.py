"""torch-tensor level wrappers of the C ABI: device pointers from tensor.data_ptr(), the stream
from torch.cuda.current_stream(); torch is plumbing (memory + streams), all arithmetic is in
liblbdrn_hip.so."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import EVAL_BACKGROUND, EVAL_FAST, EVAL_X16, TRAIN_ALONE, Geom, Net, PATH_AUTO, check, lib


def _call(fn, ref, *args):
    """One C entry point, asynchronous on the current stream of the device `ref` (a tensor or a device) lives on.
    The library launches on the stream it is handed and sizes its kernels from the thread's current HIP device,
    so both are taken from the data, never from whatever device happens to be current (one process may hold
    tensors of several GPUs; torchrun ranks > 0 must not fall back to cuda:0)."""
    dev = ref.device if isinstance(ref, torch.Tensor) else torch.device(ref)
    if dev.type != "cuda":
        raise _lib.LbdrnError("liblbdrn_hip works on device (HBM) tensors only; there is no CPU path in this package")
    with torch.cuda.device(dev):
        check(fn(*args, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.LbdrnError("liblbdrn_hip works on device (HBM) tensors only; there is no "
                                  "CPU path in this package")


def _u16(t):
    """torch has no arithmetic on uint16; planes are carried as int16 storage with the same bits."""
    if t.dtype == torch.uint16:
        return t.view(torch.int16)
    assert t.dtype == torch.int16, t.dtype
    return t


class FeatureGeometry:
    """lbdrn_geom plus the device tables it points at (kept alive here)."""

    def __init__(self, C, H, W, K, D, msb_max, cfg, device):
        from .features import pos_tables
        self.C, self.H, self.W, self.K, self.D, self.msb_max = C, H, W, K, D, int(msb_max)
        self.cfg = cfg
        rowtab, coltab = pos_tables(H, W, cfg)
        self.P = rowtab.shape[1]
        self.rowtab = torch.from_numpy(rowtab).to(device) if self.P else None
        self.coltab = torch.from_numpy(coltab).to(device) if self.P else None
        self.F = cfg.feature_dim(C, D)
        self.c = Geom(C, H, W, K, D, int(msb_max), int(cfg.use_colors), int(cfg.relative), self.P,
                      0, self.rowtab.data_ptr() if self.P else None,
                      self.coltab.data_ptr() if self.P else None)

    def with_max(self, msb_max):
        self.msb_max = int(msb_max)
        self.c.msb_max = int(msb_max)
        return self


ACT_SINE, ACT_RELU = 0, 1     # lbdrn_net.act


def make_net(F, bc, C, nl, act=ACT_SINE):
    return Net(F, bc, C, nl, act)


def param_count(net):
    return int(lib().lbdrn_param_count(ctypes.byref(net)))


def split_bits(img, K):
    """img [C,H,W] u16 (device) -> (msb [C,H,W] u16, msb_max int).  ref LBDRNdataset.py:95-101"""
    _need_cuda(img)
    img = _u16(img.contiguous())
    C, H, W = img.shape
    msb = torch.empty_like(img)
    mx = torch.zeros(1, dtype=torch.int32, device=img.device)
    _call(lib().lbdrn_split_bits, img, _ptr(img), C, H, W, K, _ptr(msb), _ptr(mx))
    return msb, int(mx.item())


def labels(img, K, idx=None):
    """[n,C] f32 normalised low bits.  ref LBDRNdataset.py:96-97,131"""
    _need_cuda(img, idx)
    img = _u16(img.contiguous())
    C, H, W = img.shape
    n = H * W if idx is None else idx.numel()
    out = torch.empty((n, C), dtype=torch.float32, device=img.device)
    if idx is not None:
        idx = idx.to(torch.int64).contiguous()
    _call(lib().lbdrn_labels, img, _ptr(img), C, H, W, K, _ptr(idx), n, _ptr(out))
    return out


def features(geom, msb, idx=None):
    """[n,F] f32 feature rows.  ref LBDRNdataset.py:104-130"""
    _need_cuda(msb, idx)
    msb = _u16(msb.contiguous())
    n = geom.H * geom.W if idx is None else idx.numel()
    out = torch.empty((n, geom.F), dtype=torch.float32, device=msb.device)
    if idx is not None:
        idx = idx.to(torch.int64).contiguous()
    _call(lib().lbdrn_features, msb, ctypes.byref(geom.c), _ptr(msb), _ptr(idx), n, _ptr(out))
    return out


def forward(net, params, x):
    """LBDRNModel.forward on a feature matrix.  ref LBDRNmodel.py:79-82"""
    _need_cuda(params, x)
    x = x.contiguous().float()
    params = params.contiguous().float()
    B = x.shape[0]
    assert x.shape[1] == net.F and params.numel() == param_count(net)
    y = torch.empty((B, net.C), dtype=torch.float32, device=x.device)
    nbytes = lib().lbdrn_forward_workspace(ctypes.byref(net), B)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    _call(lib().lbdrn_forward, x, ctypes.byref(net), _ptr(params), _ptr(x), B, _ptr(y), _ptr(ws), nbytes)
    return y


class ApplyWorkspace:
    def __init__(self, geom, net, device):
        self.nbytes = lib().lbdrn_apply_workspace(ctypes.byref(geom.c), ctypes.byref(net))
        self.buf = torch.empty(max(self.nbytes, 1), dtype=torch.uint8, device=device)


def decode_fused(geom, net, msb, params, want_y=False, path=PATH_AUTO, ws=None):
    """decode.py:73-134 in one call -> image [C,H,W] int16-storage u16 (and y [N,C])."""
    _need_cuda(msb, params)
    msb = _u16(msb.contiguous())
    params = params.contiguous().float()
    out = torch.empty_like(msb)
    y = torch.empty((geom.H * geom.W, geom.C), dtype=torch.float32, device=msb.device) if want_y else None
    ws = ws or ApplyWorkspace(geom, net, msb.device)
    _call(lib().lbdrn_decode_fused, msb, ctypes.byref(geom.c), ctypes.byref(net), _ptr(msb), _ptr(params),
                                   _ptr(out), _ptr(y), _ptr(ws.buf), ws.nbytes, path)
    return (out, y) if want_y else out


def eval_sse(geom, net, img, msb, params, path=PATH_AUTO, ws=None, out=None, background=False, fast=False, x16=False):
    """Whole-image sum of squared error as a device float64 scalar tensor (no sync).
    background: launch on half as many workgroups (LBDRN_EVAL_BACKGROUND, lbdrn_hip.h); same sum bit for bit.
    fast: the epoch-ranking pass in the training step's arithmetic (LBDRN_EVAL_FAST: within 1e-6 relative of the
    canonical sum, a fifth less time); the default is the canonical arithmetic of the decode kernels.
    x16 (with fast; opt-in): layer 0's colour features on the f16 matrix pipe with exact operands (LBDRN_EVAL_X16)."""
    _need_cuda(img, msb, params)
    img = _u16(img.contiguous())
    msb = _u16(msb.contiguous())
    params = params.contiguous().float()
    sse = out if out is not None else torch.zeros(1, dtype=torch.float64, device=msb.device)
    ws = ws or ApplyWorkspace(geom, net, msb.device)
    _call(lib().lbdrn_eval_sse, msb, ctypes.byref(geom.c), ctypes.byref(net), _ptr(img), _ptr(msb),
                               _ptr(params), _ptr(sse), _ptr(ws.buf), ws.nbytes,
                               path | (EVAL_BACKGROUND if background else 0) | (EVAL_FAST if fast else 0) |
                               (EVAL_X16 if (fast and x16) else 0))
    return sse


class TrainWorkspace:
    """Device scratch of the training path; prepare() builds the per-image state in it once."""

    def __init__(self, geom, net, batch_size, device):
        self.geom, self.net, self.batch_size = geom, net, batch_size
        self.nbytes = lib().lbdrn_train_workspace(ctypes.byref(geom.c), ctypes.byref(net), batch_size)
        self.buf = torch.empty(max(self.nbytes, 1), dtype=torch.uint8, device=device)
        self.prepared_for = None

    def prepare(self, img, msb, path=PATH_AUTO):
        img, msb = _u16(img.contiguous()), _u16(msb.contiguous())
        _call(lib().lbdrn_train_prepare, img, ctypes.byref(self.geom.c), ctypes.byref(self.net), _ptr(img),
                                        _ptr(msb), self.batch_size, _ptr(self.buf), self.nbytes, path)
        self.prepared_for = (img.data_ptr(), path)
        return self


def train_epoch(geom, net, img, msb, perm, batch_size, params, exp_avg, exp_avg_sq, adam_step0, lr,
                losses=None, path=PATH_AUTO, ws=None, alone=False):
    """One trainer epoch in place (encode.py:157 inner loop).  perm: int64 device tensor.  alone: nothing else of
    weight is in flight on the device (LBDRN_TRAIN_ALONE, lbdrn_hip.h: a performance hint, same numbers)."""
    _need_cuda(img, msb, perm, params, exp_avg, exp_avg_sq, losses)
    img = _u16(img.contiguous())
    msb = _u16(msb.contiguous())
    assert perm.dtype == torch.int64 and perm.is_contiguous()
    for t in (params, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous()
    if ws is None:
        ws = TrainWorkspace(geom, net, batch_size, img.device).prepare(img, msb, path)
    if ws.prepared_for != (img.data_ptr(), path):
        raise _lib.LbdrnError("TrainWorkspace.prepare(img, msb, path) must run once for this image first")
    _call(lib().lbdrn_train_epoch, img, ctypes.byref(geom.c), ctypes.byref(net), _ptr(img), _ptr(msb),
                                  _ptr(perm), perm.numel(), batch_size, _ptr(params), _ptr(exp_avg),
                                  _ptr(exp_avg_sq), adam_step0, float(lr), _ptr(losses), _ptr(ws.buf),
                                  ws.nbytes, path | (TRAIN_ALONE if alone else 0))


def train_group_max():
    return int(lib().lbdrn_train_group_max())


def train_group_size(C, H, W, K, D, cfg, base_channel, num_layers):
    """How many fits of this shape step in one launch per minibatch (lbdrn_train_group_size; 1: grouping gains nothing)."""
    g = Geom(C, H, W, K, D, 1, int(cfg.use_colors), int(cfg.relative), int(cfg.P), 0, None, None)
    net = Net(cfg.feature_dim(C, D), base_channel, C, num_layers, cfg.act)
    return int(lib().lbdrn_train_group_size(ctypes.byref(g), ctypes.byref(net)))


def train_step_features(geom, net):
    """Input features the fused training step of this shape multiplies (F, or F - C where the window centres are
    exact zeros and the step leaves them out: lbdrn_train_step_features)."""
    return int(lib().lbdrn_train_step_features(ctypes.byref(geom.c), ctypes.byref(net)))


def train_epoch_group(geoms, net, imgs, msbs, perms, batch_size, params, exp_avgs, exp_avg_sqs, adam_step0, lr,
                      losses=None, path=PATH_AUTO, wss=None):
    """train_epoch for several independent fits of ONE shape, stepping side by side (lbdrn_train_epoch_group: one
    launch per minibatch for the whole group where the fused step takes groups).  Lists, one entry per fit; every
    workspace prepared for its image.  Same numbers as one train_epoch call per fit."""
    n = len(imgs)
    _need_cuda(*imgs, *msbs, *perms, *params, *exp_avgs, *exp_avg_sqs)
    imgs = [_u16(t.contiguous()) for t in imgs]
    msbs = [_u16(t.contiguous()) for t in msbs]
    for t in perms:
        assert t.dtype == torch.int64 and t.is_contiguous() and t.numel() == perms[0].numel()
    for t in list(params) + list(exp_avgs) + list(exp_avg_sqs):
        assert t.dtype == torch.float32 and t.is_contiguous()
    for ws, img in zip(wss, imgs):
        if ws.prepared_for != (img.data_ptr(), path) or ws.nbytes != wss[0].nbytes:
            raise _lib.LbdrnError("every fit of a group needs its own TrainWorkspace, prepared for its image")
    arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
    garr = (ctypes.POINTER(Geom) * n)(*[ctypes.pointer(g.c) for g in geoms])
    larr = arr([_ptr(t) for t in losses]) if losses is not None else None
    _call(lib().lbdrn_train_epoch_group, imgs[0], n, ctypes.cast(garr, ctypes.c_void_p), ctypes.byref(net),
          arr([_ptr(t) for t in imgs]), arr([_ptr(t) for t in msbs]), arr([_ptr(t) for t in perms]),
          perms[0].numel(), batch_size, arr([_ptr(t) for t in params]), arr([_ptr(t) for t in exp_avgs]),
          arr([_ptr(t) for t in exp_avg_sqs]), adam_step0, float(lr), larr, arr([_ptr(w.buf) for w in wss]),
          wss[0].nbytes, path)


def train_profile_mode(mode):
    """Measurement aid: mode 1 makes train_epoch launch every reduce/Adam kernel twice (see lbdrn_hip.h)."""
    check(lib().lbdrn_train_profile_mode(int(mode)))


def train_step(net, x, t, params, exp_avg, exp_avg_sq, adam_step, lr, apply_adam=True):
    """One teacher-forced update on an explicit minibatch; returns (loss tensor, grads tensor)."""
    _need_cuda(x, t, params)
    x = x.contiguous().float()
    t = t.contiguous().float()
    B = x.shape[0]
    loss = torch.zeros(1, dtype=torch.float32, device=x.device)
    grads = torch.empty_like(params)
    g = Geom(net.C, 1, 1, 1, 0, 1, 1, 1, 0, 0, None, None)
    nbytes = lib().lbdrn_train_workspace(ctypes.byref(g), ctypes.byref(net), B)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    _call(lib().lbdrn_train_step, x, ctypes.byref(net), _ptr(x), _ptr(t), B, _ptr(params), _ptr(exp_avg),
                                 _ptr(exp_avg_sq), adam_step, float(lr), int(apply_adam), _ptr(loss),
                                 _ptr(grads), _ptr(ws), nbytes)
    return loss, grads


def randperm(seeds, n, device):
    """[len(seeds), n] int64: row c = torch.randperm(n, generator=Generator().manual_seed(seeds[c])),
    computed on the GPU in one call (a scalar seed gives a 1-D tensor)."""
    scalar = isinstance(seeds, int)
    seeds = [seeds] if scalar else list(seeds)
    out = torch.empty((len(seeds), n), dtype=torch.int64, device=device)
    for c0 in range(0, len(seeds), 32):
        chunk = seeds[c0:c0 + 32]
        arr = (ctypes.c_uint64 * len(chunk))(*[s & 0xFFFFFFFFFFFFFFFF for s in chunk])
        nbytes = lib().lbdrn_randperm_workspace(n, len(chunk))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _call(lib().lbdrn_randperm, out, arr, len(chunk), n, _ptr(out[c0:]), _ptr(ws), nbytes)
    return out[0] if scalar else out


def plane_encode(planes):
    """[C,H,W] uint16 planes in HBM -> LBB2 body as host bytes (counts + words; container.py adds the tag).
    One host sync: the body's length decides how much is copied back."""
    _need_cuda(planes)
    planes = _u16(planes.contiguous())
    C, H, W = planes.shape
    dev = planes.device
    body = torch.empty(lib().lbdrn_plane_bound(C, H, W), dtype=torch.uint8, device=dev)
    nbytes = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(lib().lbdrn_plane_workspace(C, H, W), 1), dtype=torch.uint8, device=dev)
    _call(lib().lbdrn_plane_encode, planes, _ptr(planes), C, H, W, _ptr(body), body.numel(), _ptr(nbytes), _ptr(ws),
                                   ws.numel())
    return body[:int(nbytes.item())].cpu().numpy().tobytes()


def plane_decode(body, C, H, W, device):
    """LBB2 body (bytes) -> [C,H,W] uint16 planes in HBM (int16 storage).  Raises on a malformed stream."""
    raw = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(device)
    planes = torch.empty((C, H, W), dtype=torch.int16, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    ws = torch.empty(max(lib().lbdrn_plane_workspace(C, H, W), 1), dtype=torch.uint8, device=device)
    _call(lib().lbdrn_plane_decode, planes, _ptr(raw), raw.numel(), C, H, W, _ptr(planes), _ptr(status), _ptr(ws),
                                   ws.numel())
    if int(status.item()):
        raise _lib.LbdrnError("LBB2 payload is inconsistent with its geometry (corrupt or truncated stream)")
    return planes


def to_device_u16(arr, device):
    """numpy uint16 array -> device tensor (int16 storage, same bits)."""
    a = np.ascontiguousarray(arr, dtype=np.uint16)
    return torch.from_numpy(a.view(np.int16)).to(device)


def from_device_u16(t):
    return t.cpu().numpy().view(np.uint16)
