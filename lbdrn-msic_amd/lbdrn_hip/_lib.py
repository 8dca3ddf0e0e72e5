"""ctypes binding of liblbdrn_hip.so (C ABI: include/lbdrn_hip.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LBDRN_HIP_LIB selects another build of the same ABI (e.g. the diagnostic liblbdrn_hip_stamps.so)
_LIB_PATH = os.environ.get("LBDRN_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "liblbdrn_hip.so")

ABI_VERSION = 2           # LBDRN_ABI_VERSION of include/lbdrn_hip.h this binding was written against
PATH_AUTO, PATH_GENERIC, PATH_MFMA = 0, 1, 2
EVAL_BACKGROUND = 0x200   # hint OR'ed into `path` of lbdrn_eval_sse
EVAL_FAST = 0x400         # the per-epoch ranking pass in the tolerance arithmetic (lbdrn_hip.h)
EVAL_X16 = 0x1000         # with EVAL_FAST, opt-in: layer 0's colour features on the f16 matrix pipe, operands exact (lbdrn_hip.h)
TRAIN_ALONE = 0x800       # hint OR'ed into `path` of lbdrn_train_epoch: nothing else in flight on the device (lbdrn_hip.h)


class LbdrnError(RuntimeError):
    """A failed library call; `code` is the lbdrn_status it returned (None for host-side failures)."""
    code = None


E_ARG, E_DEVICE, E_UNSUPPORTED, E_WORKSPACE = -1, -2, -3, -4   # enum lbdrn_status


class Geom(ctypes.Structure):
    """struct lbdrn_geom"""
    _fields_ = [("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("K", ctypes.c_int32), ("D", ctypes.c_int32), ("msb_max", ctypes.c_int32),
                ("use_colors", ctypes.c_int32), ("relative", ctypes.c_int32),
                ("P", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("rowtab", ctypes.c_void_p), ("coltab", ctypes.c_void_p)]


class Net(ctypes.Structure):
    """struct lbdrn_net"""
    _fields_ = [("F", ctypes.c_int32), ("bc", ctypes.c_int32), ("C", ctypes.c_int32),
                ("nl", ctypes.c_int32), ("act", ctypes.c_int32)]


# name -> (restype, argtypes); must list every symbol include/lbdrn_hip.h declares
_vp, _i32, _i64, _sz, _dbl = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t,
                              ctypes.c_double)
_GP, _NP = ctypes.POINTER(Geom), ctypes.POINTER(Net)
SIGNATURES = {
    "lbdrn_last_error": (ctypes.c_char_p, []),
    "lbdrn_abi_version": (ctypes.c_int, []),
    "lbdrn_device_check": (ctypes.c_int, []),
    "lbdrn_param_count": (_i64, [_NP]),
    "lbdrn_feature_dim": (_i32, [_GP]),
    "lbdrn_split_bits": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "lbdrn_labels": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "lbdrn_features": (ctypes.c_int, [_GP, _vp, _vp, _i64, _vp, _vp]),
    "lbdrn_forward_workspace": (_sz, [_NP, _i64]),
    "lbdrn_forward": (ctypes.c_int, [_NP, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "lbdrn_apply_workspace": (_sz, [_GP, _NP]),
    "lbdrn_decode_fused": (ctypes.c_int, [_GP, _NP, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "lbdrn_eval_sse": (ctypes.c_int, [_GP, _NP, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "lbdrn_train_workspace": (_sz, [_GP, _NP, _i32]),
    "lbdrn_train_prepare": (ctypes.c_int, [_GP, _NP, _vp, _vp, _i32, _vp, _sz, _i32, _vp]),
    "lbdrn_train_epoch": (ctypes.c_int, [_GP, _NP, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64,
                                         _dbl, _vp, _vp, _sz, _i32, _vp]),
    "lbdrn_train_group_max": (ctypes.c_int, []),
    "lbdrn_train_step_features": (_i32, [_GP, _NP]),
    "lbdrn_train_group_size": (_i32, [_GP, _NP]),
    "lbdrn_train_epoch_group": (ctypes.c_int, [_i32, _vp, _NP, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _dbl, _vp,
                                               _vp, _sz, _i32, _vp]),
    "lbdrn_train_profile_mode": (ctypes.c_int, [_i32]),
    "lbdrn_randperm_workspace": (_sz, [_i64, _i32]),
    "lbdrn_randperm": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64), _i32, _i64, _vp, _vp, _sz, _vp]),
    "lbdrn_mt19937_jump_poly": (_i64, [_i32, _vp]),
    "lbdrn_plane_bound": (_sz, [_i32, _i32, _i32]),
    "lbdrn_plane_workspace": (_sz, [_i32, _i32, _i32]),
    "lbdrn_plane_encode": (ctypes.c_int, [_vp, _i32, _i32, _i32, _vp, _sz, _vp, _vp, _sz, _vp]),
    "lbdrn_plane_decode": (ctypes.c_int, [_vp, _sz, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "lbdrn_weights_bound": (_sz, [_i64]),
    "lbdrn_weights_encode": (ctypes.c_int, [_vp, _i64, _i32, _vp, _sz, ctypes.POINTER(ctypes.c_size_t)]),
    "lbdrn_weights_info": (ctypes.c_int, [_vp, _sz, ctypes.POINTER(_i64), ctypes.POINTER(_i32)]),
    "lbdrn_weights_decode": (ctypes.c_int, [_vp, _sz, _vp, _i64]),
    "lbdrn_train_step": (ctypes.c_int, [_NP, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _dbl, _i32, _vp,
                                        _vp, _vp, _sz, _vp]),
}

_lib = None


def lib_path():
    return _LIB_PATH


def lib():
    """Load liblbdrn_hip.so.  torch must be imported first so that the HIP runtime torch ships
    (same SONAME, libamdhip64.so.7) is the one both sides use."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise LbdrnError(
            f"{_LIB_PATH} is missing: build it with `python lbdrn-msic_amd/csrc/build.py` "
            "(hipcc, gfx950).  This package has no CPU fallback.")
    import torch  # noqa: F401  (loads libamdhip64 first)
    L = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if L.lbdrn_abi_version() != ABI_VERSION:
        raise LbdrnError(f"liblbdrn_hip ABI {L.lbdrn_abi_version()} != {ABI_VERSION}")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().lbdrn_last_error()
        err = LbdrnError(f"liblbdrn_hip error {rc}: {msg.decode() if msg else ''}")
        err.code = rc
        raise err
