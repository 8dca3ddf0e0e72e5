"""Feature-switch configuration and the positional tables (host side, a3).

The reference keeps the switches as module globals in constants.py (ref constants.py:3-14) and
evaluates the positional features per pixel in float64 (ref LBDRNdataset.py:108-118).  Entry
[h,w,0,:] of its coordinate tensor depends on h only and [h,w,1,:] on w only, so this package
prepares two small float32 tables on the host ([H,P] and [W,P], P = 1+2*N_FREQ) with the very
same numpy expressions and the kernels gather from them.
"""
import numpy as np


class FeatCfg:
    def __init__(self, use_coordinates=False, embedding=False, sigma=1.4, n_freq=12,
                 use_colors=True, relative=True, activation="sine"):
        self.use_coordinates = bool(use_coordinates)
        self.embedding = bool(embedding)
        self.sigma = float(sigma)
        self.n_freq = int(n_freq)
        self.use_colors = bool(use_colors)
        self.relative = bool(relative)
        if activation not in ("sine", "relu"):
            raise ValueError(f"hidden activation {activation!r}: 'sine' (the reference's default) or 'relu' "
                             "(ref encode.py:75, decode.py:108)")
        self.activation = activation

    @classmethod
    def from_constants(cls, mod=None):
        """Read the switches from a constants module (default: the drop-in `constants`)."""
        if mod is None:
            import constants as mod
        return cls(mod.USE_COORDINATES, mod.EMBEDDING, mod.SIGMA, mod.N_FREQ, mod.USE_COLORS,
                   mod.RELATIVE, getattr(mod, "HIDDEN_ACTIVATION", "sine"))

    @property
    def act(self):
        """lbdrn_net.act (include/lbdrn_hip.h): LBDRN_ACT_SINE = 0, LBDRN_ACT_RELU = 1."""
        return 1 if self.activation == "relu" else 0

    @property
    def P(self):
        if not self.use_coordinates:
            return 0
        return 1 + 2 * self.n_freq if self.embedding else 1

    def feature_dim(self, C, D):
        side = 2 * D + 1
        return 2 * self.P + (C * side * side if self.use_colors else 0)


def _axis_table(n, cfg):
    with np.errstate(all="ignore"):
        pos = (2 * np.arange(n) / (n - 1) - 1).astype(np.float32)   # float64 ramp, stored as f32
        if not cfg.embedding:
            return np.ascontiguousarray(pos[:, None])
        freqs = cfg.sigma ** np.arange(cfg.n_freq) * np.pi           # float64
        phase = freqs[None, :] * pos[:, None]                        # f32 * f64 -> f64
        tab = np.concatenate([pos[:, None].astype(np.float64), np.sin(phase), np.cos(phase)], axis=1)
        return np.ascontiguousarray(tab.astype(np.float32))


def pos_tables(H, W, cfg):
    if cfg.P == 0:
        return np.zeros((H, 0), np.float32), np.zeros((W, 0), np.float32)
    return _axis_table(H, cfg), _axis_table(W, cfg)
