"""Derive the polynomial coefficients of the canonical sin/cos/exp kernels.

TEST INFRASTRUCTURE (oracle/): documents where the constants in
oracle/lbdrn_oracle.c come from.  The HIP library carries its own copy of the
same numbers (lbdrn-msic_amd/csrc/lbdrn_math.hpp); nothing in the product
imports this file.

Method: weighted least squares on Chebyshev nodes in float64 (near-minimax),
coefficients rounded to float32, then the float32 evaluation scheme is
replayed with numpy float32 + exact fma emulation in float64 to report the
max ulp error against float64 libm.
"""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (2 * k + 1) / (2 * n))
    return 0.5 * (a + b) + 0.5 * (b - a) * x

def fit(fun, a, b, deg, n=4000, weight=None):
    x = cheb_nodes(a, b, n)
    y = fun(x)
    V = np.vander(x, deg + 1, increasing=True)
    w = np.ones_like(x) if weight is None else weight(x)
    c, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
    return c

def f32(x):
    return np.float32(x)

if __name__ == "__main__":
    np.set_printoptions(precision=17)
    q = (np.pi / 4) * 1.02
    # sin(r) = r + r*s*P(s), s=r^2, s in [0, q^2]:  P(s) = (sin(r)/r - 1)/s
    def P(s):
        r = np.sqrt(s)
        return (np.sin(r) / r - 1.0) / s
    cs = fit(P, 1e-12, q * q, 3)
    # cos(r) = 1 + s*Q(s):  Q(s) = (cos(r) - 1)/s
    def Q(s):
        r = np.sqrt(s)
        return (np.cos(r) - 1.0) / s
    cc = fit(Q, 1e-12, q * q, 4)
    # exp(r) = 1 + r + r^2*E(r) on [-ln2/2, ln2/2]
    h = np.log(2) / 2 * 1.02
    def E(r):
        return (np.expm1(r) - r) / (r * r)
    xs = cheb_nodes(-h, h, 4001)
    xs = xs[np.abs(xs) > 1e-9]
    V = np.vander(xs, 5, increasing=True)
    ce, *_ = np.linalg.lstsq(V, E(xs), rcond=None)
    print("SIN", [float(f32(c)).hex() for c in cs], [repr(float(f32(c))) for c in cs])
    print("COS", [float(f32(c)).hex() for c in cc], [repr(float(f32(c))) for c in cc])
    print("EXP", [float(f32(c)).hex() for c in ce], [repr(float(f32(c))) for c in ce])
    # pi/2 split in three float32 pieces, ln2 split in two
    p = np.pi / 2
    hi = f32(p); mid = f32(p - float(hi))
    import decimal
    decimal.getcontext().prec = 60
    PI2 = decimal.Decimal("1.57079632679489661923132169163975144209858469968755291")
    lo = f32(float(PI2 - decimal.Decimal(float(hi)) - decimal.Decimal(float(mid))))
    print("PIO2", float(hi).hex(), float(mid).hex(), float(lo).hex())
    LN2 = decimal.Decimal("0.693147180559945309417232121458176568075500134360255254")
    lhi = f32(float(LN2)); llo = f32(float(LN2 - decimal.Decimal(float(lhi))))
    print("LN2", float(lhi).hex(), float(llo).hex())
    print("2/pi", float(f32(2 / np.pi)).hex(), "log2e", float(f32(1 / np.log(2))).hex())
