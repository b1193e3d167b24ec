#!/usr/bin/env python3
"""Fits the float32 polynomials of the device-native Gaussian generator (v2v_amd/csrc/v2v_rng.hpp: gauss16_*), and
checks the generator's accuracy exhaustively over its 2^16 radius and 2^16 angle inputs.  Build-time tool: the
coefficients it prints are pasted into v2v_rng.hpp and oracle/v2v_oracle.c (independent restatement)."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P

np.set_printoptions(precision=10)


def minimax_fit(fn, lo, hi, deg, w=None, iters=30, npts=4001):
    """Weighted-least-squares on Chebyshev nodes, iteratively re-weighted toward equi-ripple (Lawson)."""
    t = np.cos(np.pi * (np.arange(npts) + 0.5) / npts)
    x = 0.5 * (hi - lo) * t + 0.5 * (hi + lo)
    y = fn(x)
    wts = np.ones_like(x)
    best = None
    for _ in range(iters):
        c = C.chebfit(t, y, deg, w=np.sqrt(wts))
        err = np.abs(C.chebval(t, c) - y)
        if best is None or err.max() < best[1]:
            best = (c, err.max())
        wts = wts * (err / err.max() + 1e-3)
        wts /= wts.mean()
    c = best[0]
    # convert Chebyshev (in t) -> power series in x
    pt = C.cheb2poly(c)
    # t = (2x - (hi+lo))/(hi-lo) = a*x + b
    a, b = 2.0 / (hi - lo), -(hi + lo) / (hi - lo)
    px = np.zeros(1)
    lin = np.array([b, a])
    for k in range(len(pt) - 1, -1, -1):
        px = P.polyadd(P.polymul(px, lin), [pt[k]])
    return px, best[1]


def f32(x):
    return np.asarray(x, dtype=np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


if __name__ == "__main__":
    for deg in (6, 7, 8, 9):
        px, e = minimax_fit(lambda f: -2.0 * np.log1p(f), 0.0, 1.0, deg)
        print("L(f)=-2ln(1+f) deg", deg, "max err %.3g" % e)
    for deg in (3, 4):
        # sqrt2*sin(x)/x as polynomial in z = x^2 on [0,(pi/2)^2]
        px, e = minimax_fit(lambda z: np.sqrt(2) * np.sinc(np.sqrt(z) / np.pi), 0.0, (np.pi / 2) ** 2, deg)
        print("S(z) deg", deg, "max err %.3g" % e)
        px, e = minimax_fit(lambda z: np.sqrt(2) * np.cos(np.sqrt(z)), 0.0, (np.pi / 2) ** 2, deg)
        print("C(z) deg", deg, "max err %.3g" % e)


def hexf(v):
    return float(np.float32(v)).hex()


def emit():
    """Print the float32 coefficient tables (C99 hex literals) for v2v_rng.hpp / oracle/v2v_oracle.c."""
    pl, el = minimax_fit(lambda f: -2.0 * np.log1p(f), 0.0, 1.0, 7)
    ps, es = minimax_fit(lambda z: np.sqrt(2) * np.sinc(np.sqrt(z) / np.pi), 0.0, (np.pi / 2) ** 2, 3)
    pc, ec = minimax_fit(lambda z: np.sqrt(2) * np.cos(np.sqrt(z)), 0.0, (np.pi / 2) ** 2, 4)
    print("// L(f) = -2 ln(1+f), f in [0,1): degree 7, max abs err %.3g" % el)
    print("kGaussL[8] = {" + ", ".join(hexf(c) + "f" for c in pl) + "};")
    print("// S(z) = sqrt2 sin(x)/x, z = x^2, |x| <= pi/2: degree 3, max abs err %.3g" % es)
    print("kGaussS[4] = {" + ", ".join(hexf(c) + "f" for c in ps) + "};")
    print("// C(z) = sqrt2 cos(x): degree 4, max abs err %.3g" % ec)
    print("kGaussC[5] = {" + ", ".join(hexf(c) + "f" for c in pc) + "};")
    print("2ln2", hexf(2 * np.log(2.0)), "pi/65536", hexf(np.pi / 65536), "x0", hexf(-np.pi / 2 + np.pi / 131072))
    print("rsqrt: K=0x5f374000 a1", hexf(1.50118341), "a2", hexf(1.50093569), "b", hexf(0.50093571))


if __name__ == "__main__":
    emit()
