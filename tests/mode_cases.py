"""Inputs on which the three roundings of a*a + b*b + c*c (BTR_FMAD 0 / 1 / 2, see
oracle/pointnet2_oracle.c) give DIFFERENT indices, so that a mode mix-up between a HIP
library and its oracle cannot pass unnoticed (on ordinary scenes all three modes agree)."""
from fractions import Fraction

import numpy as np


def sphere_cloud(seed, n, batch=2):
    """Points at (almost) the same distance from point 0: every FPS arg-max is decided in the
    last bits of the squared distance."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(batch):
        c = rng.uniform(1.0, 2.0, 3)
        v = rng.normal(size=(n, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        p = c + v * rng.uniform(0.999999, 1.000001, (n, 1))
        p[0] = c
        out.append(p)
    return np.stack(out).astype(np.float32)


def shell_ball_case(seed, n=4096, m=64, radius=0.75):
    """Centres with a thin shell of points at distance ~radius: membership in the ball
    (d2 < radius^2) is decided in the last bits."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(1.0, 3.0, (1, m, 3))
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    owner = rng.integers(0, m, n)
    pts = centres[0, owner] + v * radius * rng.uniform(0.9999995, 1.0000005, (n, 1))
    return centres.astype(np.float32), pts[None].astype(np.float32), np.float32(radius)


def _round_f32(x):
    """Correctly rounded float32 of an exact Fraction (ties to even)."""
    f = np.float32(float(x))
    lo = np.nextafter(f, np.float32(-np.inf), dtype=np.float32)
    hi = np.nextafter(f, np.float32(np.inf), dtype=np.float32)
    best = None
    for c in (lo, f, hi):
        err = abs(Fraction(float(c)) - x)
        even = (int(np.float32(c).view(np.uint32)) & 1) == 0
        key = (err, 0 if even else 1)
        if best is None or key < best[0]:
            best = (key, c)
    return best[1]


def sq3_exact(a, b, c, mode):
    """a*a + b*b + c*c of three float32 under rounding mode `mode`, in exact rational
    arithmetic with one correct rounding per machine operation."""
    A, B, C = (Fraction(float(np.float32(v))) for v in (a, b, c))
    r = _round_f32
    if mode == 0:
        return r(Fraction(float(r(Fraction(float(r(A * A))) + Fraction(float(r(B * B)))))) +
                 Fraction(float(r(C * C))))
    if mode == 1:
        return r(C * C + Fraction(float(r(A * A + Fraction(float(r(B * B)))))))
    return r(C * C + Fraction(float(r(B * B + Fraction(float(r(A * A)))))))
