#!/usr/bin/env python3
"""CPU simulation of the bucketed FPS's per-step work (no GPU): how many buckets pass the box
test per sample, and how many trips the busiest of 16 waves makes, for 64-point buckets (one
bucket per trip: the current kernel) and for 32-point buckets processed two per trip (two
half-waves).  Decides whether the half-wave variant is worth building.
Usage: python tools/probe/fps_trips_sim.py [scene-index]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from backtoreality_amd.votenet import synthetic  # noqa: E402


def spread3(v):
    v = v.astype(np.uint64) & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


def hilbert3(x0, x1, x2, bits=5):
    X = [x0.astype(np.uint64).copy(), x1.astype(np.uint64).copy(), x2.astype(np.uint64).copy()]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = np.uint64(Q - 1)
        for i in range(3):
            hit = (X[i] & np.uint64(Q)) != 0
            t = (X[0] ^ X[i]) & P
            X0n = np.where(hit, X[0] ^ P, X[0] ^ t)
            Xin = np.where(hit, X[i], X[i] ^ t)
            if i == 0:
                X[0] = np.where(hit, X[0] ^ P, X[0])   # (X[0]^X[0]) & P = 0: unchanged
            else:
                X[0], X[i] = X0n, Xin
        Q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & np.uint64(Q)) != 0, t ^ np.uint64(Q - 1), t)
        Q >>= 1
    X = [x ^ t for x in X]
    return (spread3(X[0]) << np.uint64(2)) | (spread3(X[1]) << np.uint64(1)) | spread3(X[2])


def simulate(pts, m, bsize, nwaves=16, per_trip=1):
    n = len(pts)
    mn, mx = pts.min(0), pts.max(0)
    scale = 32.0 / np.maximum(mx - mn, 1e-9)
    q = np.clip(((pts - mn) * scale).astype(np.int64), 0, 31)
    order = np.argsort(hilbert3(q[:, 0], q[:, 1], q[:, 2]), kind="stable")
    sp = pts[order]
    nb = (n + bsize - 1) // bsize
    pad = nb * bsize - n
    spp = np.concatenate([sp, np.repeat(sp[-1:], pad, 0)]) if pad else sp
    b3 = spp.reshape(nb, bsize, 3)
    lo, hi = b3.min(1), b3.max(1)
    tmin = np.full(nb * bsize, 1e10)
    bmax = np.full(nb, 1e10)
    cur = pts[0]
    touched_tot, trips_tot, trips_max = 0, 0, 0
    for _ in range(1, m):
        c = np.clip(cur, lo, hi)
        dbox = ((c - cur) ** 2).sum(1)
        act = np.nonzero(dbox < bmax)[0]
        touched_tot += act.size
        per_wave = np.bincount(act % nwaves, minlength=nwaves)
        trips = -(-per_wave // per_trip)
        trips_tot += trips.sum()
        trips_max += trips.max()
        for b in act:
            sl = slice(b * bsize, (b + 1) * bsize)
            d = ((spp[sl] - cur) ** 2).sum(1)
            tmin[sl] = np.minimum(tmin[sl], d)
            bmax[b] = tmin[sl].max()
        best = int(np.argmax(bmax))
        sl = slice(best * bsize, (best + 1) * bsize)
        cur = spp[best * bsize + int(np.argmax(tmin[sl]))]
    steps = m - 1
    return touched_tot / steps, trips_max / steps


scene = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pts = synthetic.make_scene(scene, 40000)['point_clouds'][:, :3].astype(np.float64)
for bsize, per_trip in ((64, 1), (32, 2), (32, 1), (64, 2)):
    t, mx = simulate(pts, 2048, bsize, per_trip=per_trip)
    print("bucket %2d points, %d per trip: %.2f touched buckets / step, slowest wave %.2f trips / "
          "step" % (bsize, per_trip, t, mx), flush=True)
