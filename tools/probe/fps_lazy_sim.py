#!/usr/bin/env python3
"""CPU simulation of a LAZY variant of the bucketed FPS (no GPU): bucket bests are kept as stale
upper bounds; per step every wave refreshes its own largest-bound stale bucket (one L2 round trip,
applying the samples chosen since that bucket's last refresh), then the block arg-max decides --
if the maximum is still a stale bound, its owner refreshes it and the block votes again.
Counts rounds (dependent trips) per step against the eager kernel's busiest-wave trips.
Usage: python tools/probe/fps_lazy_sim.py [scene-index] [m]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "probe"))
from backtoreality_amd.votenet import synthetic  # noqa: E402
from fps_trips_sim import hilbert3  # noqa: E402


def main():
    scene = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    pts = synthetic.make_scene(scene, 40000, use_height=False)['point_clouds'].astype(np.float64)
    n, bsize, nw = len(pts), 64, 16
    mn, mx = pts.min(0), pts.max(0)
    q = np.clip(((pts - mn) * (32.0 / np.maximum(mx - mn, 1e-9))).astype(np.int64), 0, 31)
    order = np.argsort(hilbert3(q[:, 0], q[:, 1], q[:, 2]), kind="stable")
    sp = pts[order]
    nb = (n + bsize - 1) // bsize
    pad = nb * bsize - n
    spp = np.concatenate([sp, np.repeat(sp[-1:], pad, 0)]) if pad else sp
    b3 = spp.reshape(nb, bsize, 3)
    lo, hi = b3.min(1), b3.max(1)
    owner = np.arange(nb) % nw
    samples = [pts[0]]
    tmin = ((b3 - pts[0]) ** 2).sum(2)          # prologue: everything fresh w.r.t. sample 0
    bound = tmin.max(1)                          # per-bucket best (exact now)
    fresh_at = np.zeros(nb, dtype=np.int64)      # number of samples applied
    rounds_tot = refresh_tot = applied_tot = eager_touched = eager_max = 0
    hist = np.zeros(12, dtype=np.int64)

    def refresh(b):
        nonlocal applied_tot
        t0 = fresh_at[b]
        for s in samples[t0:]:
            c = np.clip(s, lo[b], hi[b])
            if ((c - s) ** 2).sum() < bound[b]:
                d = ((b3[b] - s) ** 2).sum(1)
                tmin[b] = np.minimum(tmin[b], d)
                applied_tot += 1
        bound[b] = tmin[b].max()
        fresh_at[b] = len(samples)

    for step in range(1, m):
        # what the eager kernel does this step (for comparison)
        cur = samples[-1]
        c = np.clip(cur, lo, hi)
        true_best = tmin.max(1)    # (eager state differs, but the touched count is the box test)
        rounds = 0
        first = True
        while True:
            stale = fresh_at < len(samples)
            gmax_b = int(np.argmax(bound))
            if not stale[gmax_b]:
                break
            rounds += 1
            if first:      # every wave refreshes its own largest stale bound
                for w in range(nw):
                    mine = np.nonzero((owner == w) & stale)[0]
                    if mine.size:
                        b = mine[np.argmax(bound[mine])]
                        refresh(b)
                        refresh_tot += 1
                first = False
            else:          # later rounds: each wave whose top bound is stale AND above the best
                fresh_best = bound[~stale].max() if (~stale).any() else -1.0
                for w in range(nw):
                    mine = np.nonzero((owner == w) & stale & (bound > fresh_best))[0]
                    if mine.size:
                        b = mine[np.argmax(bound[mine])]
                        refresh(b)
                        refresh_tot += 1
        rounds_tot += rounds
        hist[min(rounds, 11)] += 1
        b = int(np.argmax(bound))
        k = int(np.argmax(tmin[b]))
        samples.append(b3[b, k].copy())
    steps = m - 1
    print("scene %d, %d steps: rounds/step %.2f, refreshes/step %.1f, sample applications/step %.1f"
          % (scene, steps, rounds_tot / steps, refresh_tot / steps, applied_tot / steps))
    print("rounds histogram:", hist.tolist())


if __name__ == "__main__":
    main()
