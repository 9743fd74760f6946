#!/usr/bin/env python3
"""A/B of the register-resident FPS kernels (n <= 4096) on the levels of the VoteNet pyramid:
BTR_FPS_REGS=legacy vs the current kernel at BTR_FPS_REGS_NW in {1,4,8,16}; results must be
identical.  Usage: python tools/fps_small_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import scenes, timeit  # noqa: E402

xyz = scenes(8, 40000)
inds = _ext.furthest_point_sampling(xyz, 2048).long()
level = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
for (n, m) in ((2048, 1024), (1024, 512), (512, 256), (1024, 256), (4096, 1024), (256, 128),
               (64, 32)):
    pts = level[:, :n].contiguous() if n <= 2048 else torch.gather(
        xyz, 1, _ext.furthest_point_sampling(xyz, n).long().unsqueeze(-1).expand(-1, -1, 3)
    ).contiguous()
    ref = None
    for name, env in (("legacy", {"BTR_FPS_REGS": "legacy"}), ("nw1", {"BTR_FPS_REGS_NW": "1"}),
                      ("nw4", {"BTR_FPS_REGS_NW": "4"}), ("nw8", {"BTR_FPS_REGS_NW": "8"}),
                      ("nw16", {"BTR_FPS_REGS_NW": "16"})):
        for k in ("BTR_FPS_REGS", "BTR_FPS_REGS_NW"):
            os.environ.pop(k, None)
        os.environ.update(env)
        out = _ext.furthest_point_sampling(pts, m)
        med, mn = timeit(lambda: _ext.furthest_point_sampling(pts, m), iters=10)
        same = True if ref is None else bool(torch.equal(out, ref))
        ref = out if ref is None else ref
        print("n=%5d m=%5d %-7s median %7.3f ms  min %7.3f ms  %.3f us/step  identical=%s" % (
            n, m, name, med, mn, 1e3 * mn / (m - 1), same), flush=True)
for k in ("BTR_FPS_REGS", "BTR_FPS_REGS_NW"):
    os.environ.pop(k, None)
