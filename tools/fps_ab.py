#!/usr/bin/env python3
"""A/B of the large-scene FPS variants selected by BTR_FPS_IMPL (results must be identical).
Usage: python tools/fps_ab.py [impl ...]   e.g.  python tools/fps_ab.py default onebar multi"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import scenes, timeit  # noqa: E402

impls = sys.argv[1:] or ["default", "onebar"]
for (N, M) in ((40000, 2048), (80000, 2048), (20000, 2048)):
    xyz = scenes(8, N)
    ref = None
    for impl in impls:
        if impl == "default":
            os.environ.pop("BTR_FPS_IMPL", None)
        else:
            os.environ["BTR_FPS_IMPL"] = impl
        out = _ext.furthest_point_sampling(xyz, M)
        med, mn = timeit(lambda: _ext.furthest_point_sampling(xyz, M), iters=8)
        same = True if ref is None else bool(torch.equal(out, ref))
        ref = out if ref is None else ref
        print("N=%6d M=%5d %-8s median %7.3f ms  min %7.3f ms  identical=%s" % (
            N, M, impl, med, mn, same))
os.environ.pop("BTR_FPS_IMPL", None)
