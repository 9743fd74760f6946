#!/usr/bin/env python3
"""A/B of the ball-query grid build: fused one-launch build (default) vs BTR_BQ_BUILD=multi.
The env var is read once per process, so each variant runs in its own process:
    python tools/bq_ab.py; BTR_BQ_BUILD=multi python tools/bq_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import scenes, timeit  # noqa: E402

for (B, N, M, r, S, scale) in ((8, 40000, 2048, 0.2, 64, 1.0), (4, 50000, 2048, 0.2, 64, 1.0),
                               (4, 80000, 2048, 0.2, 64, 1.7), (8, 20000, 2048, 0.2, 64, 1.0)):
    xyz = scenes(B, N) * (torch.tensor([scale, scale, 1.0], device="cuda"))
    xyz = xyz.contiguous()
    inds = _ext.furthest_point_sampling(xyz, M).long()
    new_xyz = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    if os.environ.get("BQ_SORT"):   # experiment: centres in Morton order (what would a spatial
        # order of the centres buy the bucket query?)
        lo = xyz.amin(1, keepdim=True)
        q = ((new_xyz - lo) / (xyz.amax(1, keepdim=True) - lo + 1e-9) * 1023).long().clamp(0, 1023)

        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        key = spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
        order = key.argsort(1)
        new_xyz = torch.gather(new_xyz, 1, order.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = _ext.ball_query(new_xyz, xyz, r, S)
    med, mn = timeit(lambda: _ext.ball_query(new_xyz, xyz, r, S), iters=20)
    print("B=%d N=%d scale %.1f build=%s: median %.1f us min %.1f us  checksum %d" % (
        B, N, scale, os.environ.get("BTR_BQ_BUILD", "fused"), med * 1e3, mn * 1e3,
        int(idx.long().sum())), _ext.BQ_CALLS, flush=True)
