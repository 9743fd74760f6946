#!/usr/bin/env python3
"""Per-op timing of the HIP kernels on synthetic scenes (HIP events, median of several runs).
Usage: python tools/bench_ops.py [fps] [bq] [group] ...   (default: all)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from backtoreality_amd.votenet import synthetic  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def scenes(B, N):
    return torch.from_numpy(np.stack([synthetic.make_scene(i, N, use_height=False)['point_clouds']
                                      for i in range(B)], 0)).cuda()


def main():
    which = set(sys.argv[1:]) or {"fps", "bq", "group"}
    B = 8
    xyz = scenes(B, 40000)
    levels = [(40000, 2048, 0.2, 64), (2048, 1024, 0.4, 32), (1024, 512, 0.8, 16),
              (512, 256, 1.2, 16)]
    cur = xyz
    pyramid = []
    for (N, M, r, S) in levels:
        inds = _ext.furthest_point_sampling(cur, M)
        new = torch.gather(cur, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        pyramid.append((cur, new, inds, N, M, r, S))
        cur = new
    if "fps" in which:
        for (src, new, inds, N, M, r, S) in pyramid:
            med, mn = timeit(lambda: _ext.furthest_point_sampling(src, M))
            print("fps        N=%6d M=%5d  median %8.3f ms  min %8.3f ms  (%.3f us/iter)" % (
                N, M, med, mn, 1e3 * mn / (M - 1)))
        os.environ["BTR_FPS_IMPL"] = "stream"
        med, mn = timeit(lambda: _ext.furthest_point_sampling(xyz, 2048), iters=3, warmup=1)
        print("fps-stream N= 40000 M= 2048  median %8.3f ms  min %8.3f ms" % (med, mn))
        del os.environ["BTR_FPS_IMPL"]
    if "bq" in which:
        for (src, new, inds, N, M, r, S) in pyramid:
            med, mn = timeit(lambda: _ext.ball_query(new, src, r, S))
            nbytes = B * (12 * N + 12 * M + 4 * M * S)
            print("ball_query N=%6d M=%5d S=%2d median %8.3f ms  min %8.3f ms  %7.1f GB/s alg" % (
                N, M, S, med, mn, nbytes / mn / 1e6))
    if "group" in which:
        for (src, new, inds, N, M, r, S), C in zip(pyramid, (1, 128, 256, 256)):
            idx = _ext.ball_query(new, src, r, S)
            feats = torch.randn(B, C, N, device="cuda")
            go = torch.randn(B, C, M, S, device="cuda")
            med, mn = timeit(lambda: _ext.group_points(feats, idx))
            med2, mn2 = timeit(lambda: _ext.group_points_grad(go, idx, N))
            print("group      C=%3d N=%6d M=%5d S=%2d fwd %8.3f ms  grad %8.3f ms" % (
                C, N, M, S, mn, mn2))


if __name__ == "__main__":
    main()
