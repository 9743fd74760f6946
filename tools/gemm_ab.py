#!/usr/bin/env python3
"""Timing of the SA GEMM kernels at the shapes of the benchmark step (HIP events)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

lib = _ext._lib
SHAPES = [(1048576, 64, 4, False, True), (1048576, 64, 64, True, True),
          (1048576, 128, 64, True, True), (1048576, 64, 128, False, False),
          (262144, 128, 132, False, True), (262144, 256, 128, True, True),
          (262144, 128, 256, False, False), (65536, 256, 128, True, True)]


def run(rows, n, k, pro, stats):
    dev = torch.device("cuda")
    a = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev)
    c = torch.empty(rows, n, device=dev)
    pa = torch.rand(k, device=dev) if pro else None
    pb = torch.rand(k, device=dev) if pro else None
    grid = lib.btr_sa_gemm_grid(rows)
    part = torch.empty(grid, 2, n, device=dev) if stats else None
    p = _ext._p
    st = torch.cuda.current_stream().cuda_stream

    def fn():
        rc = lib.btr_sa_gemm_nt(rows, n, k, p(a), k, p(w), k, p(c), n, p(pa), p(pb), p(part), st)
        assert rc == 0
    med, mn = timeit(fn, iters=20, warmup=3)
    flops = 2.0 * rows * n * k
    nbytes = 4.0 * (rows * (n + k) + n * k)
    return med, flops / med / 1e9, nbytes / med / 1e6


if __name__ == "__main__":
    for shp in SHAPES:
        med, tf, gbs = run(*shp)
        print("gemm_nt rows=%8d n=%4d k=%4d pro=%d stats=%d  %7.3f ms  %6.1f TF  %6.0f GB/s" % (
            shp + (med, tf, gbs)))
