#!/usr/bin/env python3
"""Is the NT GEMM bound by its C stores?  Same launches with and without the output tensor
(statistics epilogue only), at the SA1 / SA2 shapes of the benchmark step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

lib = _ext._lib
p = _ext._p
dev = torch.device("cuda")
for rows, n, k in ((706504, 128, 64), (706504, 64, 64), (113000, 256, 128), (113000, 128, 128),
                   (131072, 256, 128), (65536, 256, 128)):
    a = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev)
    c = torch.empty(rows, n, device=dev)
    pa, pb = torch.rand(k, device=dev), torch.rand(k, device=dev)
    part = torch.empty(lib.btr_sa_gemm_grid(rows), 2, n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for out in (c, None):
        def fn():
            assert lib.btr_sa_gemm_nt(rows, n, k, p(a), k, p(w), k, p(out), n, p(pa), p(pb), p(part), st) == 0
        med, mn = timeit(fn, iters=20, warmup=3)
        res.append(med)
    flops = 2.0 * rows * n * k
    print("rows=%7d n=%3d k=%3d  with C %.1f us (%.1f TF, %.0f GB/s)   stats only %.1f us (%.1f TF)" % (
        rows, n, k, res[0] * 1e3, flops / res[0] / 1e9, 4.0 * rows * (n + k) / res[0] / 1e6,
        res[1] * 1e3, flops / res[1] / 1e9))
