#!/usr/bin/env python3
"""Timing of btr_pm_gemm_nt at the few-row shapes of the GroupFree3D decoder / prediction heads and
the VoteNet vote / proposal layers (HIP events).  BTR_GEMM_DEEP=0: one k chunk in flight instead
of three."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

lib = _ext._lib
SHAPES = [(1024, 288, 288, 0), (1024, 864, 288, 0), (1024, 2048, 288, 0), (1024, 288, 2048, 0),
          (4096, 576, 288, 0), (1024, 288, 288, 1), (1024, 116, 288, 2), (4096, 288, 288, 1),
          (8192, 256, 256, 1), (8192, 128, 128, 1), (2048, 288, 288, 0), (1024, 288, 8, 0)]


def run(rows, n, k, mode):
    """mode 0: plain, 1: BatchNorm + ReLU prologue and statistics, 2: bias."""
    dev = torch.device("cuda")
    a = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev)
    c = torch.empty(rows, n, device=dev)
    pa = torch.rand(k, device=dev) + 0.5 if mode == 1 else None
    pb = torch.rand(k, device=dev) - 0.5 if mode == 1 else None
    part = torch.empty(lib.btr_pm_gemm_grid(rows), 2, n, device=dev) if mode == 1 else None
    bias = torch.randn(n, device=dev) if mode == 2 else None
    p = _ext._p
    st = torch.cuda.current_stream().cuda_stream

    def fn():
        rc = lib.btr_pm_gemm_nt(rows, n, k, p(a), k, p(w), k, p(c), n, p(pa), p(pb), p(part),
                                p(bias), st)
        assert rc == 0
    med, mn = timeit(fn, iters=30, warmup=5)
    ae = torch.relu(a.double() * pa.double() + pb.double()) if mode == 1 else a.double()
    ref = ae @ w.double().t() + (bias.double() if mode == 2 else 0.0)
    err = float((c.double() - ref).abs().max() / ref.abs().max())
    return med, 2.0 * rows * n * k / med / 1e9, err


if __name__ == "__main__":
    for shp in SHAPES:
        med, tf, err = run(*shp)
        print("pm_gemm_nt rows=%5d n=%4d k=%4d mode=%d  %6.1f us  %6.1f TF  err %.1e" % (
            shp + (med * 1e3, tf, err)))
