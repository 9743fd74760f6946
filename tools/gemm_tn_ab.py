#!/usr/bin/env python3
"""Timing of the plain weight-gradient GEMM (btr_sa_gemm_tn: TN kernel + split-K reduction) at
shapes of the benchmark steps (HIP events).  BTR_GEMM=f32 for the f32-input kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

lib = _ext._lib
SHAPES = [(114624, 128, 132, True), (114624, 128, 128, True), (65536, 128, 260, False),
          (32768, 128, 128, True), (8192, 256, 256, True), (8192, 256, 512, False),
          (2048, 128, 128, True), (1024, 2048, 288, False), (1024, 288, 2048, False),
          (4096, 576, 288, False), (1024, 864, 288, False)]


def run(rows, n, k, pro):
    dev = torch.device("cuda")
    g = torch.randn(rows, n, device=dev)
    x = torch.randn(rows, k, device=dev)
    pa = torch.rand(k, device=dev) if pro else None
    pb = torch.rand(k, device=dev) if pro else None
    chunks = lib.btr_sa_gemm_tn_chunks(rows, n, k)
    pw = torch.empty(chunks, n, k, device=dev)
    dw = torch.empty(n, k, device=dev)
    p = _ext._p
    st = torch.cuda.current_stream().cuda_stream

    def fn():
        rc = lib.btr_sa_gemm_tn(rows, n, k, p(g), n, p(x), k, p(pa), p(pb), p(pw), p(dw), st)
        assert rc == 0
    med, mn = timeit(fn, iters=20, warmup=3)
    xe = torch.relu(x * pa + pb) if pro else x
    ref = g.double().t() @ xe.double()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    return med, 2.0 * rows * n * k / med / 1e9, chunks, err


POOL_SHAPES = [(8 * 2048 * 32, 32, 128, 64), (8 * 1024 * 32, 32, 256, 128),
               (8 * 512 * 16, 16, 256, 128), (8 * 256 * 16, 16, 256, 128)]


def run_pool(rows, s, n, k):
    """btr_sa_gemm_tn_pool on plain rows (BTR_GEMM=f32: the f32-input kernel)."""
    dev = torch.device("cuda")
    y = torch.randn(rows, n, device=dev)
    x = torch.randn(rows, k, device=dev)
    arg = torch.randint(0, s, (rows // s, n), device=dev, dtype=torch.uint8)
    dcl = torch.randn(rows // s, n, device=dev)
    al, be = torch.randn(n, device=dev) * 0.1, torch.randn(n, device=dev) * 0.1
    pa, pb = torch.rand(k, device=dev), torch.rand(k, device=dev)
    chunks = lib.btr_sa_gemm_tn_chunks(rows, n, k)
    pw = torch.empty(chunks, n, k, device=dev)
    dw = torch.empty(n, k, device=dev)
    p = _ext._p
    st = torch.cuda.current_stream().cuda_stream

    def fn():
        rc = lib.btr_sa_gemm_tn_pool(rows, n, k, p(y), n, s, p(arg), p(dcl), p(al), p(be), p(x), k,
                                     p(pa), p(pb), p(pw), p(dw), st)
        assert rc == 0
    med, mn = timeit(fn, iters=20, warmup=3)
    return med, 2.0 * rows * n * k / med / 1e9, rows * (n + k) * 4 / med / 1e6


if __name__ == "__main__":
    for shp in POOL_SHAPES:
        med, tf, gbs = run_pool(*shp)
        print("gemm_tn_pool rows=%7d s=%3d n=%4d k=%4d  %7.1f us  %6.1f TF  %6.0f GB/s" % (
            shp + (med * 1e3, tf, gbs)))
    for shp in SHAPES:
        med, tf, chunks, err = run(*shp)
        print("gemm_tn rows=%7d n=%4d k=%4d pro=%d chunks=%3d  %7.1f us  %6.1f TF  err %.1e" % (
            shp + (chunks, med * 1e3, tf, err)))
