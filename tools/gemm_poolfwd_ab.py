#!/usr/bin/env python3
"""What the pooling epilogue of the last layer's GEMM costs: btr_sa_gemm_nt (BN+ReLU prologue,
statistics) against btr_sa_gemm_nt_poolfwd (the same + per-group extrema) at SA1 / SA2 shapes,
dense rows, each alone on the chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.pointnet2 import _ext
from tools.bench_ops import timeit
lib = _ext._lib
dev = torch.device("cuda")
p = _ext._p
for rows, n, k, s in ((706560, 128, 64, 64), (706560, 128, 64, 16), (114688, 256, 128, 32),
                      (114688, 256, 128, 16)):
    a = torch.randn(rows, k, device=dev); w = torch.randn(n, k, device=dev)
    c = torch.empty(rows, n, device=dev)
    pa = torch.rand(k, device=dev); pb = torch.rand(k, device=dev)
    part = torch.empty(lib.btr_sa_gemm_grid(rows), 2, n, device=dev)
    gamma = torch.randn(n, device=dev)
    gext = torch.empty(rows // s, n, device=dev)
    aext = torch.empty(rows // s, n, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def plain():
        assert lib.btr_sa_gemm_nt(rows, n, k, p(a), k, p(w), k, p(c), n, p(pa), p(pb), p(part), st) == 0

    def pool():
        assert lib.btr_sa_gemm_nt_poolfwd(rows, n, k, p(a), k, p(w), k, p(c), n, p(pa), p(pb),
                                          p(part), s, p(gamma), p(gext), p(aext), st) == 0
    assert lib.btr_sa_gemm_nt_poolfwd_supported(rows, n, s)
    t0, _ = timeit(plain, iters=20, warmup=3)
    t1, _ = timeit(pool, iters=20, warmup=3)
    nbytes = 4.0 * (rows * (n + k))
    print("rows %7d n %3d k %3d s %2d: plain %6.1f us (%4.0f GB/s)  with pooling epilogue %6.1f us (%4.0f GB/s)"
          % (rows, n, k, s, t0 * 1e3, nbytes / t0 / 1e6, t1 * 1e3, nbytes / t1 / 1e6))
