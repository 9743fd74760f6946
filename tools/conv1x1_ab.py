#!/usr/bin/env python3
"""1x1 convolution on (B, C, N): F.conv1d (MIOpen) vs torch.matmul(W, x) (batched GEMM with a
broadcast weight), forward + backward, at the shapes of the FP / voting / proposal layers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tools.bench_ops import timeit  # noqa: E402

dev = torch.device("cuda")
for (B, cin, cout, N) in ((8, 256, 256, 1024), (8, 256, 259, 1024), (8, 128, 128, 256),
                          (8, 128, 119, 256), (8, 512, 256, 512), (8, 512, 256, 1024)):
    x = torch.randn(B, cin, N, device=dev, requires_grad=True)
    w = torch.randn(cout, cin, 1, device=dev, requires_grad=True)
    b = torch.randn(cout, device=dev, requires_grad=True)
    g = torch.randn(B, cout, N, device=dev)

    def conv():
        y = F.conv1d(x, w, b)
        y.backward(g)
        x.grad = w.grad = b.grad = None

    def mm():
        y = torch.matmul(w[:, :, 0], x) + b[:, None]
        y.backward(g)
        x.grad = w.grad = b.grad = None

    def lin():
        y = F.linear(x.transpose(1, 2), w[:, :, 0], b).transpose(1, 2)
        y.backward(g)
        x.grad = w.grad = b.grad = None

    res = [timeit(f, iters=20, warmup=3)[0] for f in (conv, mm, lin)]
    print("B=%d cin=%3d cout=%3d N=%4d  conv1d %.3f ms  matmul %.3f ms  linear(T) %.3f ms (fwd+bwd)"
          % ((B, cin, cout, N) + tuple(res)))
