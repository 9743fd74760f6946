#!/usr/bin/env python3
"""Stand-alone timing of the fused attention core at the decoder's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.groupfree import fused_attention as fa
dev = torch.device("cuda:0")
B, E, H = 4, 288, 8


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, Lq, Lk in (("self", 256, 256), ("cross", 256, 1024)):
    for p in (0.0, 0.1):
        if Lq == Lk:
            q = torch.randn(Lq, B, 3 * E, device=dev, requires_grad=True)
            kv = None
        else:
            q = torch.randn(Lq, B, E, device=dev, requires_grad=True)
            kv = torch.randn(Lk, B, 2 * E, device=dev, requires_grad=True)
        w = torch.randn(Lq, B, E, device=dev)
        fwd = lambda: fa._AttentionCore.apply(q, kv, H, p, 5)
        out = fwd()
        def bwd():
            q.grad = None
            torch.autograd.grad((out * w).sum(), [q] + ([kv] if kv is not None else []),
                                retain_graph=True)
        print("%-5s p=%.1f  fwd %.1f us   bwd(2 kernels + glue) %.1f us" % (
            name, p, timeit(fwd), timeit(bwd)))
