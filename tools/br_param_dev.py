"""Per-parameter gradient deviation, fused SA path vs nine-op path, BR and CR steps (GPU)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T  # noqa: E402

dev = torch.device("cuda:0")
for name, run in (("br", T.run_votenet_br), ("cr", T.run_votenet_br_jitter)):
    gr = {}
    for fused in ("1", "0"):
        os.environ["BTR_FUSED_SA"] = fused
        net = run(dev, pin=True)[0]
        gr[fused] = {n: p.grad.detach().clone() for n, p in net.named_parameters()
                     if p.grad is not None}
    print("==", name)
    for n in gr["0"]:
        a, b = gr["1"][n], gr["0"][n]
        l2 = float((a - b).norm() / (b.norm() + 1e-30))
        if l2 > 3e-3 and float(b.abs().max()) > 1e-6:
            print("  %-60s rel L2 %.3e   |g| %.3e" % (n, l2, float(b.norm())))
