#!/usr/bin/env python3
"""FSB step: wall time per step vs host enqueue time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
b = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
for _ in range(5):
    train.train_step(net, opt, b, cfg)
torch.cuda.synchronize()
n = 10
for mode in ("back-to-back", "synced"):
    t_enq = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        s0 = time.perf_counter()
        train.train_step(net, opt, b, cfg)
        t_enq += time.perf_counter() - s0
        if mode == "synced":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("FSB %s: enqueue %.2f ms/step, wall %.2f ms/step" % (mode, 1e3 * t_enq / n,
                                                               1e3 * (t2 - t0) / n))
