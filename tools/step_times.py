#!/usr/bin/env python3
"""Per-step wall times of the VoteNet training step (synchronised after every step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
batch = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
ts = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train.train_step(net, opt, batch, cfg)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join("%.1f" % t for t in ts))
if "--pipelined" in sys.argv:   # opt-in cross-step sampling prefetch (not used by bench.py)
    sampling = None
    for _ in range(3):
        _, end = train.train_step(net, opt, batch, cfg, sampling=sampling, next_batch=batch)
        sampling = end['next_sampling']
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        _, end = train.train_step(net, opt, batch, cfg, sampling=sampling, next_batch=batch)
        sampling = end['next_sampling']
    torch.cuda.synchronize()
    print("pipelined (next batch's sampling under this step's backward): %.2f ms/step" %
          (1e3 * (time.perf_counter() - t0) / n))
print("mem GB: alloc %.2f reserved %.2f" % (torch.cuda.max_memory_allocated() / 2**30, torch.cuda.max_memory_reserved() / 2**30))
