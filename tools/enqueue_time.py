#!/usr/bin/env python3
"""Host time to ENQUEUE one training step (no sync inside) vs wall time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
batch = synthetic.make_batch(0, B, N, cfg, device=dev)
for _ in range(4):
    train.train_step(net, opt, batch, cfg)
torch.cuda.synchronize()
enq = []
t_all = time.perf_counter()
for i in range(10):
    t0 = time.perf_counter()
    train.train_step(net, opt, batch, cfg)
    enq.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t_all) / 10
print("B=%d N=%d enqueue ms/step: %s | wall %.2f ms/step" % (B, N, " ".join("%.1f" % e for e in enq), wall))
