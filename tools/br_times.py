#!/usr/bin/env python3
"""Back-to-Reality step (VoteNet_DA, two forwards + get_loss_DA + one backward): wall time per
step vs host enqueue time per step (is the loop launch-bound?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev, domain_adaptation=True)
opt = train.make_optimizer(net)
bS = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
bT = synthetic.make_batch(1000, 8, 40000, cfg, device=dev)
for _ in range(5):
    train.train_step_br(net, opt, bS, bT, cfg)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    train.train_step_br(net, opt, bS, bT, cfg)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("BR step: enqueue %.2f ms/step, wall %.2f ms/step" % (1e3 * (t1 - t0) / n,
                                                            1e3 * (t2 - t0) / n))
