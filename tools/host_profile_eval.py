#!/usr/bin/env python3
"""Where the host time of the scripts' evaluation pass goes (train.evaluate_one_epoch at
8 scenes x 40 000 points): cProfile of the pass, top entries by cumulative time.
Usage: python tools/host_profile_eval.py [batches]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
B, N = 8, 40000
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [synthetic.make_batch(1000 * i, B, N, cfg, device=dev) for i in range(nb)]
opt = train.make_optimizer(net)
for b in batches[:2]:
    train.train_step(net, opt, b, cfg)
train.evaluate_one_epoch(net, batches[:2], cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
train.evaluate_one_epoch(net, batches, cfg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / nb
print("evaluate_one_epoch: %.2f ms per batch = %.0f scenes/s" % (dt * 1e3, B / dt))
pr = cProfile.Profile()
pr.enable()
train.evaluate_one_epoch(net, batches, cfg)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
