#!/usr/bin/env python3
"""cProfile of the host side of the training step (where does enqueue time go?)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
batch = synthetic.make_batch(0, 1, 4096, cfg, device=dev)
for _ in range(4):
    train.train_step(net, opt, batch, cfg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    train.train_step(net, opt, batch, cfg)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45)
print(st.getvalue()[:9000])
