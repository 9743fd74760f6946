#!/usr/bin/env python3
"""cProfile of the host side of the Back-to-Reality step (where does enqueue time go?)."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev, domain_adaptation=True)
opt = train.make_optimizer(net)
bS = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
bT = synthetic.make_batch(1000, 8, 40000, cfg, device=dev)
for _ in range(4):
    train.train_step_br(net, opt, bS, bT, cfg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    train.train_step_br(net, opt, bS, bT, cfg)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:7000])
