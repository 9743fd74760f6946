#!/usr/bin/env python3
"""Is the loss of step 2 a function of (weights, batch) only?  Fresh-net loss on each batch vs
the loss the same batch gets as the SECOND step of a run with lr = 0 (weights do not move)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
batches = [synthetic.make_batch(10 * i, 2, 4096, cfg, device=dev) for i in range(4)]


def fresh(i):
    net = train.build_model(cfg, dev, seed=0)
    opt = train.make_optimizer(net, lr=0.0)
    return float(train.train_step(net, opt, batches[i], cfg)[0])


def second(i, first=0):
    net = train.build_model(cfg, dev, seed=0)
    opt = train.make_optimizer(net, lr=0.0)
    train.train_step(net, opt, batches[first], cfg)
    return float(train.train_step(net, opt, batches[i], cfg)[0])


for fused in ("1", "0"):
    for k in ("BTR_FUSED_SA", "BTR_FUSED_MLP", "BTR_FUSED_LOSS", "BTR_FUSED_VOTES"):
        os.environ[k] = fused
    print("fused", fused, "fresh:", [round(fresh(i), 4) for i in range(4)])
    print("fused", fused, "as 2nd step (lr 0):", [round(second(i), 4) for i in range(4)])
    print("fused", fused, "as 2nd step again :", [round(second(i), 4) for i in range(4)])
