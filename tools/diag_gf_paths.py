#!/usr/bin/env python3
"""Which paths a GroupFree3D training step takes: counts of fused / stock decisions."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.groupfree import train as gf_train, fused_decoder, fused_decode
from backtoreality_amd.pointnet2 import fused_mlp
from backtoreality_amd.votenet import config, synthetic

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = gf_train.build_model(cfg, dev)
opt = gf_train.make_optimizer(net)
batch = synthetic.make_batch(0, 4, 50000, cfg, use_height=False, device=dev)
cnt = collections.Counter()
for mod, name in ((fused_mlp, "run_chain"), (fused_decoder, "layer_forward"), (fused_decode, "decode")):
    real = getattr(mod, name)
    def wrap(*a, _real=real, _name=name, **k):
        out = _real(*a, **k)
        cnt[(_name, out is not None)] += 1
        if out is None and _name == "run_chain":
            x = a[0]
            cnt[("run_chain none", tuple(x.shape), fused_mlp._min_rows())] += 1
        return out
    setattr(mod, name, wrap)
for i in range(2):
    cnt.clear()
    gf_train.train_step(net, opt, batch, cfg)
    torch.cuda.synchronize()
    print(i, dict(cnt))
