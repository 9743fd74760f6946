#!/usr/bin/env python3
"""cProfile of the host side of the BACKWARD of the pipelined step (autograd multithreading off,
so that the backward runs in the profiled thread)."""
import cProfile, os, pstats, sys, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
batches = [synthetic.make_batch(s, 2, 20000, cfg, device=dev) for s in (0, 1)]


def loop(n):
    sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
    for i in range(n):
        out = train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                               next_batch=batches[(i + 1) % 2])
        sampling = out[1].get('next_sampling')


with torch.autograd.set_multithreading_enabled(False):
    loop(5)
    torch.cuda.synchronize()
    train.freeze_gc()
    pr = cProfile.Profile()
    pr.enable()
    loop(20)
    pr.disable()
    torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(60)
print(st.getvalue()[:12000])
