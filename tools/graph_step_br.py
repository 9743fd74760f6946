#!/usr/bin/env python3
"""Back-to-Reality VoteNet step: eager loop vs the whole step replayed as one HIP graph."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

train.enable_conv_autotune()
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
bS = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
bT = synthetic.make_batch(100000, 8, 40000, cfg, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


net = train.build_model(cfg, dev, domain_adaptation=True)
opt = train.make_optimizer(net)
for _ in range(5):
    train.train_step_br(net, opt, bS, bT, cfg)
train.freeze_gc()
print("eager: %.3f ms/step" % timed(lambda: train.train_step_br(net, opt, bS, bT, cfg)), flush=True)

net = train.build_model(cfg, dev, domain_adaptation=True)
opt = torch.optim.Adam(list(net.parameters()), lr=1e-3, fused=True, capturable=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        train.train_step_br(net, opt, bS, bT, cfg)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss, _, _ = train.train_step_br(net, opt, bS, bT, cfg)
for _ in range(3):
    g.replay()
print("graph: %.3f ms/step (loss %.4f)" % (timed(g.replay), float(loss)), flush=True)
