#!/usr/bin/env python3
"""Experiment: the whole VoteNet FSB training step (forward, loss, backward, fused Adam) captured
once into a HIP graph (torch.cuda.CUDAGraph) and replayed, against the eager loop.
Usage: python tools/graph_step.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
train.enable_conv_autotune()
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
batch = synthetic.make_batch(0, 8, 40000, cfg, device=dev)


def make():
    net = train.build_model(cfg, dev, seed=0)
    params = list(net.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, fused=True, capturable=True)
    return net, opt


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# eager
net, opt = make()
losses_e = []
for _ in range(6):
    loss, _ = train.step(net, opt, batch, cfg) if hasattr(train, "step") else train.train_step(
        net, opt, batch, cfg)
    losses_e.append(float(loss))
train.freeze_gc()
ms_e = timed(lambda: train.train_step(net, opt, batch, cfg), steps)
print("eager : %.3f ms/step   losses %s" % (ms_e, ["%.5f" % v for v in losses_e]), flush=True)

# graph
net, opt = make()
losses_g = []
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        loss, _ = train.train_step(net, opt, batch, cfg)
        losses_g.append(float(loss))
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss, _ = train.train_step(net, opt, batch, cfg)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
    losses_g.append(float(static_loss))
ms_g = timed(g.replay, steps)
print("graph : %.3f ms/step   losses %s" % (ms_g, ["%.5f" % v for v in losses_g]), flush=True)
