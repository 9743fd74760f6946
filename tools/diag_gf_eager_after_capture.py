#!/usr/bin/env python3
"""Does capturing the GroupFree3D step as a HIP graph leave the EAGER loop slower in the same
process?  Eager steps before the capture, the capture, eager steps after it; with the path
counters of the chains and of the decoder stack."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.groupfree import train as gf_train, fused_stack
from backtoreality_amd.pointnet2 import fused_mlp
from backtoreality_amd.votenet import config, synthetic, train

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = gf_train.build_model(cfg, dev)
opt_cap = gf_train.make_optimizer(net, capturable=True)
opt = gf_train.make_optimizer(net)
batches = [synthetic.make_batch(s, 4, 50000, cfg, use_height=False, device=dev) for s in (0, 500000)]


def loop(n):
    sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
    for i in range(n):
        out = gf_train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                                  next_batch=batches[(i + 1) % 2] if i + 1 < n else None)
        sampling = out[1].get('next_sampling')


def timed(tag):
    loop(6)
    torch.cuda.synchronize()
    for k in fused_mlp.PATHS:
        fused_mlp.PATHS[k] = 0
    fused_stack.CALLS[0] = 0
    fused_stack.REFUSED.clear()
    t0 = time.perf_counter()
    loop(10)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.2f ms/step, to idle %.2f ms/step; chains %s, stack %d refused %s" % (
        tag, (t1 - t0) * 100, (t2 - t0) * 100, dict(fused_mlp.PATHS), fused_stack.CALLS[0],
        dict(fused_stack.REFUSED)), flush=True)


if os.environ.get("DIAG_NO_CAPTURE"):
    timed("no capture in this process")
    sys.exit(0)
g = gf_train.GraphedPipelinedStep(net, opt_cap, batches[0], batches[1], cfg)
torch.cuda.synchronize()
timed("after capture ")
g.prime(batches[0])
for i in range(6):
    g(batches[i % 2], batches[(i + 1) % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    g(batches[i % 2], batches[(i + 1) % 2])
torch.cuda.synchronize()
print("graph replay: %.2f ms/step" % ((time.perf_counter() - t0) * 100))
timed("after replays ")
