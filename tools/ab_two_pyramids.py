#!/usr/bin/env python3
"""One or two sampling pyramids in flight (software pipelining depth 1 / 2) for a workload whose
pyramid is the critical path of the pipelined loop: step i uses the pyramid issued at step i - d
and issues the one of batch i + d on prefetch stream (i % d).
Usage: ab_two_pyramids.py [c5|fsb] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
if wl == "c5":
    cfg, B, N, ext = config.matterport_md40(), 4, 80000, 1.7
else:
    cfg, B, N, ext = config.scannet_md40(), 8, 40000, 1.0
net = train.build_model(cfg, dev, seed=0)
opt = train.make_optimizer(net)
batches = [synthetic.make_batch(100 * s, B, N, cfg, extent_scale=ext, device=dev) for s in range(4)]
core = net


def loop(n, depth):
    handles = {}
    for j in range(depth):          # the head of the loop: the first `depth` pyramids
        handles[j] = core.backbone_net.prefetch_sampling(batches[j % 4]['point_clouds'], slot=j % depth)
    for i in range(n):
        h = handles.pop(i)
        if i + depth < n + depth:   # (keeps the pipeline full to the end: same work per step)
            handles[i + depth] = core.backbone_net.prefetch_sampling(
                batches[(i + depth) % 4]['point_clouds'], slot=(i + depth) % depth)
        train.train_step(net, opt, batches[i % 4], cfg, sampling=h)


for depth in (1, 2, 1, 2):
    loop(5, depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps, depth)
    torch.cuda.synchronize()
    print("%s pyramids in flight %d: %.3f ms/step" % (wl, depth, 1e3 * (time.perf_counter() - t0) / steps))
