#!/usr/bin/env python3
"""GroupFree3D step: wall vs host enqueue time per step, and where the GPU / host time goes
(torch profiler, top ops).  Usage: python tools/gf_times.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from backtoreality_amd.groupfree import train as gf_train  # noqa: E402
from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = gf_train.build_model(cfg, dev)
opt = gf_train.make_optimizer(net)
b = synthetic.make_batch(0, 4, 50000, cfg, use_height=False, device=dev)
for _ in range(5):
    gf_train.train_step(net, opt, b, cfg)
torch.cuda.synchronize()
train.freeze_gc()
n = 10
for mode in ("back-to-back", "synced"):
    t_enq = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        s0 = time.perf_counter()
        gf_train.train_step(net, opt, b, cfg)
        t_enq += time.perf_counter() - s0
        if mode == "synced":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("GF %s: enqueue %.2f ms/step, wall %.2f ms/step" % (mode, 1e3 * t_enq / n,
                                                              1e3 * (t2 - t0) / n), flush=True)
net2 = gf_train.build_model(cfg, dev)
opt2 = gf_train.make_optimizer(net2, capturable=True)
gs = gf_train.GraphedTrainStep(net2, opt2, b, cfg)
for _ in range(3):
    gs()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    gs(b)
torch.cuda.synchronize()
print("GF HIP graph replay: %.2f ms/step, loss %.4f" % (1e3 * (time.perf_counter() - t0) / n,
                                                       float(gs.loss)), flush=True)
if "--profile" not in sys.argv:
    sys.exit(0)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    gf_train.train_step(net, opt, b, cfg)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=20, max_name_column_width=60))
