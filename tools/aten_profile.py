#!/usr/bin/env python3
"""torch.profiler view of one training step: which aten ops the non-hand-written GPU time
belongs to (kernel names alone do not say who launched an `elementwise_kernel`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
batch = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
for _ in range(4):
    train.train_step(net, opt, batch, cfg)
torch.cuda.synchronize()
steps = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(steps):
        train.train_step(net, opt, batch, cfg)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    dev_t = getattr(e, "self_device_time_total", None)
    if dev_t is None:
        dev_t = e.self_cuda_time_total
    rows.append((dev_t / steps, e.count / steps, e.self_cpu_time_total / steps, e.key))
rows.sort(reverse=True)
print("%10s %8s %10s  %s" % ("gpu us", "calls", "cpu us", "op (per step)"))
for dev_t, cnt, cpu_t, key in rows[:70]:
    print("%10.1f %8.1f %10.1f  %s" % (dev_t, cnt, cpu_t, key[:110]))
print("total self cpu us per step: %.0f" % (sum(r[2] for r in rows)))
