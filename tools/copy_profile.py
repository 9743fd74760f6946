#!/usr/bin/env python3
"""Which aten::copy_ / contiguous calls of a step move the most data (shapes + source line)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
             with_stack=True) as prof:
    train.train_step(net, opt, batch, cfg)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name in ("aten::copy_", "aten::contiguous", "aten::clone") and e.device_time_total > 0:
        src = [f for f in (e.stack or []) if "/root/repo" in f or "backtoreality" in f]
        rows.append((e.device_time_total, e.name, str(e.input_shapes)[:60], (src[0] if src else "")[-70:]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total %.0f us in %d calls" % (tot, len(rows)))
for r in rows[:30]:
    print("%7.1f us  %-16s %-60s %s" % r)
