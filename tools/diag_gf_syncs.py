#!/usr/bin/env python3
"""Host-side synchronisations inside the eager, software-pipelined GroupFree3D step
(torch.cuda.set_sync_debug_mode('warn') around two steady-state steps), and the host time line
of one step: when the forward / loss / backward / optimizer calls return."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.groupfree import train as gf_train
from backtoreality_amd.votenet import config, synthetic, train

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = gf_train.build_model(cfg, dev)
opt = gf_train.make_optimizer(net)
batches = [synthetic.make_batch(s, 4, 50000, cfg, use_height=False, device=dev) for s in (0, 1)]


STEP = [0]


def loop(n, sampling):
    for _ in range(n):
        i = STEP[0]
        STEP[0] += 1
        out = gf_train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                                  next_batch=batches[(i + 1) % 2])
        sampling = out[1].get('next_sampling')
    return sampling


s = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
s = loop(6, s)
torch.cuda.synchronize()
train.freeze_gc()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    s = loop(2, s)
torch.cuda.set_sync_debug_mode("default")
print("sync warnings in 2 steps:", len(w))
for x in w[:10]:
    print("  ", str(x.message)[:200], x.filename, x.lineno)
torch.cuda.synchronize()
for n in (10, 10):
    t0 = time.perf_counter()
    s = loop(n, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
# how far ahead of the GPU the host runs: events at the end of each step
evs = []
t0 = time.perf_counter()
host = []
for i in range(10):
    s = loop(1, s)
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host return (ms):", [round(h * 1e3, 1) for h in host])
print("gpu step ends (ms, rel. first):", [round(evs[0].elapsed_time(e), 1) for e in evs])
