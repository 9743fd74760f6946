#!/usr/bin/env python3
"""GPU time of the GroupFree3D decoder stack's two library calls inside the training step
(event pairs on the caller's stream around btr_gf_stack_forward / _backward: the main lane's time
including what it waits for), and the step time, for the current settings of BTR_GRAPHS /
BTR_GF_SLOTS.  Usage: python tools/gf_stack_times.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.groupfree import fused_stack, train as gf_train  # noqa: E402
from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = gf_train.build_model(cfg, dev)
opt = gf_train.make_optimizer(net)
B, N = 4, 50000
batches = [synthetic.make_batch(s, B, N, cfg, use_height=False, device=dev) for s in (0, 1)]
pairs = {"btr_gf_stack_forward": [], "btr_gf_stack_backward": []}
orig = fused_stack._call
record = [False]


def spy(fn, *args, **kw):
    name = getattr(fn, "__name__", "")
    if record[0] and name in pairs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = orig(fn, *args, **kw)
        b.record()
        pairs[name].append((a, b))
        return r
    return orig(fn, *args, **kw)


fused_stack._call = spy


def loop(n):
    sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
    for i in range(n):
        out = gf_train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                                  next_batch=batches[(i + 1) % 2])
        sampling = out[1].get('next_sampling')


n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loop(6)
torch.cuda.synchronize()
train.freeze_gc()
t0 = time.perf_counter()
loop(n)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("BTR_GRAPHS=%s BTR_GF_SLOTS=%s: step %.3f ms (host enqueue %.3f)" % (
    os.environ.get("BTR_GRAPHS", "1"), os.environ.get("BTR_GF_SLOTS", "1"),
    1e3 * (t2 - t0) / n, 1e3 * (t1 - t0) / n))
record[0] = True
loop(n)
torch.cuda.synchronize()
for name, evs in pairs.items():
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    print("   %-24s median %.3f ms  (min %.3f, max %.3f)" % (name, ms[len(ms) // 2], ms[0], ms[-1]))
print("   graphs:", _ext.graph_stats())
