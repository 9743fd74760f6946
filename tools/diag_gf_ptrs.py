#!/usr/bin/env python3
"""Do the decoder-stack calls of consecutive GroupFree3D steps see the same device pointers?
(What a replayed HIP graph of the call needs.)  Records every argument of btr_gf_stack_forward /
_backward -- scalars, pointers, the pointer arrays' contents, the descriptor's bytes -- over the
steps of the software-pipelined loop and prints how many distinct argument tuples there were and
which positions moved."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.groupfree import fused_stack, train as gf_train  # noqa: E402
from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
BR = len(sys.argv) > 1 and sys.argv[1] == "br"   # the two-branch step (two calls in flight)
net = gf_train.build_model(cfg, dev, domain_adaptation=BR)
opt = gf_train.make_optimizer(net)
B, N = 4, 50000
batches = [synthetic.make_batch(s, B, N, cfg, use_height=False, device=dev) for s in (0, 1)]
seen = {"btr_gf_stack_forward": [], "btr_gf_stack_backward": []}
orig = fused_stack._call


def flat(a, dsize):
    if isinstance(a, ctypes.Array):
        return tuple(a)
    if isinstance(a, int) and a > (1 << 32) and dsize:
        return a
    return a


def spy(fn, *args, **kw):
    name = getattr(fn, "__name__", None) or getattr(fn, "_name", "")
    if name in seen:
        d = ctypes.string_at(args[0], ctypes.sizeof(_ext.GfStack))
        seen[name].append((d,) + tuple(flat(a, True) for a in args[2:]))
    return orig(fn, *args, **kw)


fused_stack._call = spy


def loop(n):
    if BR:   # bench.py's software-pipelined two-branch loop
        batches_T = loop.batches_T
        sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
        sampling_t = None
        for i in range(n):
            last = i + 1 >= n
            out = gf_train.train_step_br(net, opt, batches[i % 2], batches_T[i % 2], cfg,
                                         sampling_S=sampling, sampling_T=sampling_t,
                                         next_batch_S=None if last else batches[(i + 1) % 2],
                                         next_batch_T=None if last else batches_T[(i + 1) % 2])
            sampling = out[1].get('next_sampling')
            sampling_t = out[2].get('next_sampling')
        return
    sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
    for i in range(n):
        out = gf_train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                                  next_batch=batches[(i + 1) % 2])
        sampling = out[1].get('next_sampling')


loop.batches_T = [synthetic.make_batch(100000 + 7000 * i, B, N, cfg, use_height=False, device=dev)
                  for i in range(2)] if BR else None
loop(6)
torch.cuda.synchronize()
train.freeze_gc()
for k in seen:
    seen[k].clear()
loop(16)
torch.cuda.synchronize()
print("graphs:", _ext.graph_stats())
for name, calls in seen.items():
    print("%s: %d calls, %d distinct argument tuples" % (name, len(calls), len(set(calls))))
    if not calls:
        continue
    for pos in range(len(calls[0])):
        vals = [c[pos] for c in calls]
        nd = len(set(vals))
        if nd > 1:
            what = "descriptor bytes" if pos == 0 else "argument %d" % (pos + 1)
            extra = ""
            if pos == 0:
                a, b = vals[0], next(v for v in vals if v != vals[0])
                diff = [i for i in range(len(a)) if a[i] != b[i]]
                extra = " (%d bytes differ, offsets %s ...)" % (len(diff), diff[:12])
            print("   %s: %d distinct values over the calls%s; sequence of ids %s" % (
                what, nd, extra, [sorted(set(vals), key=vals.index).index(v) for v in vals]))
