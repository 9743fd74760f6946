#!/usr/bin/env python3
"""Host side of the eager GroupFree3D Back-to-Reality step: synchronisation points
(torch.cuda.set_sync_debug_mode('warn')) and when each phase's calls return."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.groupfree import train as gf_train, fused_attention
from backtoreality_amd.groupfree.loss_helper import get_loss_DA
from backtoreality_amd.votenet import config, synthetic, train

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = gf_train.build_model(cfg, dev, domain_adaptation=True)
opt = gf_train.make_optimizer(net)
B, N = int(os.environ.get("HP_B", 4)), int(os.environ.get("HP_N", 50000))
bS = synthetic.make_batch(0, B, N, cfg, use_height=False, device=dev)
bT = synthetic.make_batch(50, B, N, cfg, use_height=False, device=dev)
for _ in range(4):
    gf_train.train_step_br(net, opt, bS, bT, cfg)
torch.cuda.synchronize()
train.freeze_gc()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    gf_train.train_step_br(net, opt, bS, bT, cfg)
torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one step:", len(w))
for x in w[:20]:
    print("  ", x.filename.replace(ROOT, ""), x.lineno, str(x.message)[:90])
torch.cuda.synchronize()
marks = []
for it in range(3):
    t = [time.perf_counter()]
    fused_attention.bump_step(dev)
    eS = net({'point_clouds': bS['point_clouds']}); t.append(time.perf_counter())
    eT = net({'point_clouds': bT['point_clouds']}); t.append(time.perf_counter())
    eS.update(bS); eT.update(bT)
    loss, eS, eT = get_loss_DA(eS, eT, cfg, **gf_train.LOSS_ARGS); t.append(time.perf_counter())
    gf_train._zero_grad(net, opt)
    gf_train.backward(loss); t.append(time.perf_counter())
    gf_train.clip_and_step(net, opt, 0.1); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    marks.append([1e3 * (b - a) for a, b in zip(t, t[1:])])
for m in marks:
    print("fwd S %.2f  fwd T %.2f  loss %.2f  backward %.2f  step %.2f  drain %.2f ms" % tuple(m))
