#!/usr/bin/env python3
"""Does the back-to-back BR loop hit the device allocator (hipMalloc is synchronous)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev, domain_adaptation=True)
opt = train.make_optimizer(net)
bS = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
bT = synthetic.make_batch(1000, 8, 40000, cfg, device=dev)
for _ in range(5):
    train.train_step_br(net, opt, bS, bT, cfg)
torch.cuda.synchronize()
def snap():
    st = torch.cuda.memory_stats()
    return {k: st.get(k, 0) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries",
                                       "reserved_bytes.all.current", "allocated_bytes.all.peak")}
a = snap()
t0 = time.perf_counter()
for _ in range(10):
    train.train_step_br(net, opt, bS, bT, cfg)
torch.cuda.synchronize()
t1 = time.perf_counter()
b = snap()
print("wall %.2f ms/step" % (1e2 * (t1 - t0)))
for k in a:
    print(k, a[k], "->", b[k])
