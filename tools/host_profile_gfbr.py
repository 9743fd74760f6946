#!/usr/bin/env python3
"""cProfile of the host side of the eager GroupFree3D two-branch step (train_step_br); with
GF_PLAIN=1 of the fully supervised step (train_step)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.groupfree import train as gf_train
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
PLAIN = os.environ.get("GF_PLAIN") == "1"
net = gf_train.build_model(cfg, dev, domain_adaptation=not PLAIN)
opt = gf_train.make_optimizer(net)
B, N = 4, 50000
bS = [synthetic.make_batch(s, B, N, cfg, use_height=False, device=dev) for s in (0, 1)]
bT = [synthetic.make_batch(s, B, N, cfg, use_height=False, device=dev) for s in (50, 51)]


def loop(n):
    for i in range(n):
        if PLAIN:
            gf_train.train_step(net, opt, bS[i % 2], cfg)
        else:
            gf_train.train_step_br(net, opt, bS[i % 2], bT[i % 2], cfg)


loop(5)
torch.cuda.synchronize()
train.freeze_gc()
t0 = time.perf_counter()
loop(10)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, to GPU idle %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile()
pr.enable()
loop(10)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats(key).print_stats(32)
    print(st.getvalue()[:5500])
