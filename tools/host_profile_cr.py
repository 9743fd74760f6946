#!/usr/bin/env python3
"""cProfile of the host side of the VoteNet CenterRefine step (train_step_br_jitter); with
BR_PLAIN=1 of the Back-to-Reality step without the centre branch (train_step_br)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
PLAIN = os.environ.get('BR_PLAIN') == '1'
net = train.build_model(cfg, dev, domain_adaptation=True, center_refine=not PLAIN)
opt = train.make_optimizer(net)
jit = 0.1
bS = [synthetic.make_batch(s, 8, 40000, cfg, device=dev, center_jitter=jit) for s in (0, 1)]
bT = [synthetic.make_batch(s, 8, 40000, cfg, device=dev, center_jitter=jit) for s in (1000, 1001)]


def loop(n):
    for i in range(n):
        if PLAIN:
            train.train_step_br(net, opt, bS[i % 2], bT[i % 2], cfg)
        else:
            train.train_step_br_jitter(net, opt, bS[i % 2], bT[i % 2], cfg, epoch=30)


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
    pstats.Stats(pr, stream=st).sort_stats(key).print_stats(30)
    print(st.getvalue()[:5000])
