#!/usr/bin/env python3
"""cProfile of the host side of the eager, software-pipelined GroupFree3D step.  A small batch,
so that the GPU never holds the host back."""
import cProfile, os, pstats, sys, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.groupfree import train as gf_train
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = gf_train.build_model(cfg, dev)
opt = gf_train.make_optimizer(net)
B, N = int(os.environ.get("HP_B", 4)), int(os.environ.get("HP_N", 20000))
batches = [synthetic.make_batch(s, B, N, cfg, use_height=False, device=dev) for s in (0, 1)]


def loop(n):
    sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
    for i in range(n):
        out = gf_train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                                  next_batch=batches[(i + 1) % 2])
        sampling = out[1].get('next_sampling')


loop(5)
torch.cuda.synchronize()
train.freeze_gc()
t0 = time.perf_counter()
loop(20)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, to GPU idle %.2f ms/step" % ((t1 - t0) * 50, (t2 - t0) * 50))
pr = cProfile.Profile()
pr.enable()
loop(20)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats(key).print_stats(45)
    print(st.getvalue()[:9000])
