#!/usr/bin/env python3
"""GroupFree3D's evaluation pass (groupfree/train.evaluate_one_epoch, train_GF_FSB.py:354-445:
eval-mode forward, loss statistics, parse_predictions of all eight prediction heads, AP at IoU
0.25 and 0.5) at BASELINE config[3]'s batch shape: 4 scenes x 50 000 points, 256 queries.
Usage: python tools/gf_eval_times.py [batches]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.groupfree import train as gf_train  # noqa: E402
from backtoreality_amd.votenet import config, synthetic  # noqa: E402

if os.environ.get("GF_EVAL_WALK") == "1":
    # the pass as it was before the lists carried their arrays: plain lists (eval_det walks the
    # tuples), one calculator per threshold
    from backtoreality_amd.votenet import ap_helper
    _parse = ap_helper.parse_predictions
    ap_helper.parse_predictions = lambda *a, **k: [list(x) for x in _parse(*a, **k)]
    _metrics = ap_helper.APCalculator.compute_metrics
    ap_helper.APCalculator.compute_metrics = lambda self, thr=None: (
        {t: _metrics(self, t) for t in thr} if isinstance(thr, (list, tuple)) else _metrics(self, thr))
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = gf_train.build_model(cfg, dev)
B, N = 4, 50000
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [synthetic.make_batch(1000 * i, B, N, cfg, use_height=False, device=dev)
           for i in range(nb)]
opt = gf_train.make_optimizer(net)
for b in batches[:2]:
    gf_train.train_step(net, opt, b, cfg)
gf_train.evaluate_one_epoch(net, batches[:2], cfg)
torch.cuda.synchronize()
with torch.no_grad():
    net.eval()
    t0 = time.perf_counter()
    for b in batches:
        net({'point_clouds': b['point_clouds']})
    torch.cuda.synchronize()
    print("forward only: %.2f ms per batch" % ((time.perf_counter() - t0) / nb * 1e3))
    net.train()
t0 = time.perf_counter()
gf_train.evaluate_one_epoch(net, batches, cfg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / nb
print("evaluate_one_epoch: %.2f ms per batch = %.0f scenes/s" % (dt * 1e3, B / dt))
pr = cProfile.Profile()
pr.enable()
gf_train.evaluate_one_epoch(net, batches, cfg)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
