#!/usr/bin/env python3
"""One Adam step (lr 5e-5) from the same weights on batch 0 through the fused and the nine-op
path: where do the updated weights differ, and which path's EVALUATION of batch 1 differs?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
batches = [synthetic.make_batch(10 * i, 2, 4096, cfg, device=dev) for i in range(4)]
KEYS = ("BTR_FUSED_SA", "BTR_FUSED_MLP", "BTR_FUSED_LOSS", "BTR_FUSED_VOTES")
LR = float(os.environ.get("TEST_LR", "5e-5"))


def setf(f):
    for k in KEYS:
        os.environ[k] = f


def one_step(f):
    setf(f)
    net = train.build_model(cfg, dev, seed=0)
    opt = train.make_optimizer(net, lr=LR)
    l0 = float(train.train_step(net, opt, batches[0], cfg)[0])
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    return net, l0, grads


def evaluate(net, f, i=1):
    setf(f)
    opt = train.make_optimizer(net, lr=0.0)
    import copy
    n2 = copy.deepcopy(net)
    return float(train.train_step(n2, train.make_optimizer(n2, lr=0.0), batches[i], cfg)[0])


net_h, l0h, gh = one_step("1")
net_n, l0n, gn = one_step("0")
print("step-0 loss fused %.4f nine %.4f" % (l0h, l0n))
print("batch 1 loss:  fused-weights/fused-eval %.4f  fused-weights/nine-eval %.4f  "
      "nine-weights/fused-eval %.4f  nine-weights/nine-eval %.4f" % (
          evaluate(net_h, "1"), evaluate(net_h, "0"), evaluate(net_n, "1"), evaluate(net_n, "0")))
rows = []
ph, pn = dict(net_h.named_parameters()), dict(net_n.named_parameters())
for n in ph:
    d = (ph[n] - pn[n]).abs()
    flips = int((d > 1.5 * LR).sum())
    g1, g2 = gh.get(n), gn.get(n)
    rel = float((g1 - g2).abs().max() / g2.abs().max().clamp_min(1e-30)) if g1 is not None else -1
    rows.append((flips / d.numel(), flips, d.numel(), rel, float(g2.abs().max()), n))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.3f flips %7d / %7d  grad rel diff %.2e  |g|max %.2e  %s" % r)
