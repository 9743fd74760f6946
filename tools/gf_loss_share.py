#!/usr/bin/env python3
"""How much of the GroupFree3D step is its loss?  The whole step as a HIP graph with the real
get_loss against the same graph with a trivial loss (sum of the head outputs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.groupfree import train as gf_train
from backtoreality_amd.votenet import config, synthetic
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
b = synthetic.make_batch(0, 4, 50000, cfg, use_height=False, device=dev)


def trivial(end_points, cfg, **kw):
    keys = [k for k in end_points if k.endswith("objectness_scores") or k.endswith("center")
            or k.endswith("sem_cls_scores") or k.endswith("size_residuals_normalized")
            or k.endswith("heading_scores") or k.endswith("heading_residuals_normalized")
            or k.endswith("size_scores") or k == "seeds_obj_cls_logits"]
    loss = sum(end_points[k].float().mean() for k in keys)
    end_points["loss"] = loss
    return loss, end_points


for name, crit in (("real loss", None), ("trivial loss", trivial)):
    net = gf_train.build_model(cfg, dev)
    opt = gf_train.make_optimizer(net, capturable=True)
    step_fn = gf_train.train_step
    orig = gf_train.train_step
    if crit is not None:
        gf_train.train_step = lambda n, o, bb, c, la=None, cn=0.1, **kw: orig(n, o, bb, c, la, cn,
                                                                               criterion=crit, **kw)
    gs = gf_train.GraphedTrainStep(net, opt, b, cfg)
    gf_train.train_step = orig
    for _ in range(3):
        gs(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gs(b)
    torch.cuda.synchronize()
    print("%-14s %.2f ms/step" % (name, 1e2 * (time.perf_counter() - t0)))
