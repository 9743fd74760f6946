"""BR step: gradient w.r.t. every inter-layer activation, fused SA path and nine-op path on the
GPU against the SAME step in float64 on the CPU (tools/f64_truth.py's ExtF64): which layer's
backward moves the fused path away from the truth?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import test_golden_cpu as T  # noqa: E402
from f64_truth import ExtF64, to64  # noqa: E402
from backtoreality_amd.pointnet2 import _ext, pointnet2_utils  # noqa: E402
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

g = np.load(os.path.join(T.GOLD, "votenet_br_step.npz"))
cfg = config.scannet_md40()
KEYS = ('sa1_features', 'sa2_features', 'sa3_features', 'sa4_features', 'fp2_features',
        'vote_features', 'aggregated_vote_features')


def run(device, f64, fused):
    os.environ["BTR_FUSED_SA"] = fused
    os.environ["BTR_FUSED_LOSS"] = "0" if f64 else "1"
    pointnet2_utils._ext = ExtF64() if f64 else _ext
    bS = synthetic.make_batch(0, 2, 4096, cfg, device=device)
    bT = synthetic.make_batch(100, 2, 4096, cfg, device=device)
    net = train.build_model(cfg, device, domain_adaptation=True, seed=0)
    if f64:
        bS, bT, net = to64(bS), to64(bT), net.double()
    with T.pinned_vote_inds(net, g['S_aggregated_vote_inds'], g['T_aggregated_vote_inds'],
                            idx_per_forward=[g['S_vote_agg_idx'], g['T_vote_agg_idx']]):
        eS = net({'point_clouds': bS['point_clouds']})
        eT = net({'point_clouds': bT['point_clouds']})
    for e in (eS, eT):
        for k in KEYS:
            e[k].retain_grad()
    eS.update(bS)
    eT.update(bT)
    loss, eS, eT = loss_helper.get_loss_DA(eS, eT, cfg)
    loss.backward()
    out = {}
    for tag, e in (("S", eS), ("T", eT)):
        for k in KEYS:
            out[tag + " " + k] = (e[k].detach().double().cpu(), e[k].grad.double().cpu())
    for n, p in net.named_parameters():
        if n.endswith("layer0.conv.weight") and p.grad is not None:
            out["dW " + n] = (p.detach().double().cpu(), p.grad.double().cpu())
    return out


truth = run(torch.device("cpu"), True, "0")
dev = torch.device("cuda:0")
fu = run(dev, False, "1")
un = run(dev, False, "0")


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-300))


print("%-50s %-23s %-23s" % ("gradient of", "fused: max-norm / L2", "nine-op: max-norm / L2"))
for k in truth:
    print("%-50s %.1e / %.1e      %.1e / %.1e     (value: %.1e / %.1e)" % (
        k, rel(fu[k][1], truth[k][1]), l2(fu[k][1], truth[k][1]),
        rel(un[k][1], truth[k][1]), l2(un[k][1], truth[k][1]),
        rel(fu[k][0], truth[k][0]), rel(un[k][0], truth[k][0])))
