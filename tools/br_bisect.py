"""BR step: which single fused SA layer moves grad_sa1_w0 away from the float64 truth?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T  # noqa: E402
from backtoreality_amd.pointnet2 import fused_sa  # noqa: E402
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(T.GOLD, "votenet_br_step.npz"))
truth = np.load(os.path.join(T.GOLD, "f64_truth.npz"))["br_grad_sa1_w0"]
cfg = config.scannet_md40()
bS = synthetic.make_batch(0, 2, 4096, cfg, device=dev)
bT = synthetic.make_batch(100, 2, 4096, cfg, device=dev)
real_can = fused_sa.can_fuse


def run(which, branches=("S", "T")):
    net = train.build_model(cfg, dev, domain_adaptation=True, seed=0)
    mods = {"sa1": net.backbone_net.sa1, "sa2": net.backbone_net.sa2, "sa3": net.backbone_net.sa3,
            "sa4": net.backbone_net.sa4, "vote_agg": net.pnet.vote_aggregation}
    chosen = [mods[w] for w in which]
    state = {"branch": None}
    fused_sa.can_fuse = lambda module, xyz, features: (
        any(module is c for c in chosen) and state["branch"] in branches and
        real_can(module, xyz, features))
    try:
        with T.pinned_vote_inds(net, g['S_aggregated_vote_inds'], g['T_aggregated_vote_inds'],
                                idx_per_forward=[g['S_vote_agg_idx'], g['T_vote_agg_idx']]):
            state["branch"] = "S"
            eS = net({'point_clouds': bS['point_clouds']})
            state["branch"] = "T"
            eT = net({'point_clouds': bT['point_clouds']})
        eS.update(bS)
        eT.update(bT)
        loss, eS, eT = loss_helper.get_loss_DA(eS, eT, cfg)
        loss.backward()
    finally:
        fused_sa.can_fuse = real_can
    got = net.backbone_net.sa1.mlp_module.layer0.conv.weight.grad.cpu().numpy().astype(np.float64)
    return np.abs(got - truth).max() / np.abs(truth).max()


os.environ["BTR_FUSED_SA"] = "1"
ALL = ["sa1", "sa2", "sa3", "sa4", "vote_agg"]
print("all fused                         : %.2e" % run(ALL))
print("all fused, S only                 : %.2e" % run(ALL, ("S",)))
print("all fused, T only                 : %.2e" % run(ALL, ("T",)))
os.environ["BTR_SA_CL_SHORTCUT"] = "0"
print("all fused, no channel-last reuse  : %.2e" % run(ALL))
os.environ["BTR_SA_CL_SHORTCUT"] = "1"
import itertools
for a_, b_ in itertools.combinations(ALL, 2):
    print("fused %-8s + %-8s: %.2e" % (a_, b_, run([a_, b_])))
for drop in ALL:
    print("all but %-8s: %.2e" % (drop, run([w for w in ALL if w != drop])))
