"""Uninitialised-read hunt: torch.empty() hands back freed blocks of the caching allocator, so
fill the free pool with NaN before each phase of a fused BR step; any kernel that reads memory
it (or its producer) never wrote then turns a gradient into NaN."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T  # noqa: E402
from backtoreality_amd.pointnet2 import _ext, fused_sa  # noqa: E402
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(T.GOLD, "votenet_br_step.npz"))
cfg = config.scannet_md40()
bS = synthetic.make_batch(0, 2, 4096, cfg, device=dev)
bT = synthetic.make_batch(100, 2, 4096, cfg, device=dev)


def poison(val=float("nan")):
    torch.cuda.synchronize()
    blocks = [torch.full((n,), val, device=dev) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20)
              for _ in range(4)]
    small = [torch.full((n,), val, device=dev) for n in (1 << 16, 1 << 12, 1 << 8, 64) for _ in range(64)]
    torch.cuda.synchronize()
    del blocks, small


# wrap every _call so that poisoning also happens between kernels of one backward
real_call = _ext._call
os.environ["BTR_FUSED_SA"] = "1"
net = train.build_model(cfg, dev, domain_adaptation=True, seed=0)
poison()
with T.pinned_vote_inds(net, g['S_aggregated_vote_inds'], g['T_aggregated_vote_inds'],
                        idx_per_forward=[g['S_vote_agg_idx'], g['T_vote_agg_idx']]):
    eS = net({'point_clouds': bS['point_clouds']})
    poison()
    eT = net({'point_clouds': bT['point_clouds']})
eS.update(bS)
eT.update(bT)
poison()
loss, eS, eT = loss_helper.get_loss_DA(eS, eT, cfg)
poison()
loss.backward()
torch.cuda.synchronize()
bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print("loss", float(loss), "non-finite gradients:", len(bad))
for n in bad[:40]:
    print("   ", n)
truth = np.load(os.path.join(T.GOLD, "f64_truth.npz"))["br_grad_sa1_w0"]
got = net.backbone_net.sa1.mlp_module.layer0.conv.weight.grad.cpu().numpy().astype(np.float64)
print("grad_sa1_w0 vs truth with NaN-poisoned pool: %.3e" % (np.abs(got - truth).max() / np.abs(truth).max()))
