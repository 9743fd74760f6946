"""Do forward activations of the fused BR step change between the forward and the end of the
backward?  (They must not: autograd's saved tensors alias them.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T  # noqa: E402
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(T.GOLD, "votenet_br_step.npz"))
cfg = config.scannet_md40()
bS = synthetic.make_batch(0, 2, 4096, cfg, device=dev)
bT = synthetic.make_batch(100, 2, 4096, cfg, device=dev)
os.environ["BTR_FUSED_SA"] = sys.argv[1] if len(sys.argv) > 1 else "1"
net = train.build_model(cfg, dev, domain_adaptation=True, seed=0)

# record every tensor autograd saves during the two forwards (pack hook sees them all)
saved = []


def pack(t):
    if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 0:
        saved.append((t, t.detach().clone()))
    return t


with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
    with T.pinned_vote_inds(net, g['S_aggregated_vote_inds'], g['T_aggregated_vote_inds'],
                            idx_per_forward=[g['S_vote_agg_idx'], g['T_vote_agg_idx']]):
        eS = net({'point_clouds': bS['point_clouds']})
        nS = len(saved)
        eT = net({'point_clouds': bT['point_clouds']})
torch.cuda.synchronize()
print("saved tensors: S forward %d, T forward %d" % (nS, len(saved) - nS))
eS.update(bS)
eT.update(bT)
loss, eS, eT = loss_helper.get_loss_DA(eS, eT, cfg)
torch.cuda.synchronize()
changed = [(i, tuple(t.shape), str(t.dtype)) for i, (t, c) in enumerate(saved) if not torch.equal(t, c)]
print("changed by the loss:", changed[:20])
saved2 = [(t, t.detach().clone()) for t, _ in saved]
loss.backward()
torch.cuda.synchronize()
print("changed by the backward (in-place reuse by the fused backward is expected for ITS OWN "
      "saved Y tensors):")
for i, (t, c) in enumerate(saved2):
    if not torch.equal(t, c):
        print("   #%d %s %s branch %s  max|delta| %.3e" % (
            i, tuple(t.shape), t.dtype, "S" if i < nS else "T", float((t.float() - c.float()).abs().max())))
