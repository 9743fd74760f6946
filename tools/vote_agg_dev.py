"""Isolates the vote-aggregation layer of the BR step's TARGET branch: same inputs, weights,
pinned proposals and upstream gradient through (a) the fused HIP path, (b) the nine-op path,
(c) float64 on the CPU; prints where (a) and (b) leave (c)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import test_golden_cpu as T  # noqa: E402
from f64_truth import ExtF64  # noqa: E402
from backtoreality_amd.pointnet2 import _ext, pointnet2_utils  # noqa: E402
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(T.GOLD, "votenet_br_step.npz"))
cfg = config.scannet_md40()

# ---- capture the layer's inputs / upstream gradients from a nine-op BR step
os.environ["BTR_FUSED_SA"] = "0"
bS = synthetic.make_batch(0, 2, 4096, cfg, device=dev)
bT = synthetic.make_batch(100, 2, 4096, cfg, device=dev)
net = train.build_model(cfg, dev, domain_adaptation=True, seed=0)
with T.pinned_vote_inds(net, g['S_aggregated_vote_inds'], g['T_aggregated_vote_inds']):
    eS = net({'point_clouds': bS['point_clouds']})
    eT = net({'point_clouds': bT['point_clouds']})
for e in (eS, eT):
    for k in ('vote_xyz', 'vote_features', 'aggregated_vote_xyz', 'aggregated_vote_features'):
        e[k].retain_grad()
eS.update(bS)
eT.update(bT)
loss, eS, eT = loss_helper.get_loss_DA(eS, eT, cfg)
loss.backward()
sa = net.pnet.vote_aggregation
state = {k: v.detach().clone() for k, v in sa.state_dict().items()}

for tag, e in (("T", eT), ("S", eS)):
    xyz0 = e['vote_xyz'].detach().clone()
    f0 = e['vote_features'].detach().clone()
    inds = torch.as_tensor(g[tag + '_aggregated_vote_inds'], dtype=torch.int32, device=dev)
    d_feat_out = e['aggregated_vote_features'].grad.clone()
    d_xyz_out = e['aggregated_vote_xyz'].grad.clone()

    def run(mode):
        import copy
        layer = copy.deepcopy(sa)
        layer.load_state_dict(state)
        layer.train()
        if mode == "f64":
            layer = layer.double().cpu()
            pointnet2_utils._ext = ExtF64()
            x = xyz0.double().cpu().requires_grad_(True)
            f = f0.double().cpu().requires_grad_(True)
            ii, go, gx = inds.cpu(), d_feat_out.double().cpu(), d_xyz_out.double().cpu()
        else:
            os.environ["BTR_FUSED_SA"] = "1" if mode == "fused" else "0"
            pointnet2_utils._ext = _ext
            x = xyz0.clone().requires_grad_(True)
            f = f0.clone().requires_grad_(True)
            ii, go, gx = inds, d_feat_out, d_xyz_out
        nx, nf, _ = layer(x, f, ii)
        torch.autograd.backward([nf, nx], [go, gx])
        w = layer.mlp_module.layer0.conv.weight.grad
        return [t.detach().double().cpu() for t in (nf, f.grad, x.grad, w)]

    truth = run("f64")
    print("== %s branch   |dout|max %.3e, nonzero proposals %d / %d" % (
        tag, float(d_feat_out.abs().max()),
        int((d_feat_out.abs().amax(1) > 1e-3 * d_feat_out.abs().max()).sum()), d_feat_out.shape[0] * 256))
    for mode in ("fused", "nine-op"):
        got = run(mode)
        for name, a, b in zip(("out", "dfeat", "dxyz", "dW0"), got, truth):
            err = (a - b).abs()
            rel = float(err.max() / b.abs().max())
            l2 = float((a - b).norm() / b.norm())
            extra = ""
            if name == "dfeat":   # which points carry the deviation?
                per_pt = err.amax(1)           # (B, N)
                top = torch.topk(per_pt.flatten(), 5)
                extra = "  top points %s err %s" % (top.indices.tolist(),
                                                    ["%.1e" % v for v in (top.values / b.abs().max()).tolist()])
            print("   %-8s %-6s max-norm %.2e  rel L2 %.2e%s" % (mode, name, rel, l2, extra))
