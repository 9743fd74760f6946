"""Scene-100 batch through the BR step, fused vs nine-op: per-channel view of the FP2 MLP
(torch ops in both runs): which channels' gradients differ, and are they degenerate?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
b0 = synthetic.make_batch(0, 2, 4096, cfg, device=dev)
b1 = synthetic.make_batch(100, 2, 4096, cfg, device=dev)


def run(fused, pins=None):
    os.environ["BTR_FUSED_SA"] = fused
    net = train.build_model(cfg, dev, domain_adaptation=True, seed=0)
    sa = net.pnet.vote_aggregation
    own = sa.forward
    rec = []

    def fwd(xyz, features=None, inds=None):
        if pins is not None:
            inds = pins[len(rec)]
        out = own(xyz, features, inds)
        rec.append(out[2])
        return out
    sa.forward = fwd
    cap = {}

    def hook(name):
        def h(mod, inp, out):
            if name in cap:      # second forward (scene 100) only
                out.retain_grad()
                cap[name + "_T"] = out
                inp[0].retain_grad()
                cap[name + "_T_in"] = inp[0]
            else:
                cap[name] = out
        return h
    mods = {"fp2_l0_conv": net.backbone_net.fp2.mlp.layer0.conv,
            "fp2_l0_bn": net.backbone_net.fp2.mlp.layer0.bn.bn,
            "fp2_l1_conv": net.backbone_net.fp2.mlp.layer1.conv,
            "fp2_l1_bn": net.backbone_net.fp2.mlp.layer1.bn.bn}
    for n, m in mods.items():
        m.register_forward_hook(hook(n))
    e1 = net({'point_clouds': b0['point_clouds']})
    e2 = net({'point_clouds': b1['point_clouds']})
    e1.update(b0)
    e2.update(b1)
    loss, e1, e2 = loss_helper.get_loss_DA(e1, e2, cfg)
    loss.backward()
    return {k: (v.detach(), v.grad) for k, v in cap.items() if k.endswith("_T") or k.endswith("_T_in")}, rec


un, pins = run("0")
fu, _ = run("1", pins)
for k in un:
    va, ga = fu[k]
    vb, gb = un[k]
    C = vb.shape[1]
    dv = (va - vb).abs().amax(dim=(0, 2, 3)) / vb.abs().amax()
    dg = (ga - gb).abs().amax(dim=(0, 2, 3)) / gb.abs().amax()
    var = vb.var(dim=(0, 2, 3))
    top = torch.topk(dg, 5)
    print("%-16s value dev max %.1e | grad dev max %.1e (L2 %.1e) | worst channels %s dev %s their var %s  (median var %.2e, min var %.2e)" % (
        k, float(dv.max()), float(dg.max()), float((ga - gb).norm() / gb.norm()), top.indices.tolist(),
        ["%.1e" % v for v in top.values.tolist()], ["%.1e" % float(var[i]) for i in top.indices.tolist()],
        float(var.median()), float(var.min())))


# ---- dump raw per-channel arrays for offline analysis
import numpy as np
out = {}
for name, r in (("un", un), ("fu", fu)):
    x, dx = r["fp2_l0_bn_T_in"]          # conv output (pre-BN) and its gradient
    y, dy = r["fp2_l0_bn_T"]             # post-(in-place)ReLU value and its gradient
    for c in (43, 60, 0):
        out["%s_x_%d" % (name, c)] = x[:, c].cpu().numpy()
        out["%s_dx_%d" % (name, c)] = dx[:, c].cpu().numpy()
        out["%s_y_%d" % (name, c)] = y[:, c].cpu().numpy()
        out["%s_dy_%d" % (name, c)] = dy[:, c].cpu().numpy()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fp2_ch.npz"), **out)
print("dumped")
