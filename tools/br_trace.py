"""Scene-100 batch as the 2nd forward of a BR step: gradient of EVERY module output, fused SA
path vs nine-op path, in forward order -- where does the relative-L2 deviation appear first
(reading from the bottom = backward order)?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from backtoreality_amd.pointnet2 import fused_sa  # noqa: E402
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
b0 = synthetic.make_batch(0, 2, 4096, cfg, device=dev)
b1 = synthetic.make_batch(100, 2, 4096, cfg, device=dev)
real_can = fused_sa.can_fuse


def run(fuse_names, pins=None):
    os.environ["BTR_FUSED_SA"] = "1"
    net = train.build_model(cfg, dev, domain_adaptation=True, seed=0)
    mods = {"sa1": net.backbone_net.sa1, "sa2": net.backbone_net.sa2, "sa3": net.backbone_net.sa3,
            "sa4": net.backbone_net.sa4, "vote_agg": net.pnet.vote_aggregation}
    chosen = [mods[n] for n in fuse_names]
    fused_sa.can_fuse = lambda m, x, f: any(m is c for c in chosen) and real_can(m, x, f)
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
    cap, count = [], {}

    def hook(name):
        def h(mod, inp, out):
            count[name] = count.get(name, 0) + 1
            if count[name] != 2:
                return
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for i, o in enumerate(outs):
                if isinstance(o, torch.Tensor) and o.requires_grad and o.is_floating_point():
                    o.retain_grad()
                    cap.append(("%s[%d]" % (name, i), o))
        return h
    for n, m in net.named_modules():
        if n and (len(list(m.children())) == 0 or n in ("backbone_net.sa1", "backbone_net.sa2",
                  "backbone_net.sa3", "backbone_net.sa4", "pnet.vote_aggregation",
                  "backbone_net.fp1", "backbone_net.fp2", "vgen")):
            m.register_forward_hook(hook(n))
    try:
        e1 = net({'point_clouds': b0['point_clouds']})
        e2 = net({'point_clouds': b1['point_clouds']})
        e1.update(b0)
        e2.update(b1)
        loss, e1, e2 = loss_helper.get_loss_DA(e1, e2, cfg)
        loss.backward()
    finally:
        fused_sa.can_fuse = real_can
    return {n: (o.detach(), o.grad) for n, o in cap if o.grad is not None}, rec


un, pins = run([])
for label, names in (("all fused", ["sa1", "sa2", "sa3", "sa4", "vote_agg"]),
                     ("sa1-4 fused, vote_agg nine-op", ["sa1", "sa2", "sa3", "sa4"])):
    fu, _ = run(names, pins)
    print("==", label)
    for n in un:
        if n not in fu:
            continue
        va, ga = fu[n]
        vb, gb = un[n]
        if ga.shape != gb.shape:
            continue
        l2 = float((ga - gb).norm() / (gb.norm() + 1e-30))
        mx = float((ga - gb).abs().max() / (gb.abs().max() + 1e-30))
        vl = float((va - vb).abs().max() / (vb.abs().max() + 1e-30))
        flag = "  <--" if l2 > 5e-3 else ""
        print("   %-52s grad L2 %.1e max %.1e   value max %.1e%s" % (n, l2, mx, vl, flag))
