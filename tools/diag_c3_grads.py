#!/usr/bin/env python3
"""Full-size Back-to-Reality step (2 x 8 x 40 000), fused HIP path against the nine-op + torch
composition with pinned proposals, for several scene seeds: which parameters' gradients deviate
and by how much (tests/test_configs_gpu.py::test_c3_back_to_reality_full_size_step picks its
scenes with this).  Usage: diag_c3_grads.py [first_seed ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402


def step(cfg, bS, bT, dev, fused, pins=None):
    os.environ["BTR_FUSED_SA"] = "1" if fused else "0"
    net = train.build_model(cfg, dev, seed=0, domain_adaptation=True)
    if pins is not None:
        sa = net.pnet.vote_aggregation
        own, queue = sa.forward, list(pins)
        sa.forward = lambda xyz, features=None, inds=None: own(xyz, features, queue.pop(0))
    eS = net({'point_clouds': bS['point_clouds']})
    eT = net({'point_clouds': bT['point_clouds']})
    eS.update(bS)
    eT.update(bT)
    loss, eS, eT = loss_helper.get_loss_DA(eS, eT, cfg)
    loss.backward()
    return loss.detach(), eS, eT, {n: p.grad.detach().clone() for n, p in net.named_parameters()
                                   if p.grad is not None}


def main():
    dev = torch.device("cuda:0")
    cfg = config.scannet_md40()
    seeds = [int(a) for a in sys.argv[1:]] or [0, 8, 16, 24]
    for s in seeds:
        bS = synthetic.make_batch(s, 8, 40000, cfg, device=dev)
        bT = synthetic.make_batch(100000 + s, 8, 40000, cfg, device=dev)
        lu, uS, uT, gu = step(cfg, bS, bT, dev, False)
        lf, fS, fT, gf = step(cfg, bS, bT, dev, True,
                              (uS['aggregated_vote_inds'], uT['aggregated_vote_inds']))
        gmax = max(float(g.abs().max()) for g in gu.values())
        devs = sorted(((float((gf[n] - gu[n]).norm() / (gu[n].norm() + 1e-20)), n) for n in gu
                       if float(gu[n].abs().max()) > 1e-4 * gmax), reverse=True)
        print("seed %d: loss rel %.2e; worst %.4f; first non-backbone %.4f" % (
            s, abs(float(lf) - float(lu)) / abs(float(lu)), devs[0][0],
            max(d for d, n in devs if not n.startswith("backbone_net.sa"))))
        for d, n in devs[:6]:
            print("   %.4f  %s" % (d, n))
        # where the deviation starts: worst relative L2 per module, in BACKWARD order (the heads
        # first); a module the deviation has not reached yet sits at float32 rounding level
        stages = ('backbone_net.sa1.', 'backbone_net.sa2.', 'backbone_net.sa3.',
                  'backbone_net.sa4.', 'backbone_net.fp1.', 'backbone_net.fp2.', 'vgen.',
                  'pnet.vote_aggregation.')
        by = {}
        for dv, n in devs:
            st = next((p for p in stages if n.startswith(p)), n.split('.')[0] + '.' + n.split('.')[1])
            if st not in by or dv > by[st][0]:
                by[st] = (dv, n)
        order = [k for k in by if k not in stages] + list(reversed(stages))
        for k in order:
            if k in by:
                print("   stage %-28s worst %.5f  %s" % (k, by[k][0], by[k][1]))
        for tag, f, u in (("S", fS, uS), ("T", fT, uT)):
            for k in ('object_assignment', 'objectness_label', 'objectness_mask'):
                if k in f and k in u:
                    print("   %s %s differing entries: %d" % (tag, k, int((f[k] != u[k]).sum())))
        # the flip-aware reading of tests/test_configs_gpu.py: rows of every pooled layer's weight
        # gradient that moved by more than 1e-2 of the largest row, most downstream layer first
        for prefix in ('pnet.vote_aggregation.', 'backbone_net.sa4.', 'backbone_net.sa3.',
                       'backbone_net.sa2.', 'backbone_net.sa1.'):
            last = sorted(n for n in gu if n.startswith(prefix) and n.endswith('.conv.weight'))[-1]
            wf, wu = gf[last].flatten(1), gu[last].flatten(1)
            row = (wf - wu).norm(dim=1) / float(wu.norm(dim=1).max())
            top = row.topk(6)
            print("   %-52s rows > 1e-2: %3d   largest %s" % (
                last, int((row > 1e-2).sum()), ["%.3f" % float(v) for v in top.values]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
