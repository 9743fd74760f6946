#!/usr/bin/env python3
"""Everything btr_backbone_sampling computes beside the running step -- level-1 FPS, the centre
gathers, the four ball queries, the two 3-NN tables -- against the same ops run ALONE on an
idle device, every step of the software-pipelined loop on changing batches: bit for bit.
(Companion of tools/diag_pipeline_inds.py, which checks levels 2-4 of the FPS pyramid.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.pointnet2 import fused_backbone as _fb, pointnet2_utils as pu
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev, seed=0)
opt = train.make_optimizer(net)
B, N = int(os.environ.get("B", 4)), int(os.environ.get("N", 20000))
batches = [synthetic.make_batch(7 * s, B, N, cfg, device=dev) for s in range(5)]
iters = int(os.environ.get("ITERS", "500"))
LAST = {}
_orig_apply = _fb.FusedBackboneFn.apply


def _spy(cloud, handle, entry, *params):
    LAST["h"], LAST["cloud"] = handle, cloud
    return _orig_apply(cloud, handle, entry, *params)


_fb.FusedBackboneFn.apply = staticmethod(_spy)
bb = net.backbone_net
radii = [bb.sa1.grouper.radius, bb.sa2.grouper.radius, bb.sa3.grouper.radius, bb.sa4.grouper.radius]
nsamp = [bb.sa1.grouper.nsample, bb.sa2.grouper.nsample, bb.sa3.grouper.nsample, bb.sa4.grouper.nsample]
bad = {}
sampling = bb.prefetch_sampling(batches[0]['point_clouds'])
for it in range(iters):
    b = batches[it % 5]
    loss, end = train.train_step(net, opt, b, cfg, sampling=sampling,
                                 next_batch=batches[(it + 1) % 5])
    sampling = end['next_sampling']
    torch.cuda.synchronize()
    h = LAST["h"]
    xyz = LAST["cloud"][..., :3].contiguous()
    cur = xyz
    for l in range(4):
        if l == 0:
            inds = pu.furthest_point_sample(cur, h.inds[0].shape[1])
            if not torch.equal(inds, h.inds[0]):
                bad["fps1"] = bad.get("fps1", 0) + 1
        new_xyz = pu.gather_rows(cur, h.inds[l])
        if not torch.equal(new_xyz, h.xyz[l]):
            bad["xyz%d" % (l + 1)] = bad.get("xyz%d" % (l + 1), 0) + 1
        idx = pu.ball_query(radii[l], nsamp[l], cur, new_xyz)
        if not torch.equal(idx, h.idx(l)):
            bad["bq%d" % (l + 1)] = bad.get("bq%d" % (l + 1), 0) + 1
        cur = new_xyz
    for j, (u, k) in enumerate(((2, 3), (1, 2))):   # fp1: sa3 <- sa4, fp2: sa2 <- sa3
        idx, w = pu.three_nn_weights(h.xyz[u], h.xyz[k])
        hi, hw = h.three_nn(j)
        if not (torch.equal(idx, hi) and torch.equal(w, hw)):
            bad["nn%d" % (j + 1)] = bad.get("nn%d" % (j + 1), 0) + 1
    torch.cuda.synchronize()
print("pipelined steps", iters, "items that differed from the stand-alone ops (item: count):", bad)
