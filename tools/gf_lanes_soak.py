#!/usr/bin/env python3
"""Soak test of the decoder stack's replayed graphs and lanes: N forward + backward passes of one
detector on two alternating batches without dropout and without parameter updates -- every pass
over the same batch must reproduce the first one bit for bit (loss, every head's centres, every
decoder / head gradient); a race between the lanes would show as a difference.
Usage: python tools/gf_lanes_soak.py [passes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd import groupfree  # noqa: E402
from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from backtoreality_amd.votenet import config, synthetic  # noqa: E402

LOSS_ARGS = dict(num_decoder_layers=6, query_points_generator_loss_coef=0.8, obj_loss_coef=0.1,
                 box_loss_coef=1, sem_cls_loss_coef=0.1, query_points_obj_topk=4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cuda = torch.device('cuda:0')
cfg = config.scannet_md40()
batches = [synthetic.make_batch(s, 4, 20000, cfg, use_height=False, device=cuda) for s in (3, 4)]
torch.manual_seed(0)
net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                  cfg.mean_size_arr, input_feature_dim=0, num_proposal=256,
                                  dropout=0.0, self_position_embedding='loc_learned',
                                  cross_position_embedding='xyz_learned').to(cuda)
ref = [None, None]
bad = 0
for it in range(n):
    for p in net.parameters():
        p.grad = None
    batch = batches[it % 2]
    ep = net({'point_clouds': batch['point_clouds']})
    ep.update(batch)
    loss, ep = groupfree.get_loss(ep, cfg, **LOSS_ARGS)
    loss.backward()
    cur = {'center:' + k: ep[k].detach().clone() for k in ep if k.endswith('center') and
           torch.is_tensor(ep[k]) and ep[k].is_floating_point()}
    for k, p in net.named_parameters():
        if p.grad is not None and k.startswith(('decoder.', 'prediction_heads.')):
            cur['g:' + k] = p.grad.detach().clone()
    if ref[it % 2] is None:
        ref[it % 2] = cur
        continue
    diff = [k for k in cur if not torch.equal(cur[k], ref[it % 2][k])]
    if diff:
        bad += 1
        print("pass %d: %d tensors differ, e.g. %s" % (it, len(diff), diff[:3]), flush=True)
print("%d passes, %d with a difference; graphs: %s" % (n, bad, _ext.graph_stats()))
sys.exit(1 if bad else 0)
