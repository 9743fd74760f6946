#!/usr/bin/env python3
"""Every kernel of the forward pass, beside another stream's work, against the same forward run
alone: the backbone forward is bit-deterministic (tests/test_backbone_gpu.py), so the outputs of
a forward that overlaps the NEXT batch's sampling pyramid (side stream) must equal, bit for bit,
those of a copy of the model run on an idle device.  Catches any kernel whose result depends on
what else shares the chip (cf. tools/diag_pipeline_inds.py)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev, seed=0)
B, N = int(os.environ.get("B", 4)), int(os.environ.get("N", 20000))
batches = [synthetic.make_batch(7 * s, B, N, cfg, device=dev) for s in range(5)]
iters = int(os.environ.get("ITERS", "300"))
KEYS = ("sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features",
        "vote_xyz", "aggregated_vote_xyz", "objectness_scores", "center")
bad = {}
bb = net.backbone_net
for it in range(iters):
    b = batches[it % 5]
    ref_net = copy.deepcopy(net)
    torch.cuda.synchronize()
    # in flow: this batch's sampling was prefetched; the next batch's pyramid runs beside the forward
    h = bb.prefetch_sampling(b['point_clouds'])
    torch.cuda.synchronize()
    nxt = bb.prefetch_sampling(batches[(it + 1) % 5]['point_clouds'])
    end = net({'point_clouds': b['point_clouds'], 'sampling': h})
    got = {k: end[k].detach().clone() for k in KEYS if k in end}
    torch.cuda.synchronize()
    del nxt
    # alone
    end2 = ref_net({'point_clouds': b['point_clouds']})
    torch.cuda.synchronize()
    for k in got:
        if not torch.equal(got[k], end2[k]):
            bad[k] = bad.get(k, 0) + 1
    if it == 0:
        print("compared keys:", sorted(got))
print("forwards", iters, "keys that differed (key: count):", bad)
