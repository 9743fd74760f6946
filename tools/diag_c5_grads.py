#!/usr/bin/env python3
"""C5 (4 x 80 000 points, Matterport heads): fused vs op-by-op gradient deviation per scene
seed.  One max-pool / ReLU element within float32 noise of a tie flips between two correct
implementations, changes ONE row of one weight gradient by O(1) and reaches every layer upstream
diffusely (seed 0 at batch 4: sa4 layer 2, channel 83: 0.30 in that row, <= 0.003 in the 255
others, 4 % relative L2 upstream -- identical on the whole-backbone and the layer-by-layer fused
path).  tests/test_configs_gpu.py uses a seed without such an element."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic  # noqa: E402
import test_configs_gpu as T  # noqa: E402


class MP(object):
    def setenv(self, k, v):
        os.environ[k] = v


cuda = torch.device('cuda:0')
cfg = config.matterport_md40()
B = 4
for seed in [int(a) for a in sys.argv[1:]] or [0, 4, 8, 12, 16]:
    batch = synthetic.make_batch(seed, B, 80000, cfg, extent_scale=1.7, device=cuda)
    loss_u, end_u, g_u = T._votenet_step(cfg, batch, cuda, False, MP())
    loss_f, end_f, g_f = T._votenet_step(cfg, batch, cuda, True, MP(),
                                         vote_inds=end_u['aggregated_vote_inds'])
    gmax = max(float(g.abs().max()) for g in g_u.values())
    rows = []
    for n in g_u:
        if float(g_u[n].abs().max()) > 1e-4 * gmax:
            rows.append((float((g_f[n] - g_u[n]).norm() / (g_u[n].norm() + 1e-20)), n))
    rows.sort(reverse=True)
    print("seed %d: worst rel L2 %.4f (%s), loss rel diff %.1e" % (
        seed, rows[0][0], rows[0][1], abs(float(loss_f) - float(loss_u)) / float(loss_u)))
