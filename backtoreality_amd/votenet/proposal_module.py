"""Vote aggregation + proposal head (detection/Votenet/models/proposal_module.py).

`vote_aggregation` is the hot-path piece: a fifth set-abstraction layer run on the votes
(npoint=num_proposal, r=0.3, nsample=16, mlp [256,128,128,128], :66-73), whose `xyz` input
requires grad, so backward reaches group_points_grad AND gather_points_grad.  The head
(3x Conv1d) and `decode_scores` (:18-50) are stock torch ops.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2 import fused_mlp, pointnet2_utils
from ..pointnet2.pointnet2_modules import PointnetSAModuleVotes


def decode_scores(net, end_points, num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                  mean_size_dev=None):
    """Split the (B, 2+3+NH*2+NS*4+NC, K) head output into named predictions (:18-50).
    `mean_size_dev`: the (NS,3) table already on the device (the reference re-uploads the
    numpy array on every call, :40 -- a synchronous H2D copy per step)."""
    t = net.transpose(2, 1)  # (B, K, channels)
    B, K = t.shape[0], t.shape[1]
    NH, NS = num_heading_bin, num_size_cluster

    end_points['objectness_scores'] = t[:, :, 0:2]
    end_points['center'] = end_points['aggregated_vote_xyz'] + t[:, :, 2:5]

    o = 5
    end_points['heading_scores'] = t[:, :, o:o + NH]
    hres = t[:, :, o + NH:o + 2 * NH]
    end_points['heading_residuals_normalized'] = hres  # in [-1, 1]
    end_points['heading_residuals'] = hres * (np.pi / NH)

    o += 2 * NH
    size_scores = t[:, :, o:o + NS]
    sres = t[:, :, o + NS:o + 4 * NS].view([B, K, NS, 3])
    end_points['size_scores'] = size_scores
    end_points['size_residuals_normalized'] = sres
    if mean_size_dev is None or mean_size_dev.device != net.device:
        mean_size_dev = torch.from_numpy(mean_size_arr.astype(np.float32)).to(net.device)
    mean_size = mean_size_dev.unsqueeze(0).unsqueeze(0)
    end_points['size_residuals'] = sres * mean_size
    size_recover = mean_size + end_points['size_residuals']  # (B, K, NS, 3)
    pred_cls = torch.argmax(size_scores, -1)
    pred_cls = pred_cls.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, 3)
    end_points['pred_size'] = torch.gather(size_recover, 2, pred_cls).squeeze_(2)  # (B, K, 3)

    end_points['sem_cls_scores'] = t[:, :, o + 4 * NS:]
    return end_points


DECODED_KEYS = frozenset([
    'objectness_scores', 'center', 'heading_scores', 'heading_residuals_normalized',
    'heading_residuals', 'size_scores', 'size_residuals_normalized', 'size_residuals',
    'pred_size', 'sem_cls_scores'])


class DecodedEndPoints(dict):
    """`end_points` whose decode_scores entries are computed on first use.

    A training step through the fused loss (votenet/fused_loss.py) reads the raw head output
    and never touches the decoded predictions: decoding them eagerly, as the reference does
    (proposal_module.py:108-113), is ten launches per step for nothing.  Any access -- a decoded
    key, `in`, `get`, iteration, `len`, `keys/items/values`, `copy` -- decodes first, so the
    dict always LOOKS like the reference's; only plain reads / writes of other keys do not."""

    def __init__(self, base=(), decode=None):   # (dict-like construction: nn.DataParallel's
        super().__init__(base)                  # gather rebuilds type(out)(pairs))
        self._decode = decode

    def _force(self):
        decode, self._decode = self._decode, None
        if decode is not None:
            decode(self)

    def __missing__(self, key):
        if self._decode is not None and key in DECODED_KEYS:
            self._force()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (self._decode is not None and key in DECODED_KEYS)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __iter__(self):
        self._force()
        return dict.__iter__(self)

    def __len__(self):
        self._force()
        return dict.__len__(self)

    def keys(self):
        self._force()
        return dict.keys(self)

    def items(self):
        self._force()
        return dict.items(self)

    def values(self):
        self._force()
        return dict.values(self)

    def copy(self):
        self._force()
        return dict(self)


class ProposalModule(nn.Module):
    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                 num_proposal, sampling, seed_feat_dim=256):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_proposal = num_proposal
        self.sampling = sampling
        self.seed_feat_dim = seed_feat_dim

        self.vote_aggregation = PointnetSAModuleVotes(
            npoint=self.num_proposal, radius=0.3, nsample=16,
            mlp=[self.seed_feat_dim, 128, 128, 128], use_xyz=True, normalize_xyz=True)

        out_ch = 2 + 3 + num_heading_bin * 2 + num_size_cluster * 4 + self.num_class
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.conv2 = nn.Conv1d(128, 128, 1)
        self.conv3 = nn.Conv1d(128, out_ch, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.bn2 = nn.BatchNorm1d(128)
        # device copy of the size table; non-persistent so the state dict keeps the
        # reference's keys
        self.register_buffer("_mean_size_dev",
                             torch.from_numpy(np.asarray(mean_size_arr, np.float32)),
                             persistent=False)

    def forward(self, xyz, features, end_points):
        """xyz (B,K,3) votes, features (B,C,K) -> end_points with the decoded proposals."""
        if self.sampling == 'vote_fps':
            xyz, features, sample_inds = self.vote_aggregation(xyz, features)
        elif self.sampling == 'seed_fps':
            # FPS on the seeds, then aggregate the votes of the chosen seeds (:97-100)
            sample_inds = pointnet2_utils.furthest_point_sample(end_points['seed_xyz'],
                                                                self.num_proposal)
            xyz, features, _ = self.vote_aggregation(xyz, features, sample_inds)
        elif self.sampling == 'random':
            num_seed = end_points['seed_xyz'].shape[1]
            sample_inds = torch.randint(0, num_seed, (xyz.shape[0], self.num_proposal),
                                        dtype=torch.int, device=xyz.device)
            xyz, features, _ = self.vote_aggregation(xyz, features, sample_inds)
        else:
            raise ValueError('Unknown sampling strategy: %s' % (self.sampling,))
        end_points['aggregated_vote_xyz'] = xyz            # (B, num_proposal, 3)
        end_points['aggregated_vote_features'] = features  # (B, 128, num_proposal)
        end_points['aggregated_vote_inds'] = sample_inds   # (B, num_proposal)

        net = fused_mlp.run_chain(features, [(self.conv1, self.bn1, True),
                                             (self.conv2, self.bn2, True),
                                             (self.conv3, None, False)])
        if net is None:   # stock ops (CPU, eval mode, BTR_FUSED_MLP=0)
            net = F.relu(self.bn1(self.conv1(features)))
            net = F.relu(self.bn2(self.conv2(net)))
            net = self.conv3(net)
        end_points['_head_output'] = net  # raw (B, Cout, K) scores for the fused loss

        def decode(ep):
            decode_scores(net, ep, self.num_class, self.num_heading_bin, self.num_size_cluster,
                          self.mean_size_arr, self._mean_size_dev)
        if self.training and torch.is_grad_enabled() and net.is_cuda and \
                os.environ.get("BTR_LAZY_DECODE", "1") != "0":
            return DecodedEndPoints(end_points, decode)   # decoded when somebody looks
        decode(end_points)
        return end_points
