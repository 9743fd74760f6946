"""Sampling and head modules of GroupFree3D (detection/GroupFree3D/models/modules.py), same
class / attribute names so reference checkpoints load."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2 import fused_mlp, pointnet2_utils
from . import fused_decode


class PointsObjClsModule(nn.Module):
    """Per-seed objectness logit used to pick the query points (modules.py:17-47):
    (B, C, num_seed) -> (B, 1, num_seed)."""

    def __init__(self, seed_feature_dim):
        super().__init__()
        self.in_dim = seed_feature_dim
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.bn2 = nn.BatchNorm1d(self.in_dim)
        self.conv3 = nn.Conv1d(self.in_dim, 1, 1)

    def forward(self, seed_features):
        net = fused_mlp.run_chain(seed_features, [(self.conv1, self.bn1, True),
                                                  (self.conv2, self.bn2, True),
                                                  (self.conv3, None, False)])
        if net is not None:   # the chain on this package's GEMM / BatchNorm kernels
            return net
        net = F.relu(self.bn1(self.conv1(seed_features)))
        net = F.relu(self.bn2(self.conv2(net)))
        return self.conv3(net)


class PositionEmbeddingLearned(nn.Module):
    """Learned absolute position embedding: (B, P, 3|6) -> (B, 288, P) (modules.py:50-65)."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(input_channel, num_pos_feats, kernel_size=1),
            nn.BatchNorm1d(num_pos_feats),
            nn.ReLU(inplace=True),
            nn.Conv1d(num_pos_feats, num_pos_feats, kernel_size=1))

    def forward(self, xyz):
        # (B, C, P) form: made once per coordinate tensor (the key positions feed every decoder
        # layer's embedding; the head decode kernel writes the query positions in both forms)
        cached = getattr(xyz, '_btr_t', None)   # (transpose, version of xyz it was made from)
        if cached is not None and cached[1] == xyz._version:
            x = cached[0]
        else:
            x = xyz.transpose(1, 2).contiguous()
            if not xyz.requires_grad:
                xyz._btr_t = (x, xyz._version)
        head = self.position_embedding_head
        out = fused_mlp.run_chain(x, [(head[0], head[1], True), (head[3], None, False)])
        return out if out is not None else head(x)


def _take(xyz, features, sample_inds):
    """Rows `sample_inds` of (B,K,3) coordinates and (B,C,K) features (gather_points kernel)."""
    new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(),
                                               sample_inds).transpose(1, 2).contiguous()
    new_features = pointnet2_utils.gather_operation(features, sample_inds).contiguous()
    return new_xyz, new_features, sample_inds


class FPSModule(nn.Module):
    """Query points by furthest point sampling of the seeds (modules.py:68-86)."""

    def __init__(self, num_proposal):
        super().__init__()
        self.num_proposal = num_proposal

    def forward(self, xyz, features):
        return _take(xyz, features, pointnet2_utils.furthest_point_sample(xyz, self.num_proposal))


class GeneralSamplingModule(nn.Module):
    """Query points at given seed indices (modules.py:89-104)."""

    def forward(self, xyz, features, sample_inds):
        return _take(xyz, features, sample_inds)


class _CatConv(object):
    """Several 1x1 Conv1d layers of the same input seen as ONE layer (weights / biases
    concatenated along the output channels): what fused_mlp.run_chain needs of a conv."""
    kernel_size, stride, groups = (1,), (1,), 1

    def __init__(self, convs):
        self.weight = torch.cat([c.weight for c in convs], 0)
        self.bias = torch.cat([c.bias for c in convs], 0)
        self.in_channels = convs[0].in_channels
        self.out_channels = self.weight.shape[0]


class PredictHead(nn.Module):
    """Box head on (B, C, num_proposal) features (modules.py:107-193): objectness (1 logit),
    centre residual w.r.t. `base_xyz`, heading / size class scores and normalised residuals,
    semantic scores; fills end_points['<prefix>...'] and returns (center, pred_size)."""

    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal,
                 seed_feat_dim=256):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_proposal = num_proposal
        self.seed_feat_dim = seed_feat_dim

        self.conv1 = nn.Conv1d(seed_feat_dim, seed_feat_dim, 1)
        self.bn1 = nn.BatchNorm1d(seed_feat_dim)
        self.conv2 = nn.Conv1d(seed_feat_dim, seed_feat_dim, 1)
        self.bn2 = nn.BatchNorm1d(seed_feat_dim)

        self.objectness_scores_head = nn.Conv1d(seed_feat_dim, 1, 1)
        self.center_residual_head = nn.Conv1d(seed_feat_dim, 3, 1)
        self.heading_class_head = nn.Conv1d(seed_feat_dim, num_heading_bin, 1)
        self.heading_residual_head = nn.Conv1d(seed_feat_dim, num_heading_bin, 1)
        self.size_class_head = nn.Conv1d(seed_feat_dim, num_size_cluster, 1)
        self.size_residual_head = nn.Conv1d(seed_feat_dim, num_size_cluster * 3, 1)
        self.sem_cls_scores_head = nn.Conv1d(seed_feat_dim, self.num_class, 1)
        self._mean_size = None

    def _mean_size_on(self, device):
        if self._mean_size is None or self._mean_size.device != device:
            self._mean_size = torch.from_numpy(
                np.ascontiguousarray(self.mean_size_arr, np.float32)).to(device)
        return self._mean_size

    def _heads(self):
        return (self.objectness_scores_head, self.center_residual_head, self.heading_class_head,
                self.heading_residual_head, self.size_class_head, self.size_residual_head,
                self.sem_cls_scores_head)

    def chain(self, last=None):
        """The head as one conv/BN/ReLU chain for fused_mlp.run_chain (and the decoder stack).
        The seven output layers are 1x1 convolutions of the same `net`: one convolution with the
        concatenated weights (116 output channels at ScanNet sizes) instead of seven of 1..66
        channels; the parameters stay separate (state-dict keys of the reference).  `last`: a
        stand-in for the concatenated layer (the decoder stack's slots keep their own copy of it
        and fill it from the seven layers themselves: no torch.cat)."""
        return [(self.conv1, self.bn1, True), (self.conv2, self.bn2, True),
                (last if last is not None else _CatConv(self._heads()), None, False)]

    def cat_standin(self, device):
        """A _CatConv-shaped object whose weight / bias only carry the shapes (never read)."""
        s = self.__dict__.get('_cat_standin')
        if s is None or s.weight.device != device:
            heads = self._heads()
            s = _CatConv.__new__(_CatConv)
            n = sum(h.out_channels for h in heads)
            s.weight = torch.empty((n, heads[0].in_channels, 1), dtype=torch.float32, device=device)
            s.bias = torch.empty((n,), dtype=torch.float32, device=device)
            s.in_channels, s.out_channels = heads[0].in_channels, n
            self.__dict__['_cat_standin'] = s
        return s

    def forward(self, features, base_xyz, end_points, prefix=''):
        chain = self.chain()
        out = fused_mlp.run_chain(features, chain)
        if out is None:   # stock ops (CPU, eval mode, BTR_FUSED_MLP=0)
            net = F.relu(self.bn1(self.conv1(features)))
            net = F.relu(self.bn2(self.conv2(net)))
            out = F.conv1d(net, chain[-1][0].weight, chain[-1][0].bias)
        dec = fused_decode.decode(out, base_xyz, self._mean_size_on(features.device),
                                  self.num_heading_bin, self.num_size_cluster)
        return self.publish(out, dec, base_xyz, end_points, prefix)

    def publish(self, out_raw, dec, base_xyz, end_points, prefix):
        """Fills end_points['<prefix>...'] from the raw head output (B, sum, P) and, when given,
        the decoded tensors (center, heading_residuals, size_residuals, pred_size, query_pos,
        query_pos_t) of fused_decode.decode; returns (center, pred_size)."""
        B, P = out_raw.shape[0], out_raw.shape[-1]
        heads = self._heads()
        end_points[prefix + '_head_output'] = out_raw   # what groupfree/fused_loss.py reads
        out = out_raw.transpose(2, 1)   # (B, P, sum)
        (objectness_scores, center_residual, heading_scores, heading_residuals_normalized,
         size_scores, size_residuals_flat, sem_cls_scores) = torch.split(
            out, [h.out_channels for h in heads], dim=2)
        mean_size_2d = self._mean_size_on(out_raw.device)
        size_residuals_normalized = size_residuals_flat.reshape(B, P, self.num_size_cluster, 3)
        if dec is not None:   # one launch (csrc/gf_loss.hip); also the next layer's query position
            center, heading_residuals, size_residuals, pred_size, qpos, qpos_t = dec
            qpos._btr_t = (qpos_t, qpos._version)
            center._btr_query_pos = (qpos, center._version, pred_size, pred_size._version)
        else:
            center = base_xyz + center_residual
            heading_residuals = heading_residuals_normalized * (np.pi / self.num_heading_bin)
            mean_size = mean_size_2d.unsqueeze(0).unsqueeze(0)
            size_residuals = size_residuals_normalized * mean_size
            size_recover = size_residuals + mean_size
            pick = torch.argmax(size_scores, -1).unsqueeze(-1).unsqueeze(-1).expand(-1, -1, 1, 3)
            pred_size = torch.gather(size_recover, 2, pick).squeeze(2)

        for key, value in (('base_xyz', base_xyz), ('objectness_scores', objectness_scores),
                           ('center', center), ('heading_scores', heading_scores),
                           ('heading_residuals_normalized', heading_residuals_normalized),
                           ('heading_residuals', heading_residuals),
                           ('size_scores', size_scores),
                           ('size_residuals_normalized', size_residuals_normalized),
                           ('size_residuals', size_residuals), ('pred_size', pred_size),
                           ('sem_cls_scores', sem_cls_scores)):
            end_points[prefix + key] = value
        # the fused per-head loss reads `_head_output`, not the entries above: it may only do so
        # while they still ARE this output's views and nothing was written to them in place
        # (views share the version counter of their base)
        out_raw._btr_head_views = (out_raw._version, base_xyz, {
            'objectness_scores': objectness_scores, 'heading_scores': heading_scores,
            'heading_residuals_normalized': heading_residuals_normalized,
            'size_scores': size_scores, 'size_residuals_normalized': size_residuals_normalized,
            'sem_cls_scores': sem_cls_scores})
        return center, pred_size
