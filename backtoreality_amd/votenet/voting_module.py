"""Vote generation from seed features (detection/Votenet/models/voting_module.py:16-65):
three 1x1 Conv1d (256 -> 256 -> 256 -> (3+256)*vote_factor) with BN+ReLU on the first two;
votes = seed xyz + predicted offset, vote features = seed features + predicted residual."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from ..pointnet2 import _ext, fused_mlp


class _VoteAssemble(Function):
    """votes = seeds + the generator's output (forward :57-64), one launch each way;
    normalize: also the L2 normalisation of the vote features VoteNet applies next."""

    @staticmethod
    def forward(ctx, net, seed_xyz, seed_features, normalize):
        B, C, N = seed_features.shape
        net_cl = _ext.twin_of(net)                           # (B*N, 3 + C)
        seed_cl = _ext.twin_of(seed_features)
        if seed_cl is None or seed_cl.shape != (B * N, C) or not seed_cl.is_contiguous():
            seed_cl = seed_features.transpose(1, 2).contiguous().view(B * N, C)
        dev = net.device
        vote_xyz = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        feat = torch.empty((B, C, N), dtype=torch.float32, device=dev)
        feat_cl = torch.empty((B * N, C), dtype=torch.float32, device=dev)
        nrm = torch.empty((B * N,), dtype=torch.float32, device=dev) if normalize else None
        with _ext._on(net) as d:
            _ext._call(_ext._lib.btr_vote_assemble, B, N, C, _ext._p(net_cl), net_cl.shape[1],
                       _ext._p(seed_xyz.contiguous()), _ext._p(seed_cl), _ext._p(vote_xyz),
                       _ext._p(feat), _ext._p(feat_cl), _ext._p(nrm), _ext._stream(d))
        _ext.attach_twin(feat, feat_cl.view(B, N, C))       # what the vote aggregation gathers
        ctx.dims = (B, C, N)
        if normalize:
            ctx.save_for_backward(feat_cl, nrm)
        return vote_xyz, feat

    @staticmethod
    def backward(ctx, dxyz, dfeat):
        B, C, N = ctx.dims
        dev = (dxyz if dxyz is not None else dfeat).device
        dxyz = dxyz.contiguous() if dxyz is not None else torch.zeros((B, N, 3), device=dev)
        dfeat = dfeat.contiguous() if dfeat is not None else torch.zeros((B, C, N), device=dev)
        dnet = torch.empty((B, 3 + C, N), dtype=torch.float32, device=dxyz.device)
        y_cl, nrm = ctx.saved_tensors if ctx.saved_tensors else (None, None)
        dseed = torch.empty_like(dfeat) if nrm is not None else dfeat
        with _ext._on(dxyz) as d:
            _ext._call(_ext._lib.btr_vote_assemble_bwd, B, N, C, _ext._p(dxyz), _ext._p(dfeat),
                       _ext._p(y_cl), _ext._p(nrm), _ext._p(dnet),
                       _ext._p(dseed) if nrm is not None else None, _ext._stream(d))
        return dnet, None, dseed, None


class VotingModule(nn.Module):
    def __init__(self, vote_factor, seed_feature_dim):
        super().__init__()
        self.vote_factor = vote_factor
        self.in_dim = seed_feature_dim
        self.out_dim = self.in_dim  # residual features: in_dim == out_dim
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv3 = nn.Conv1d(self.in_dim, (3 + self.out_dim) * self.vote_factor, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.bn2 = nn.BatchNorm1d(self.in_dim)

    def forward(self, seed_xyz, seed_features, normalize=False):
        """seed_xyz (B,S,3), seed_features (B,C,S) -> vote_xyz (B,S*vf,3), vote_features (B,C,S*vf).
        normalize (not in the reference's signature): also divide the vote features by their L2
        norm over the channels, which VoteNet.forward does right after (models/votenet.py:98-99)."""
        B, S = seed_xyz.shape[0], seed_xyz.shape[1]
        V = S * self.vote_factor
        net = fused_mlp.run_chain(seed_features, [(self.conv1, self.bn1, True),
                                                  (self.conv2, self.bn2, True),
                                                  (self.conv3, None, False)])
        if net is None:   # stock ops (CPU, eval mode, BTR_FUSED_MLP=0)
            net = F.relu(self.bn1(self.conv1(seed_features)))
            net = F.relu(self.bn2(self.conv2(net)))
            net = self.conv3(net)
        if (self.vote_factor == 1 and net.is_cuda and seed_xyz.dtype == torch.float32 and
                _ext.twin_of(net) is not None and
                os.environ.get("BTR_FUSED_VOTES", "1") != "0" and
                (not normalize or self.out_dim <= 256)):
            return _VoteAssemble.apply(net, seed_xyz, seed_features, bool(normalize))
        net = net.transpose(2, 1).view(B, S, self.vote_factor, 3 + self.out_dim)
        vote_xyz = (seed_xyz.unsqueeze(2) + net[:, :, :, 0:3]).contiguous().view(B, V, 3)
        vote_features = seed_features.transpose(2, 1).unsqueeze(2) + net[:, :, :, 3:]
        vote_features = vote_features.contiguous().view(B, V, self.out_dim)
        vote_features = vote_features.transpose(2, 1).contiguous()
        if normalize:
            vote_features = vote_features.div(torch.norm(vote_features, p=2, dim=1).unsqueeze(1))
        return vote_xyz, vote_features
