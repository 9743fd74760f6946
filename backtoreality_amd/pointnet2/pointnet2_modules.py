"""PointNet++ set-abstraction / feature-propagation modules with the reference's class names,
keyword-only constructors, return tuples and parameter names
(/root/reference/detection/Votenet/pointnet2/pointnet2_modules.py), built on this package's
`pointnet2_utils` (HIP kernels) and `pytorch_utils`.

Hot-path classes: `PointnetSAModuleVotes` (:164-272) and `PointnetFPModule` (:454-514).
The MSG / Centers / LFP variants are kept for API coverage and share the same code path.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

if __package__:
    from . import pointnet2_utils
    from . import pytorch_utils as pt_utils
    from . import fused_sa
    from . import fused_mlp
    from . import _twin
else:
    import os
    import sys
    sys.path.append(os.path.dirname(os.path.abspath(__file__)))
    import pointnet2_utils
    import pytorch_utils as pt_utils
    import fused_sa
    import fused_mlp
    import _twin


def _sample_centres(xyz, npoint, inds=None):
    """FPS (unless `inds` is given) + gather of the sampled coordinates (:233-240).
    Returns (new_xyz (B,npoint,3) or None, inds)."""
    if npoint is None:
        return None, inds
    if inds is None:
        inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    # gathered by whoever sampled (prefetched pyramid); void once inds / xyz were modified
    pre = _twin.derived(inds, "_btr_new_xyz", xyz)
    if pre is not None and pre[1] is xyz and pre[0].shape == (xyz.shape[0], inds.shape[1], 3):
        return pre[0], inds
    return pointnet2_utils.gather_rows(xyz, inds), inds


def _pool(new_features, grouped_xyz, pooling, sigma, nsample):
    """Reduce the nsample axis of (B,C,npoint,nsample): max / avg / RBF-weighted (:254-267)."""
    if pooling == 'max':
        out = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)])
    elif pooling == 'avg':
        out = F.avg_pool2d(new_features, kernel_size=[1, new_features.size(3)])
    elif pooling == 'rbf':
        rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (sigma ** 2) / 2)
        out = torch.sum(new_features * rbf.unsqueeze(1), -1, keepdim=True) / float(nsample)
    else:
        raise ValueError("unknown pooling %r" % (pooling,))
    return out.squeeze(-1)


def _make_groupers(npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly):
    groupers, nets = nn.ModuleList(), nn.ModuleList()
    for radius, nsample, spec in zip(radii, nsamples, mlps):
        groupers.append(
            pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                          sample_uniformly=sample_uniformly)
            if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
        if use_xyz:
            spec[0] += 3  # in place on the caller's list, like the reference (:117-118)
        nets.append(pt_utils.SharedMLP(spec, bn=bn))
    return groupers, nets


def _group_mlp_max(grouper, net, xyz, new_xyz, features):
    f = net(grouper(xyz, new_xyz, features))
    return F.max_pool2d(f, kernel_size=[1, f.size(3)]).squeeze(-1)


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None):
        new_xyz, _ = _sample_centres(xyz, self.npoint)
        outs = [_group_mlp_max(g, m, xyz, new_xyz, features)
                for g, m in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale grouping set abstraction (:75-123)."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int],
                 mlps: List[List[int]], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers, self.mlps = _make_groupers(npoint, radii, nsamples, mlps, bn, use_xyz,
                                                  sample_uniformly)


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction (:126-161)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz)


class _SingleScaleSA(nn.Module):
    """Shared constructor of PointnetSAModuleVotes / PointnetSAModuleCenters."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True,
                 pooling: str = 'max', sigma: float = None, normalize_xyz: bool = False,
                 sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.pooling = pooling
        self.mlp_module = None
        self.use_xyz = use_xyz
        self.sigma = sigma
        if self.sigma is None:
            self.sigma = self.radius / 2
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt

        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)

        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

    def train(self, mode: bool = True):
        # constants derived from the running statistics for the inference forward
        # (fused_sa._eval_constants) do not outlive a mode switch
        self._btr_eval_consts = None
        return super().train(mode)

    def _group_and_pool(self, xyz, new_xyz, features):
        if new_xyz is not None and fused_sa.can_fuse(self, xyz, features):
            # MI355X fast path: grouping + shared MLP + max-pool as fused HIP kernels
            return fused_sa.fused_group_mlp_max(self, xyz, new_xyz, features), None
        grouped = self.grouper(xyz, new_xyz, features)
        unique_cnt = grouped[2] if self.ret_unique_cnt else None
        grouped_features, grouped_xyz = grouped[0], grouped[1]
        new_features = self.mlp_module(grouped_features)
        return _pool(new_features, grouped_xyz, self.pooling, self.sigma, self.nsample), unique_cnt


class PointnetSAModuleVotes(_SingleScaleSA):
    """Set abstraction that also returns the sampled indices (for GT votes) (:164-272).

    forward(xyz (B,N,3), features (B,C,N), inds=None) ->
        (new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds (B,npoint) int32)
        [+ unique_cnt when ret_unique_cnt]."""

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None,
                inds: torch.Tensor = None):
        if inds is not None:
            assert inds.shape[1] == self.npoint
        new_xyz, inds = _sample_centres(xyz, self.npoint, inds)
        new_features, unique_cnt = self._group_and_pool(xyz, new_xyz, features)
        if not self.ret_unique_cnt:
            return new_xyz, new_features, inds
        return new_xyz, new_features, inds, unique_cnt


class PointnetSAModuleMSGVotes(nn.Module):
    """Multi-scale variant returning the sampled indices (:275-354)."""

    def __init__(self, *, mlps: List[List[int]], npoint: int, radii: List[float],
                 nsamples: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.npoint = npoint
        self.groupers, self.mlps = _make_groupers(npoint, radii, nsamples, mlps, bn, use_xyz,
                                                  sample_uniformly)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None,
                inds: torch.Tensor = None):
        new_xyz, inds = _sample_centres(xyz, self.npoint, inds)
        outs = [_group_mlp_max(g, m, xyz, new_xyz, features)
                for g, m in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(outs, dim=1), inds


class PointnetSAModuleCenters(_SingleScaleSA):
    """Set abstraction around GIVEN centres (CenterRefine models) (:357-451):
    forward(xyz, features, centers (B,npoint,3)) -> new_features (B,mlp[-1],npoint)."""

    def forward(self, xyz: torch.Tensor, features: torch.Tensor, centers: torch.Tensor):
        if self.ret_unique_cnt:
            return None  # the reference falls through without a result here (:414-451)
        new_features, _ = self._group_and_pool(xyz, centers, features)
        return new_features


class PointnetSAModuleOffset(_SingleScaleSA):
    """Set abstraction around GIVEN points, GroupFree3D's detector_DA
    (detection/GroupFree3D/pointnet2/pointnet2_modules.py:481-576):
    forward(xyz, features, new_xyz (B,npoint,3)) -> new_features (B,mlp[-1],npoint)
    [, unique_cnt when ret_unique_cnt]."""

    def forward(self, xyz: torch.Tensor, features: torch.Tensor, new_xyz: torch.Tensor):
        new_features, unique_cnt = self._group_and_pool(xyz, new_xyz, features)
        if not self.ret_unique_cnt:
            return new_features
        return new_features, unique_cnt


def ThreeNNInterpolate(known_feats, known_xyz, unknown_xyz):
    """Inverse-distance 3-NN interpolation of `known_feats (B,C,m)` at `unknown_xyz (B,n,3)`
    (detection/GroupFree3D/pointnet2/pointnet2_modules.py:722-730)."""
    dist, idx = pointnet2_utils.three_nn(unknown_xyz, known_xyz)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    return pointnet2_utils.three_interpolate(known_feats, idx, weight)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance 3-NN interpolation of `known_feats` onto the
    `unknown` points, concatenated with their skip features, then a SharedMLP (:454-514)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        if known is not None:
            # computed with a prefetched pyramid; void once unknown / known were modified
            pre = _twin.derived(unknown, "_btr_three_nn", known)
            if pre is not None and pre[0] is known:
                idx, weight = pre[1], pre[2]
            elif unknown.is_cuda and hasattr(pointnet2_utils._ext, "three_nn_weights"):
                idx, weight = pointnet2_utils.three_nn_weights(unknown, known)   # one launch
            else:
                dist, idx = pointnet2_utils.three_nn(unknown, known)
                dist_recip = 1.0 / (dist + 1e-8)
                norm = torch.sum(dist_recip, dim=2, keepdim=True)
                weight = dist_recip / norm
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))

        if unknow_feats is not None:
            new_features = torch.cat([interpolated, unknow_feats], dim=1)
        else:
            new_features = interpolated
        chain = fused_mlp.shared_mlp_chain(self.mlp)
        out = fused_mlp.run_chain(new_features, chain) if chain is not None else None
        if out is not None:   # MI355X fast path: the SharedMLP as fused GEMM + BN kernels
            return out
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)


class PointnetLFPModuleMSG(nn.Module):
    """Learnable feature propagation (:517-595): group `features1` around `xyz2`, MLP + max,
    concatenate `features2`, post-MLP."""

    def __init__(self, *, mlps: List[List[int]], radii: List[float], nsamples: List[int],
                 post_mlp: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.post_mlp = pt_utils.SharedMLP(post_mlp, bn=bn)
        self.groupers, self.mlps = _make_groupers(0, radii, nsamples, mlps, bn, use_xyz,
                                                  sample_uniformly)

    def forward(self, xyz2: torch.Tensor, xyz1: torch.Tensor, features2: torch.Tensor,
                features1: torch.Tensor) -> torch.Tensor:
        outs = []
        for grouper, net in zip(self.groupers, self.mlps):
            f = _group_mlp_max(grouper, net, xyz1, xyz2, features1)
            if features2 is not None:
                f = torch.cat([f, features2], dim=1)
            outs.append(self.post_mlp(f.unsqueeze(-1)))
        return torch.cat(outs, dim=1).squeeze(-1)
