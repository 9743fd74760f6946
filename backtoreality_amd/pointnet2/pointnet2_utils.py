"""Autograd-level op API of the reference's `pointnet2_utils`
(/root/reference/detection/Votenet/pointnet2/pointnet2_utils.py) over the MI355X `_ext`.

Same module-level names and call conventions: `furthest_point_sample(xyz, npoint)`,
`gather_operation(features, idx)`, `three_nn(unknown, known) -> (dist, idx)` (dist is the
square root of what `_ext.three_nn` returns, :140-142), `three_interpolate(features, idx,
weight)`, `grouping_operation(features, idx)`, `ball_query(radius, nsample, xyz, new_xyz)`
(Python order; `_ext.ball_query` takes (new_xyz, xyz, radius, nsample), :282), and the
`QueryAndGroup` / `GroupAll` modules.

`_ext` is a module-level name on purpose: like the reference (:25-33) everything goes through
it, and tests swap it for the CPU oracle adapter to exercise this host logic without a GPU.
The product never falls back: with the HIP `_ext`, CPU tensors raise "CPU not supported".
"""
import builtins
import os
import sys

import torch
import torch.nn as nn
from torch.autograd import Function

if __package__:
    from . import pytorch_utils as pt_utils
else:  # imported top-level after sys.path.append(.../pointnet2), as the reference's callers do
    import pytorch_utils as pt_utils

try:
    if __package__:
        from . import _ext
    else:
        _here = os.path.dirname(os.path.abspath(__file__))
        if os.path.dirname(_here) not in sys.path:
            sys.path.append(os.path.dirname(_here))
        import pointnet2._ext as _ext
except ImportError:
    # same escape hatch as the reference (:27-33): layers importable without the binary
    if not getattr(builtins, "__POINTNET2_SETUP__", False):
        raise
    _ext = None


class RandomDropout(nn.Module):
    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p = p
        self.inplace = inplace

    def forward(self, X):
        theta = torch.Tensor(1).uniform_(0, self.p)[0]
        return pt_utils.feature_dropout_no_scaling(X, theta, self.train, self.inplace)


class FurthestPointSampling(Function):
    """xyz (B,N,3) -> int32 (B,npoint) indices; not differentiable (:51-77)."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


def furthest_point_sample(xyz, npoint):
    """pointnet2_utils.py:80 (`furthest_point_sample = FurthestPointSampling.apply`).  The result
    remembers which tensor it sampled: `gather_rows(xyz, inds)` then marks ITS result as
    FPS-ordered, and an FPS of that result (the next pyramid level) checks "0, 1, 2, ..." in
    parallel before it runs the serial kernel (_ext.mark_fps_ordered: a speed hint only)."""
    inds = FurthestPointSampling.apply(xyz, npoint)
    if xyz.is_cuda and hasattr(_ext, "mark_fps_ordered"):
        inds._btr_fps_of = (xyz, xyz._version, inds._version)
    return inds


class GatherOperation(Function):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint); backward scatter-adds (:83-114)."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.for_backwards = (idx, features.size(1), features.size(2))
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, _, N = ctx.for_backwards
        return _ext.gather_points_grad(grad_out.contiguous(), idx, N), None


gather_operation = GatherOperation.apply


class GatherRows(Function):
    """rows `idx (B,M)` of a channel-last `(B,N,C)` tensor -> `(B,M,C)`; one launch on the GPU
    instead of transpose + gather_operation + transpose, same values and gradient."""

    @staticmethod
    def forward(ctx, src, idx):
        ctx.save_for_backward(idx)
        ctx.n = src.size(1)
        return _ext.gather_rows(src.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        if hasattr(_ext, "gather_rows_grad"):
            return _ext.gather_rows_grad(grad_out.contiguous(), idx, ctx.n), None
        g = _ext.gather_points_grad(grad_out.transpose(1, 2).contiguous(), idx, ctx.n)
        return g.transpose(1, 2).contiguous(), None


def gather_rows(src, idx):
    """(B,N,C)[idx (B,M)] -> (B,M,C).  Extensions without `gather_rows` (the CPU oracle
    adapter) take the reference's transpose + gather + transpose route."""
    if hasattr(_ext, "gather_rows") and src.is_cuda:
        out = GatherRows.apply(src, idx)
        tag = getattr(idx, "_btr_fps_of", None)
        if tag is not None and tag[0] is src and tag[1] == src._version and \
                tag[2] == idx._version and src.size(-1) == 3 and hasattr(_ext, "mark_fps_ordered"):
            _ext.mark_fps_ordered(out)     # the points an FPS sampled, in sampling order
        return out
    return gather_operation(src.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()


class ThreeNN(Function):
    """unknown (B,n,3), known (B,m,3) -> (l2 dist (B,n,3), idx (B,n,3)); no gradient (:120-146)."""

    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeNNWeights(Function):
    """three_nn + the blend weights PointnetFPModule derives from it
    (pointnet2_modules.py:492-496: 1 / (dist + 1e-8), normalised over the three neighbours) in
    one launch: unknown (B,n,3), known (B,m,3) -> (idx (B,n,3), weight (B,n,3)); no gradient,
    like three_nn."""

    @staticmethod
    def forward(ctx, unknown, known):
        _, idx, weight = _ext.three_nn_weights(unknown, known)
        ctx.mark_non_differentiable(idx, weight)
        return idx, weight

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn_weights = ThreeNNWeights.apply


class ThreeInterpolate(Function):
    """features (B,c,m), idx/weight (B,n,3) -> (B,c,n) weighted blend (:152-203)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.three_interpolate_for_backward = (idx, weight, features.size(2))
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B,C,N), idx (B,npoint,nsample) -> fresh (B,C,npoint,nsample) tensor (callers
    modify it in place, :350-352); backward scatter-adds into (B,C,N) (:209-254)."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.for_backwards = (idx, features.size(2))
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        return _ext.group_points_grad(grad_out.contiguous(), idx, N), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3)) -> int32 (B,npoint,nsample) (:260-288)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping: returns (B, 3+C, npoint, nsample) features whose first three
    channels are the neighbours' offsets from their centre, optionally divided by the radius
    (:294-376)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False,
                 normalize_xyz=False, sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def _resample_uniformly(self, idx):
        """:336-345 on the device, one pass of tensor ops instead of the reference's host
        double loop (B * npoint iterations of torch.unique + torch.randint + a host sync each).
        A ball-query row IS its distinct hits in ascending order followed by copies of the
        first one (ball_query_gpu.cu:39-43), so `torch.unique(row)` = its first k entries with
        k = 1 + #(entries != the first) -- no sort needed; the padding slots are then redrawn
        uniformly from those k.  Same distribution as the reference; the draws come from the
        device generator, so they are not the reference's CPU-generator draws (checked against
        the reference loop with the draws injected: tests/test_host_logic.py).
        Returns unique_cnt (B, npoint) float32 like the reference."""
        B, M, S = idx.shape
        first = idx[:, :, :1]
        k = (idx != first).sum(-1, keepdim=True) + 1                     # (B, M, 1)
        slot = torch.arange(S, device=idx.device).view(1, 1, S)
        draw = self._uniform_draws((B, M, S), idx.device)                # in [0, 1)
        src = torch.minimum((draw * k).long(), k - 1)                    # uniform in [0, k)
        src = torch.where(slot < k, slot.expand(B, M, S), src)           # keep the k hits
        idx.copy_(torch.gather(idx, 2, src))
        return k.squeeze(-1).to(torch.float32)

    @staticmethod
    def _uniform_draws(shape, device):
        return torch.rand(shape, device=device)

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = self._resample_uniformly(idx) if self.sample_uniformly else None

        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius

        if features is not None:
            grouped = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz

        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """Single group holding every point: (B, 3+C, 1, N) (:379-426)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            new_features = grouped_xyz
        else:
            grouped = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
