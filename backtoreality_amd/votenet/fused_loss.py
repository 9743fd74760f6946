"""The VoteNet loss as three HIP kernels (csrc/votenet_loss.hip) instead of ~250 torch launches.

`loss_helper.get_loss` (reference: detection/Votenet/models/loss_helper.py:336-400) dispatches
here when everything it needs is on the GPU; `BTR_FUSED_LOSS=0` keeps the op-by-op torch
composition, which is the definition this path is tested against (tests/test_fused_loss_gpu.py)
and which is itself pinned by the reference golden (tests/golden/votenet_fsb_step.npz).

Differences visible to a caller: the per-term entries of `end_points` ('vote_loss', ...) are
detached views of one statistics tensor -- only `loss` carries the autograd graph (the
reference's training loop only ever calls `loss.backward()`, train.py:263).
"""
import ctypes
import os

import torch
from torch.autograd import Function

from ..pointnet2 import _ext

_lib = _ext._lib

STAT_KEYS = ('loss', 'vote_loss', 'objectness_loss', 'center_loss', 'heading_cls_loss',
             'heading_reg_loss', 'size_cls_loss', 'size_reg_loss', 'sem_cls_loss', 'box_loss',
             'pos_ratio', 'neg_ratio', 'obj_acc')
_LABEL_KEYS = ('vote_label', 'vote_label_mask', 'center_label', 'box_label_mask',
               'heading_class_label', 'heading_residual_label', 'size_class_label',
               'size_residual_label', 'sem_cls_label')
_LABEL_DTYPES = (torch.float32, torch.int64, torch.float32, torch.float32, torch.int64,
                 torch.float32, torch.int64, torch.float32, torch.int64)
HEAD_KEY = '_head_output'  # raw (B, Cout, K) proposal-head output, stored by ProposalModule
# weights of (vote, objectness, center, heading_cls, heading_reg, size_cls, size_reg, sem_cls)
# in loss / 10
W_FSB = (1.0, 0.5, 1.0, 0.1, 1.0, 0.1, 1.0, 0.1)             # get_loss, loss_helper.py:375-385
W_DA_SOURCE = tuple(0.1 * w for w in W_FSB)                   # get_loss_DA :575-600 (x 0.1)
W_DA_TARGET = (1.0, 0.5, 1.0, 0.0, 0.0, 0.1, 0.0, 0.1)        # weak labels: no heading / size reg


def enabled():
    return os.environ.get("BTR_FUSED_LOSS", "1") != "0"


def can_fuse(end_points, config):
    """The fused kernels cover vote_factor 1, K <= 1024 proposals, <= 256 GT slots, all CUDA."""
    if not enabled() or HEAD_KEY not in end_points:
        return False
    net = end_points[HEAD_KEY]
    if not (net.is_cuda and net.dtype == torch.float32):
        return False
    if end_points['vote_xyz'].shape[1] != end_points['seed_xyz'].shape[1]:
        return False
    if net.shape[2] > 1024 or end_points['center_label'].shape[1] > 256:
        return False
    cout = 5 + 2 * config.num_heading_bin + 4 * config.num_size_cluster + config.num_class
    if net.shape[1] != cout or end_points['center_label'].shape[2] != 3:
        return False
    return all(end_points[k].is_cuda and end_points[k].dtype == d
               for k, d in zip(_LABEL_KEYS, _LABEL_DTYPES))


class FusedVoteNetLoss(Function):
    @staticmethod
    def forward(ctx, net, agg_xyz, vote_xyz, seed_xyz, seed_inds, mean_size, dims, *labels):
        nh, ns, nc, weights, vote_mode = dims
        B, cout, K = net.shape
        K2, S1, N = labels[2].shape[1], seed_xyz.shape[1], labels[0].shape[1]
        dev = net.device
        if seed_inds.dtype != torch.int32:
            seed_inds = seed_inds.int()
        net, agg_xyz, vote_xyz, seed_xyz, seed_inds = (
            t.contiguous() for t in (net, agg_xyz, vote_xyz, seed_xyz, seed_inds))
        labels = tuple(t.contiguous() for t in labels)
        objectness_label = torch.empty((B, K), dtype=torch.int64, device=dev)
        objectness_mask = torch.empty((B, K), dtype=torch.float32, device=dev)
        object_assignment = torch.empty((B, K), dtype=torch.int64, device=dev)
        j1c = torch.empty((B, K), dtype=torch.int32, device=dev)
        k2c = torch.empty((B, K2), dtype=torch.int32, device=dev)
        vote_arg = torch.empty((B, S1), dtype=torch.int8, device=dev)
        part = torch.empty((B, 16), dtype=torch.float32, device=dev)
        stats = torch.empty((14,), dtype=torch.float32, device=dev)
        norm = torch.empty((4,), dtype=torch.float32, device=dev)
        i2v = torch.empty((B, K2), dtype=torch.int32, device=dev) if vote_mode else None
        w8 = (ctypes.c_float * 8)(*weights)
        p = _ext._p
        with _ext._on(net) as d:
            _ext._call(_lib.btr_votenet_loss_fwd, B, K, K2, nh, ns, nc, S1, N, cout, p(net),
                       p(agg_xyz), p(vote_xyz), p(seed_xyz), p(seed_inds),
                       *[p(t) for t in labels], p(mean_size), p(objectness_label),
                       p(objectness_mask), p(object_assignment), p(j1c), p(k2c), p(vote_arg),
                       p(part), p(stats), p(norm), w8, int(vote_mode), p(i2v), _ext._stream(d))
        ctx.dims = (B, K, K2, nh, ns, nc, S1, N, cout)
        ctx.weights, ctx.vote_mode, ctx.i2v = tuple(weights), int(vote_mode), i2v
        ctx.save_for_backward(net, agg_xyz, vote_xyz, seed_xyz, seed_inds, mean_size, norm,
                              objectness_label, objectness_mask, object_assignment, j1c, k2c,
                              vote_arg, *labels)
        # (a word of the statistics vector, not a clone: one copy launch per step less.
        # NOT an autograd view of `stats` either -- a view created inside a Function refuses
        # every in-place op (`loss *= w`); a 0-dim tensor set on the same storage is a base
        # tensor autograd treats like any other output.  It is the vector's LAST word, a second
        # copy of the total the kernel writes for this purpose: autograd does not know the two
        # alias, so an in-place edit of the loss must not reach the reported stats[0..12].
        # `stats` itself is returned non-differentiable below)
        loss = torch.empty((), dtype=torch.float32, device=dev).set_(
            stats.untyped_storage(), stats.storage_offset() + 13, (), ())
        ctx.mark_non_differentiable(stats, objectness_label, objectness_mask, object_assignment)
        # (no zero-filled gradients for the statistics / label outputs: four fill launches)
        ctx.set_materialize_grads(False)
        return loss, stats, objectness_label, objectness_mask, object_assignment

    @staticmethod
    def backward(ctx, gloss, *_unused):
        (net, agg_xyz, vote_xyz, seed_xyz, seed_inds, mean_size, norm, objectness_label,
         objectness_mask, object_assignment, j1c, k2c, vote_arg, *labels) = ctx.saved_tensors
        B, K, K2, nh, ns, nc, S1, N, cout = ctx.dims
        if gloss is None:
            return (None,) * (7 + len(labels))
        gout = gloss.reshape(1).to(torch.float32).contiguous()
        dnet = torch.empty_like(net)
        dagg = torch.empty_like(agg_xyz)
        dvote = torch.empty_like(vote_xyz)
        p = _ext._p
        with _ext._on(net) as d:
            _ext._call(_lib.btr_votenet_loss_bwd, B, K, K2, nh, ns, nc, S1, N, cout, p(gout),
                       p(norm), p(net), p(agg_xyz), p(vote_xyz), p(seed_xyz), p(seed_inds),
                       *[p(t) for t in labels], p(mean_size), p(objectness_label),
                       p(objectness_mask), p(object_assignment), p(j1c), p(k2c), p(vote_arg),
                       p(dnet), p(dagg), p(dvote), (ctypes.c_float * 8)(*ctx.weights),
                       ctx.vote_mode, p(ctx.i2v), _ext._stream(d))
        return (dnet, dagg, dvote, None, None, None, None) + (None,) * len(labels)


def _mean_size(config, dev):
    cache = getattr(config, "_mean_size_dev", None)
    if cache is None or cache.device != dev:
        import numpy as np
        cache = torch.from_numpy(np.ascontiguousarray(config.mean_size_arr, np.float32)).to(dev)
        try:
            config._mean_size_dev = cache
        except AttributeError:
            pass
    return cache


def _apply(end_points, config, weights, vote_mode):
    net = end_points[HEAD_KEY]
    dims = (config.num_heading_bin, config.num_size_cluster, config.num_class, weights,
            vote_mode)
    return FusedVoteNetLoss.apply(
        net, end_points['aggregated_vote_xyz'], end_points['vote_xyz'], end_points['seed_xyz'],
        end_points['seed_inds'], _mean_size(config, net.device), dims,
        *[end_points[k] for k in _LABEL_KEYS])


def get_loss_branch(end_points, config, weights, keys):
    """One branch of get_loss_DA: the weighted sum (x 10) of the branch's terms with the weak
    vote loss; fills `keys` (a subset of STAT_KEYS) plus the objectness bookkeeping."""
    loss, stats, label, mask, assignment = _apply(end_points, config, weights, 1)
    for i, k in enumerate(STAT_KEYS):
        if k in keys:
            end_points[k] = stats[i]
    end_points['objectness_label'] = label
    end_points['objectness_mask'] = mask
    end_points['object_assignment'] = assignment
    return loss


def get_loss(end_points, config):
    """Same contract as loss_helper.get_loss: returns (loss, end_points) with every term,
    'objectness_label', 'objectness_mask' and 'object_assignment' filled in."""
    net = end_points[HEAD_KEY]
    dims = (config.num_heading_bin, config.num_size_cluster, config.num_class, W_FSB, 0)
    loss, stats, label, mask, assignment = FusedVoteNetLoss.apply(
        net, end_points['aggregated_vote_xyz'], end_points['vote_xyz'], end_points['seed_xyz'],
        end_points['seed_inds'], _mean_size(config, net.device), dims,
        *[end_points[k] for k in _LABEL_KEYS])
    for i, k in enumerate(STAT_KEYS):
        end_points[k] = stats[i]
    end_points['loss'] = loss
    end_points['objectness_label'] = label
    end_points['objectness_mask'] = mask
    end_points['object_assignment'] = assignment
    return loss, end_points


class FusedDomainLoss(Function):
    """The domain-adaptation term of get_loss_DA (loss_helper._domain_loss) as one launch each
    way: (global_S (B,2), local_S (B,1,K), label_S (B,K) i64, global_T, local_T, label_T) ->
    the scalar coef * (mean(l_S^2 w_S) + focal(g_S, 0) + mean((1 - l_T)^2 w_T) + focal(g_T, 1))."""

    @staticmethod
    def forward(ctx, g_S, l_S, w_S, g_T, l_T, w_T, gamma, coef):
        B, K = w_S.shape
        dev = g_S.device
        g_S, l_S, g_T, l_T = (t.contiguous() for t in (g_S, l_S, g_T, l_T))
        w_S, w_T = w_S.contiguous(), w_T.contiguous()
        out = torch.empty((3,), dtype=torch.float32, device=dev)
        grads = torch.empty((4 * B + 2 * B * K,), dtype=torch.float32, device=dev)
        p = _ext._p
        with _ext._on(g_S) as d:
            _ext._call(_lib.btr_domain_loss, B, K, float(gamma), float(coef), p(g_S), p(l_S), p(w_S), p(g_T),
                       p(l_T), p(w_T), p(out), p(grads), _ext._stream(d))
        ctx.dims = (B, K, tuple(l_S.shape))
        ctx.save_for_backward(grads)
        ctx.set_materialize_grads(False)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            return (None,) * 8
        (grads,) = ctx.saved_tensors
        B, K, lshape = ctx.dims
        g = grads * gout          # (one launch: the four gradients are slices of one buffer)
        dgS, dlS, dgT, dlT = g.split([2 * B, B * K, 2 * B, B * K])
        return (dgS.view(B, 2), dlS.view(lshape), None, dgT.view(B, 2), dlT.view(lshape), None,
                None, None)


def domain_loss_fusable(end_points_S, end_points_T, prefix=''):
    """`prefix`: '' for VoteNet's end_points keys, 'last_' for GroupFree3D's last head."""
    if not enabled():
        return False
    for e in (end_points_S, end_points_T):
        g, l, w = e['global_d_pred'], e[prefix + 'local_d_pred'], e[prefix + 'objectness_label']
        if not (g.is_cuda and g.dtype == torch.float32 and g.dim() == 2 and g.shape[1] == 2 and
                l.dtype == torch.float32 and l.dim() == 3 and l.shape[1] == 1 and
                w.dtype == torch.int64 and w.dim() == 2 and
                l.shape[0] == g.shape[0] == w.shape[0] and l.shape[2] == w.shape[1]):
            return False
    S, T = end_points_S, end_points_T
    return S['global_d_pred'].shape == T['global_d_pred'].shape and \
        S[prefix + 'local_d_pred'].shape == T[prefix + 'local_d_pred'].shape


def domain_loss(end_points_S, end_points_T, gamma=3.0, coef=0.5, prefix=''):
    return FusedDomainLoss.apply(
        end_points_S['global_d_pred'], end_points_S[prefix + 'local_d_pred'],
        end_points_S[prefix + 'objectness_label'], end_points_T['global_d_pred'],
        end_points_T[prefix + 'local_d_pred'], end_points_T[prefix + 'objectness_label'], gamma,
        coef)
