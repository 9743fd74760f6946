"""The per-head part of GroupFree3D's loss as three HIP launches (csrc/gf_loss.hip).

`loss_helper.get_loss` (reference: detection/GroupFree3D/models/loss_helper.py:278-319) adds, for
the proposal head and every decoder layer's head, an objectness focal loss and box / semantic
losses against shared targets (`compute_objectness_loss_based_on_query_points` :81-137,
`compute_box_and_sem_cls_loss` :140-275).  Evaluated with torch ops on the stacked heads that is
~150 launches forward and ~200 backward; here one launch computes every (head, scene, query
point) row's seven terms AND the gradient w.r.t. the raw head outputs, which PredictHead leaves
in `end_points[prefix + '_head_output']` ((B, C, P), the concatenated 1x1-conv output).

`BTR_FUSED_GF_LOSS=0` keeps the op-by-op composition, which is the definition this path is
tested against (tests/test_gf_loss_gpu.py) and which the reference golden pins.  The per-term
entries of `end_points` are detached views of one statistics tensor; only the weighted total
carries the autograd graph (the reference's loop only calls `loss.backward()`)."""
import ctypes
import os

import torch
from torch.autograd import Function

from ..pointnet2 import _ext

_lib, _p = _ext._lib, _ext._p
HEAD_KEY = '_head_output'
_LABELS = (('point_obj_mask', torch.int64), ('point_instance_label', torch.int64),
           ('center_label', torch.float32), ('heading_class_label', torch.int64),
           ('heading_residual_label', torch.float32), ('size_class_label', torch.int64),
           ('size_residual_label', torch.float32), ('sem_cls_label', torch.int64))
TERMS = ('objectness_loss', 'center_loss', 'heading_cls_loss', 'heading_reg_loss',
         'size_cls_loss', 'size_reg_loss', 'box_loss', 'sem_cls_loss')


def enabled():
    return os.environ.get("BTR_FUSED_GF_LOSS", "1") != "0"


def can_fuse(end_points, config, prefixes, kinds):
    """All heads' raw outputs on the GPU, smooth-L1 forms, the loader's label dtypes."""
    if not enabled() or any(k != 'smoothl1' for k in kinds) or len(prefixes) > 8:
        return False
    heads = [end_points.get(p + HEAD_KEY) for p in prefixes]
    if any(h is None or not h.is_cuda or h.dtype != torch.float32 or h.dim() != 3 for h in heads):
        return False
    c = 4 + 2 * config.num_heading_bin + 4 * config.num_size_cluster + config.num_class
    if any(tuple(h.shape) != tuple(heads[0].shape) for h in heads) or heads[0].shape[1] != c or \
            c > 192:
        return False
    if end_points['center_label'].shape[2] != 3:
        return False
    # one centre base for all heads (the kernel takes a single base_xyz: every head of the
    # reference detector predicts offsets from cluster_xyz, detector.py:176-219; a variant that
    # refines the base per layer goes through the op-by-op composition), and the raw outputs
    # must still be what the published entries are views of
    base0 = end_points.get(prefixes[0] + 'base_xyz')
    for p, h in zip(prefixes, heads):
        if end_points.get(p + 'base_xyz') is not base0:
            return False
        tied = getattr(h, '_btr_head_views', None)
        if tied is None or tied[0] != h._version or tied[1] is not base0 or \
                any(end_points.get(p + k) is not v for k, v in tied[2].items()):
            return False
    return all(k in end_points and end_points[k].is_cuda and end_points[k].dtype == dt
               for k, dt in _LABELS)


class FusedHeadsLoss(Function):
    """(dims, base_xyz, seed_inds, sample_inds, mean_size, *labels, *heads) ->
    (weighted total, stats, objectness_label, object_assignment)."""

    @staticmethod
    def forward(ctx, dims, base_xyz, seed_inds, sample_inds, mean_size, *rest):
        nh, ns, nc, w_obj, w_box, w_sem, deltas = dims
        labels, heads = rest[:len(_LABELS)], rest[len(_LABELS):]
        H = len(heads)
        B, C, P = heads[0].shape
        dev = heads[0].device
        heads = [h.contiguous() for h in heads]
        labels = [t.contiguous() for t in labels]
        seed_inds = seed_inds.int().contiguous()
        sample_inds = sample_inds.int().contiguous()
        base_xyz = base_xyz.contiguous()
        d = _ext.GfLoss()
        d.b, d.p, d.k2, d.nh, d.ns, d.nc, d.heads, d.c = B, P, labels[2].shape[1], nh, ns, nc, H, C
        d.s1, d.n = seed_inds.shape[1], labels[0].shape[1]
        d.w_obj, d.w_box, d.w_sem = w_obj, w_box, w_sem
        d.center_delta, d.heading_delta, d.size_delta = deltas
        label = torch.empty((B, P), dtype=torch.int64, device=dev)
        assign = torch.empty((B, P), dtype=torch.int64, device=dev)
        npos = torch.empty((B,), dtype=torch.float32, device=dev)
        part = torch.empty((_lib.btr_gf_loss_part_floats(B, P, H),), dtype=torch.float32,
                           device=dev)
        stats = torch.empty((8 * H + 7,), dtype=torch.float32, device=dev)
        grads = torch.empty((H, B, C, P), dtype=torch.float32, device=dev)
        ptrs = (ctypes.c_void_p * H)(*[h.data_ptr() for h in heads])
        with _ext._on(heads[0]) as dv:
            _ext._call(_lib.btr_gf_loss_fwd, ctypes.addressof(d), ctypes.addressof(ptrs),
                       _p(base_xyz), _p(seed_inds), _p(sample_inds), *[_p(t) for t in labels],
                       _p(mean_size), _p(label), _p(assign), _p(npos), _p(part), _p(stats),
                       _p(grads), _ext._stream(dv))
        ctx.n_static = 5 + len(_LABELS)
        ctx.save_for_backward(grads)
        ctx.mark_non_differentiable(stats, label, assign)
        # (no zero-filled gradients for the statistics / label outputs: four fill launches)
        ctx.set_materialize_grads(False)
        # the total: a 0-dim tensor on the statistics vector's storage -- no copy launch, and a
        # base tensor (not an autograd view of `stats`), so `loss *= w` style code still works
        loss = torch.empty((), dtype=torch.float32, device=stats.device).set_(
            stats.untyped_storage(), stats.storage_offset() + 8 * H + 6, (), ())
        return loss, stats, label, assign

    @staticmethod
    def backward(ctx, gtotal, *_unused):
        grads, = ctx.saved_tensors
        if gtotal is None:
            return (None,) * (ctx.n_static + grads.shape[0])
        g = grads * gtotal.to(torch.float32)
        return (None,) * ctx.n_static + tuple(g.unbind(0))


_WEAK_LABELS = (('center_label', torch.float32), ('size_class_label', torch.int64),
                ('sem_cls_label', torch.int64))


def can_fuse_weak(end_points, config, prefixes, center_kind):
    """can_fuse for the weakly supervised loss: the centre, size-class and semantic labels only."""
    if not enabled() or center_kind != 'smoothl1' or len(prefixes) > 8:
        return False
    heads = [end_points.get(p + HEAD_KEY) for p in prefixes]
    if any(h is None or not h.is_cuda or h.dtype != torch.float32 or h.dim() != 3 for h in heads):
        return False
    c = 4 + 2 * config.num_heading_bin + 4 * config.num_size_cluster + config.num_class
    if any(tuple(h.shape) != tuple(heads[0].shape) for h in heads) or heads[0].shape[1] != c or \
            c > 192 or end_points['center_label'].shape[2] != 3:
        return False
    base0 = end_points.get(prefixes[0] + 'base_xyz')
    for p, h in zip(prefixes, heads):
        if end_points.get(p + 'base_xyz') is not base0:
            return False
        tied = getattr(h, '_btr_head_views', None)
        if tied is None or tied[0] != h._version or tied[1] is not base0 or \
                any(end_points.get(p + k) is not v for k, v in tied[2].items()) or \
                end_points.get(p + 'center') is None:
            return False
    return all(k in end_points and end_points[k].is_cuda and end_points[k].dtype == dt
               for k, dt in _WEAK_LABELS)


class FusedWeakHeadsLoss(Function):
    """(dims, base_xyz, label, assignment, mean_size, center_label, size_class_label,
    sem_cls_label, *heads) -> (weighted total, stats): the weakly supervised per-head loss
    (csrc/gf_loss.hip, btr_gf_loss_weak_fwd) and its gradient w.r.t. the raw head outputs."""

    @staticmethod
    def forward(ctx, dims, base_xyz, label, assign, mean_size, center_label, size_class_label,
                sem_cls_label, *heads):
        nh, ns, nc, w_obj, w_box, w_sem, center_delta = dims
        H = len(heads)
        B, C, P = heads[0].shape
        dev = heads[0].device
        heads = [h.contiguous() for h in heads]
        d = _ext.GfLoss()
        d.b, d.p, d.k2, d.nh, d.ns, d.nc, d.heads, d.c = B, P, center_label.shape[1], nh, ns, nc, H, C
        d.s1 = d.n = 1
        d.w_obj, d.w_box, d.w_sem = w_obj, w_box, w_sem
        d.center_delta, d.heading_delta, d.size_delta = center_delta, 1.0, 1.0
        npos = torch.empty((B,), dtype=torch.float32, device=dev)
        part = torch.empty((_lib.btr_gf_loss_part_floats(B, P, H),), dtype=torch.float32,
                           device=dev)
        stats = torch.empty((8 * H + 7,), dtype=torch.float32, device=dev)
        grads = torch.empty((H, B, C, P), dtype=torch.float32, device=dev)
        ptrs = (ctypes.c_void_p * H)(*[h.data_ptr() for h in heads])
        with _ext._on(heads[0]) as dv:
            _ext._call(_lib.btr_gf_loss_weak_fwd, ctypes.addressof(d), ctypes.addressof(ptrs),
                       _p(base_xyz.contiguous()), _p(label.contiguous()), _p(assign.contiguous()),
                       _p(center_label.contiguous()), _p(size_class_label.contiguous()),
                       _p(sem_cls_label.contiguous()), _p(mean_size), _p(npos), _p(part),
                       _p(stats), _p(grads), _ext._stream(dv))
        ctx.n_static = 8
        ctx.save_for_backward(grads)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        loss = torch.empty((), dtype=torch.float32, device=stats.device).set_(
            stats.untyped_storage(), stats.storage_offset() + 8 * H + 6, (), ())
        return loss, stats

    @staticmethod
    def backward(ctx, gtotal, *_unused):
        grads, = ctx.saved_tensors
        if gtotal is None:
            return (None,) * (ctx.n_static + grads.shape[0])
        g = grads * gtotal.to(torch.float32)
        return (None,) * ctx.n_static + tuple(g.unbind(0))


def weak_heads_loss(end_points, config, prefixes, coefs, center_delta, mean_size, label,
                    assignment, weights):
    """Fills the per-head entries of `end_points` like compute_objectness_loss_based_on_query_
    points_weak + compute_center_and_sem_cls_loss and returns 10 / (L + 1) * (obj_coef * sum
    objectness + box_coef * sum box + sem_coef * sum sem)."""
    H = len(prefixes)
    scale = 10.0 / H
    dims = (config.num_heading_bin, config.num_size_cluster, config.num_class,
            scale * coefs[0], scale * coefs[1], scale * coefs[2], float(center_delta))
    heads = [end_points[p + HEAD_KEY] for p in prefixes]
    total, stats = FusedWeakHeadsLoss.apply(
        dims, end_points[prefixes[0] + 'base_xyz'], label, assignment, mean_size,
        *[end_points[k] for k, _ in _WEAK_LABELS], *heads)
    for h, prefix in enumerate(prefixes):
        end_points[prefix + 'objectness_label'] = label
        end_points[prefix + 'objectness_mask'] = weights
        end_points[prefix + 'object_assignment'] = assignment
        for j, name in ((0, 'objectness_loss'), (1, 'center_loss'), (4, 'size_cls_loss'),
                        (6, 'box_loss'), (7, 'sem_cls_loss')):
            end_points[prefix + name] = stats[8 * h + j]
    end_points['sum_heads_objectness_loss'] = stats[8 * H]
    end_points['sum_heads_box_loss'] = stats[8 * H + 1]
    end_points['sum_heads_sem_cls_loss'] = stats[8 * H + 2]
    return total


def heads_loss(end_points, config, prefixes, coefs, deltas, mean_size):
    """Fills the per-head entries of `end_points` like the two reference functions and returns
    10 / (L + 1) * (obj_coef * sum objectness + box_coef * sum box + sem_coef * sum sem)."""
    H = len(prefixes)
    scale = 10.0 / H
    dims = (config.num_heading_bin, config.num_size_cluster, config.num_class,
            scale * coefs[0], scale * coefs[1], scale * coefs[2], tuple(float(x) for x in deltas))
    heads = [end_points[p + HEAD_KEY] for p in prefixes]
    total, stats, label, assign = FusedHeadsLoss.apply(
        dims, end_points[prefixes[0] + 'base_xyz'], end_points['seed_inds'],
        end_points['query_points_sample_inds'], mean_size,
        *[end_points[k] for k, _ in _LABELS], *heads)
    B, P = label.shape
    weights = torch.full((B, P), 1.0 / P, device=label.device)
    for h, prefix in enumerate(prefixes):
        end_points[prefix + 'objectness_label'] = label
        end_points[prefix + 'objectness_mask'] = weights
        end_points[prefix + 'object_assignment'] = assign
        end_points[prefix + 'pos_ratio'] = stats[8 * H + 4]
        end_points[prefix + 'neg_ratio'] = stats[8 * H + 5]
        for j, name in enumerate(TERMS):
            end_points[prefix + name] = stats[8 * h + j]
    end_points['sum_heads_objectness_loss'] = stats[8 * H]
    end_points['sum_heads_box_loss'] = stats[8 * H + 1]
    end_points['sum_heads_sem_cls_loss'] = stats[8 * H + 2]
    return total


class FusedFocalSum(Function):
    """scale * sum(sigmoid_focal_loss(logits, label) * w) per group as one launch each way
    (csrc/gf_loss.hip focal_sum_kernel): logits (groups, n) f32, label (n,) i64 in {0, 1} shared
    by the groups -> (groups,) sums."""

    @staticmethod
    def forward(ctx, logits, label, w, scale, gamma, alpha):
        groups = logits.shape[0]
        x = logits.contiguous().view(groups, -1)
        label = label.contiguous().view(-1)
        n = x.shape[1]
        assert label.numel() == n
        out = torch.empty((groups,), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        with _ext._on(x) as d:
            _ext._call(_lib.btr_focal_sum, groups, n, n, _ext._p(x), _ext._p(label), float(w),
                       float(scale), float(gamma), float(alpha), _ext._p(out), _ext._p(grad),
                       _ext._stream(d))
        ctx.shape = logits.shape
        ctx.save_for_backward(grad)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            return (None,) * 6
        (grad,) = ctx.saved_tensors
        return (grad * gout.view(-1, 1)).view(ctx.shape), None, None, None, None, None


def focal_sum_fusable(logits, label):
    return enabled() and logits.is_cuda and logits.dtype == torch.float32 and \
        label.dtype == torch.int64 and label.numel() > 0 and \
        logits.numel() % label.numel() == 0


def focal_sum(logits, label, w, scale, gamma=2.0, alpha=0.25):
    """logits (groups, ...) with label.numel() elements per group -> (groups,) sums."""
    return FusedFocalSum.apply(logits, label, w, scale, gamma, alpha)
