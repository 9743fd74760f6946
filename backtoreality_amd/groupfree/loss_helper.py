"""GroupFree3D training loss (detection/GroupFree3D/models/loss_helper.py:17-319,
models/losses.py): query-point generation (sigmoid focal loss on the k seeds nearest to each
box centre), and per prediction head (proposal + every decoder layer) the objectness focal
loss, box losses and semantic cross-entropy with the labels of the object the query point
lies in.  Labels beyond VoteNet's: 'size_gts' (B,K2,3), 'point_obj_mask' (B,N) i64,
'point_instance_label' (B,N) i64 (-1 = background)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import fused_loss


def _mean_size(config, dev):
    cache = getattr(config, "_mean_size_dev", None)
    if cache is None or cache.device != dev:
        cache = torch.from_numpy(np.ascontiguousarray(config.mean_size_arr, np.float32)).to(dev)
        try:
            config._mean_size_dev = cache
        except AttributeError:
            pass
    return cache


def smoothl1_loss(error, delta=1.0):
    """0.5 x^2 / d for |x| < d, |x| - 0.5 d beyond (losses.py:5-13)."""
    diff = torch.abs(error)
    return torch.where(diff < delta, 0.5 * diff * diff / delta, diff - 0.5 * delta)


def l1_loss(error):
    return torch.abs(error)


def sigmoid_focal_loss(logits, target, weights, gamma=2.0, alpha=0.25):
    """SigmoidFocalClassificationLoss.forward (losses.py:21-81): logits / target (B,P,C),
    weights (B,P) -> weighted focal loss (B,P,C)."""
    p = torch.sigmoid(logits)
    alpha_w = target * alpha + (1 - target) * (1 - alpha)
    pt = target * (1.0 - p) + (1.0 - target) * p
    bce = torch.clamp(logits, min=0) - logits * target + torch.log1p(torch.exp(-torch.abs(logits)))
    return alpha_w * torch.pow(pt, gamma) * bce * weights.unsqueeze(-1)


def head_prefixes(num_decoder_layers):
    if num_decoder_layers > 0:
        return ['proposal_', 'last_'] + ['%dhead_' % i for i in range(num_decoder_layers - 1)]
    return ['proposal_']


def compute_points_obj_cls_loss_hard_topk(end_points, topk):
    """loss_helper.py:17-78: a seed is positive when it belongs to an object and is among the
    `topk` seeds of that object closest (in box-size units) to the object's centre."""
    box_label_mask = end_points['box_label_mask']
    seed_inds = end_points['seed_inds'].long()
    seed_xyz = end_points['seed_xyz']
    logits = end_points['seeds_obj_cls_logits']
    gt_center = end_points['center_label'][:, :, 0:3]
    gt_size = end_points['size_gts'][:, :, 0:3]
    B, K, K2 = gt_center.shape[0], seed_xyz.shape[1], gt_center.shape[1]
    dev = seed_xyz.device

    point_instance_label = end_points['point_instance_label']
    seed_instance = torch.gather(point_instance_label, 1, seed_inds)              # (B,K)
    assignment = torch.where(seed_instance < 0, torch.full_like(seed_instance, K2 - 1),
                             seed_instance)
    one_hot = torch.zeros((B, K, K2), device=dev)
    one_hot.scatter_(2, assignment.unsqueeze(-1), 1)
    delta = (seed_xyz.unsqueeze(2) - gt_center.unsqueeze(1)) / (gt_size.unsqueeze(1) + 1e-6)
    dist = torch.sqrt(torch.sum(delta ** 2, dim=-1) + 1e-6)                         # (B,K,K2)
    dist = (dist * one_hot + 100 * (1 - one_hot)).transpose(1, 2).contiguous()      # (B,K2,K)
    topk_inds = torch.topk(dist, topk, largest=False)[1] * box_label_mask[:, :, None] + \
        (box_label_mask[:, :, None] - 1)          # padded boxes -> index -1 (the spare column)
    topk_inds = topk_inds.long().view(B, -1)

    label = torch.zeros((B, K + 1), dtype=torch.long, device=dev)
    label.scatter_(1, torch.where(topk_inds < 0, torch.full_like(topk_inds, K), topk_inds), 1)
    label = label[:, :K]
    label = torch.where(seed_instance < 0, torch.zeros_like(label), label)

    total = B * K
    end_points['points_hard_topk%d_pos_ratio' % topk] = torch.sum(label.float()) / float(total)
    end_points['points_hard_topk%d_neg_ratio' % topk] = \
        1 - end_points['points_hard_topk%d_pos_ratio' % topk]

    if fused_loss.focal_sum_fusable(logits, label) and K > 0:
        # every label is 0 or 1, so the weights (label >= 0) / count are 1 / K for every point:
        # the focal terms, their sum and its gradient as one launch each way
        return fused_loss.focal_sum(logits.reshape(1, B * K), label, 1.0 / K, 1.0 / B)[0]
    weights = (label >= 0).float()
    weights = weights / torch.clamp(weights.sum(dim=1, keepdim=True), min=1.0)
    loss = sigmoid_focal_loss(logits.view(B, K, 1), label.unsqueeze(-1).float(), weights)
    return loss.sum() / B


def _stack(end_points, prefixes, key):
    """(H, ...) tensor of one head output over all prediction heads: the seven heads share
    their targets, so every loss term is evaluated once on the stacked tensor instead of once
    per head (7x fewer launches; the per-head scalars are views of one (H,) vector)."""
    return torch.stack([end_points[p + key] for p in prefixes], 0)


def compute_objectness_loss_based_on_query_points(end_points, num_decoder_layers):
    """loss_helper.py:81-137: objectness target of a query point = whether its seed point lies
    in an object; its box target = that object (background -> the last ground-truth slot)."""
    seed_inds = end_points['seed_inds'].long()
    sample_inds = end_points['query_points_sample_inds'].long()
    K2 = end_points['center_label'].shape[1]
    B, K = sample_inds.shape
    obj_gt = torch.gather(torch.gather(end_points['point_obj_mask'], 1, seed_inds), 1, sample_inds)
    instance = torch.gather(torch.gather(end_points['point_instance_label'], 1, seed_inds), 1,
                            sample_inds)
    assignment = torch.where(instance < 0, torch.full_like(instance, K2 - 1), instance)
    mask = torch.ones((B, K), device=seed_inds.device)
    total = float(B * K)
    weights = mask / torch.clamp(mask.sum(dim=1, keepdim=True), min=1.0)
    pos_ratio = torch.sum(obj_gt.float()) / total
    neg_ratio = torch.sum(mask) / total - pos_ratio

    prefixes = head_prefixes(num_decoder_layers)
    scores = _stack(end_points, prefixes, 'objectness_scores')            # (H, B, K, 1)
    # (the reference flattens the TRANSPOSED (B,1,K) tensor: the same memory order)
    loss = sigmoid_focal_loss(scores.reshape(-1, K, 1),
                              obj_gt.unsqueeze(-1).float().repeat(len(prefixes), 1, 1),
                              weights.repeat(len(prefixes), 1))
    loss = loss.view(len(prefixes), -1).sum(1) / B                         # (H,)
    for h, prefix in enumerate(prefixes):
        end_points[prefix + 'objectness_label'] = obj_gt
        # (the reference normalises its all-ones mask IN PLACE after storing it, :114-128: what
        # callers find under this key is 1/K per query point)
        end_points[prefix + 'objectness_mask'] = weights
        end_points[prefix + 'object_assignment'] = assignment
        end_points[prefix + 'pos_ratio'] = pos_ratio
        end_points[prefix + 'neg_ratio'] = neg_ratio
        end_points[prefix + 'objectness_loss'] = loss[h]
    return loss.sum(), end_points


def compute_box_and_sem_cls_loss(end_points, config, num_decoder_layers,
                                 center_loss_type='smoothl1', center_delta=1.0,
                                 size_loss_type='smoothl1', size_delta=1.0,
                                 heading_loss_type='smoothl1', heading_delta=1.0):
    """loss_helper.py:140-275: per head, masked by the objectness label and normalised by the
    number of positive query points."""
    nh, ns = config.num_heading_bin, config.num_size_cluster
    gt_center = end_points['center_label'][:, :, 0:3]
    dev = gt_center.device
    mean_size = _mean_size(config, dev)   # cached on the device: no per-step H2D copy
    prefixes = head_prefixes(num_decoder_layers)
    H = len(prefixes)

    # targets: identical for every head (set by compute_objectness_loss_based_on_query_points)
    assignment = end_points[prefixes[0] + 'object_assignment']
    label = end_points[prefixes[0] + 'objectness_label'].float()           # (B, K)
    npos = torch.sum(label) + 1e-6
    B, K = label.shape
    a3 = assignment.unsqueeze(2).expand(-1, -1, 3)
    per_head = lambda t: t.reshape(H, -1).sum(1) / npos                    # noqa: E731

    err = torch.gather(gt_center, 1, a3).unsqueeze(0) - _stack(end_points, prefixes, 'center')
    if center_loss_type == 'smoothl1':
        center_loss = smoothl1_loss(err, delta=center_delta)
    elif center_loss_type == 'l1':
        center_loss = l1_loss(err)
    else:
        raise NotImplementedError
    center_loss = per_head(center_loss * label.unsqueeze(2))

    def cls_loss(key, target, nclass):
        scores = _stack(end_points, prefixes, key).reshape(-1, nclass)     # (H*B*K, C)
        ce = F.cross_entropy(scores, target.reshape(-1).repeat(H), reduction='none')
        return per_head(ce.view(H, B, K) * label)

    def res_loss(err, kind, delta):
        if kind == 'smoothl1':
            return delta * smoothl1_loss(err, delta=delta)
        if kind == 'l1':
            return l1_loss(err)
        raise NotImplementedError

    hcls = torch.gather(end_points['heading_class_label'], 1, assignment)
    heading_class_loss = cls_loss('heading_scores', hcls, nh)
    hres = torch.gather(end_points['heading_residual_label'], 1, assignment) / (np.pi / nh)
    h_one_hot = F.one_hot(hcls, nh).float()
    h_err = torch.sum(_stack(end_points, prefixes, 'heading_residuals_normalized') * h_one_hot,
                      -1) - hres
    heading_reg = per_head(res_loss(h_err, heading_loss_type, heading_delta) * label)

    scls = torch.gather(end_points['size_class_label'], 1, assignment)
    size_class_loss = cls_loss('size_scores', scls, ns)
    sres = torch.gather(end_points['size_residual_label'], 1, a3)
    s_one_hot = F.one_hot(scls, ns).float().unsqueeze(-1).expand(-1, -1, -1, 3)
    pred_res = torch.sum(_stack(end_points, prefixes, 'size_residuals_normalized') * s_one_hot, 3)
    mean_label = torch.sum(s_one_hot * mean_size.unsqueeze(0).unsqueeze(0), 2)
    size_reg = per_head(res_loss(pred_res - sres / mean_label, size_loss_type, size_delta) *
                        label.unsqueeze(2))

    sem_label = torch.gather(end_points['sem_cls_label'], 1, assignment)
    sem_loss = cls_loss('sem_cls_scores', sem_label, config.num_class)

    box_loss = center_loss + 0.1 * heading_class_loss + heading_reg + 0.1 * size_class_loss + \
        size_reg
    for h, prefix in enumerate(prefixes):
        end_points[prefix + 'center_loss'] = center_loss[h]
        end_points[prefix + 'heading_cls_loss'] = heading_class_loss[h]
        end_points[prefix + 'heading_reg_loss'] = heading_reg[h]
        end_points[prefix + 'size_cls_loss'] = size_class_loss[h]
        end_points[prefix + 'size_reg_loss'] = size_reg[h]
        end_points[prefix + 'box_loss'] = box_loss[h]
        end_points[prefix + 'sem_cls_loss'] = sem_loss[h]
    return box_loss.sum(), sem_loss.sum(), end_points


def get_loss(end_points, config, num_decoder_layers, query_points_generator_loss_coef,
             obj_loss_coef, box_loss_coef, sem_cls_loss_coef, query_points_obj_topk=5,
             center_loss_type='smoothl1', center_delta=1.0, size_loss_type='smoothl1',
             size_delta=1.0, heading_loss_type='smoothl1', heading_delta=1.0):
    """loss_helper.py:278-319; returns (loss, end_points)."""
    if 'seeds_obj_cls_logits' in end_points:
        gen_loss = compute_points_obj_cls_loss_hard_topk(end_points, query_points_obj_topk)
        end_points['query_points_generation_loss'] = gen_loss
    else:
        gen_loss = 0.0
    prefixes = head_prefixes(num_decoder_layers)
    if fused_loss.can_fuse(end_points, config, prefixes,
                           (center_loss_type, size_loss_type, heading_loss_type)):
        # every head's objectness / box / semantic terms and their gradient: csrc/gf_loss.hip
        heads = fused_loss.heads_loss(
            end_points, config, prefixes, (obj_loss_coef, box_loss_coef, sem_cls_loss_coef),
            (center_delta, heading_delta, size_delta),
            _mean_size(config, end_points['center_label'].device))
        loss = 10 * (query_points_generator_loss_coef * gen_loss) + heads
        end_points['loss'] = loss
        return loss, end_points
    obj_sum, end_points = compute_objectness_loss_based_on_query_points(end_points,
                                                                        num_decoder_layers)
    end_points['sum_heads_objectness_loss'] = obj_sum
    box_sum, sem_sum, end_points = compute_box_and_sem_cls_loss(
        end_points, config, num_decoder_layers, center_loss_type, center_delta=center_delta,
        size_loss_type=size_loss_type, size_delta=size_delta,
        heading_loss_type=heading_loss_type, heading_delta=heading_delta)
    end_points['sum_heads_box_loss'] = box_sum
    end_points['sum_heads_sem_cls_loss'] = sem_sum
    loss = query_points_generator_loss_coef * gen_loss + 1.0 / (num_decoder_layers + 1) * (
        obj_loss_coef * obj_sum + box_loss_coef * box_sum + sem_cls_loss_coef * sem_sum)
    loss = loss * 10
    end_points['loss'] = loss
    return loss, end_points


# ------------------------------------------------------------------ Back-to-Reality (weak labels)
def compute_points_obj_cls_loss_hard_topk_weak(end_points, topk):
    """loss_helper.py:322-384: like compute_points_obj_cls_loss_hard_topk, but with centre
    labels only -- the `topk` seeds nearest (Euclidean) to each labelled centre are positive."""
    box_label_mask = end_points['box_label_mask']
    seed_xyz = end_points['seed_xyz']
    logits = end_points['seeds_obj_cls_logits']
    gt_center = end_points['center_label'][:, :, 0:3]
    B, K, K2 = gt_center.shape[0], seed_xyz.shape[1], gt_center.shape[1]
    delta = seed_xyz.unsqueeze(2) - gt_center.unsqueeze(1)
    dist = torch.sqrt(torch.sum(delta ** 2, dim=-1) + 1e-6).transpose(1, 2).contiguous()
    topk_inds = torch.topk(dist, topk, largest=False)[1] * box_label_mask[:, :, None] + \
        (box_label_mask[:, :, None] - 1)
    topk_inds = topk_inds.long().view(B, -1)
    label = torch.zeros((B, K + 1), dtype=torch.long, device=seed_xyz.device)
    label.scatter_(1, torch.where(topk_inds < 0, torch.full_like(topk_inds, K), topk_inds), 1)
    label = label[:, :K]
    total = B * K
    end_points['points_hard_topk%d_pos_ratio' % topk] = torch.sum(label.float()) / float(total)
    end_points['points_hard_topk%d_neg_ratio' % topk] = \
        1 - end_points['points_hard_topk%d_pos_ratio' % topk]
    if fused_loss.focal_sum_fusable(logits, label) and K > 0:   # (as the fully supervised form)
        return fused_loss.focal_sum(logits.reshape(1, B * K), label, 1.0 / K, 1.0 / B)[0]
    weights = (label >= 0).float()
    weights = weights / torch.clamp(weights.sum(dim=1, keepdim=True), min=1.0)
    loss = sigmoid_focal_loss(logits.view(B, K, 1), label.unsqueeze(-1).float(), weights)
    return loss.sum() / B


def compute_objectness_loss_based_on_query_points_weak(end_points, num_decoder_layers):
    """loss_helper.py:416-476: a query point is positive when it lies within 0.3 m of a
    labelled centre; its target is the nearest centre."""
    xyz = end_points['query_points_xyz']
    gt_center = end_points['center_label'][:, :, 0:3]
    B, K = xyz.shape[:2]
    d = torch.sum((xyz.unsqueeze(2) - gt_center.unsqueeze(1)) ** 2, dim=-1)      # (B,K,K2)
    dist1, assignment = torch.min(d, dim=2)
    label = (torch.sqrt(dist1 + 1e-6) < 0.3).long()
    mask = torch.ones((B, K), device=xyz.device)
    weights = mask / torch.clamp(mask.sum(dim=1, keepdim=True), min=1.0)
    prefixes = head_prefixes(num_decoder_layers)
    scores = _stack(end_points, prefixes, 'objectness_scores')
    if fused_loss.focal_sum_fusable(scores, label) and K > 0:
        # (every weight is 1 / K: one launch for the seven heads' sums and their gradient)
        loss = fused_loss.focal_sum(scores.reshape(len(prefixes), B * K), label, 1.0 / K, 1.0 / B)
    else:
        loss = sigmoid_focal_loss(scores.reshape(-1, K, 1),
                                  label.unsqueeze(-1).float().repeat(len(prefixes), 1, 1),
                                  weights.repeat(len(prefixes), 1))
        loss = loss.view(len(prefixes), -1).sum(1) / B
    for h, prefix in enumerate(prefixes):
        end_points[prefix + 'objectness_label'] = label
        end_points[prefix + 'objectness_mask'] = weights   # (normalised in place upstream)
        end_points[prefix + 'object_assignment'] = assignment
        end_points[prefix + 'objectness_loss'] = loss[h]
    return loss.sum(), end_points


def compute_center_and_sem_cls_loss(end_points, config, num_decoder_layers,
                                    center_loss_type='smoothl1', center_delta=1.0, **_unused):
    """loss_helper.py:479-554: the weakly supervised box loss -- centre regression with a dead
    zone of 5 % of the class's mean size, size-class and semantic cross-entropy."""
    mean_size = _mean_size(config, end_points['center_label'].device)
    gt_center = end_points['center_label'][:, :, 0:3]
    prefixes = head_prefixes(num_decoder_layers)
    H = len(prefixes)
    assignment = end_points[prefixes[0] + 'object_assignment']
    label = end_points[prefixes[0] + 'objectness_label'].float()
    npos = torch.sum(label) + 1e-6
    B, K = label.shape
    per_head = lambda t: t.reshape(H, -1).sum(1) / npos                    # noqa: E731
    scls = torch.gather(end_points['size_class_label'], 1, assignment)
    margin = 0.05 * mean_size[scls]                                         # (B,K,3)
    err = torch.gather(gt_center, 1, assignment.unsqueeze(2).expand(-1, -1, 3)).unsqueeze(0) - \
        _stack(end_points, prefixes, 'center')
    if center_loss_type == 'smoothl1':
        center_loss = smoothl1_loss(err, delta=center_delta)
    elif center_loss_type == 'l1':
        center_loss = l1_loss(err)
    else:
        raise NotImplementedError
    center_loss = per_head(torch.clamp(center_loss - margin, min=0) * label.unsqueeze(2))

    def cls_loss(key, target, nclass):
        scores = _stack(end_points, prefixes, key).reshape(-1, nclass)
        ce = F.cross_entropy(scores, target.reshape(-1).repeat(H), reduction='none')
        return per_head(ce.view(H, B, K) * label)

    size_class_loss = cls_loss('size_scores', scls, config.num_size_cluster)
    sem_label = torch.gather(end_points['sem_cls_label'], 1, assignment)
    sem_loss = cls_loss('sem_cls_scores', sem_label, config.num_class)
    box_loss = center_loss + 0.1 * size_class_loss
    for h, prefix in enumerate(prefixes):
        end_points[prefix + 'center_loss'] = center_loss[h]
        end_points[prefix + 'size_cls_loss'] = size_class_loss[h]
        end_points[prefix + 'box_loss'] = box_loss[h]
        end_points[prefix + 'sem_cls_loss'] = sem_loss[h]
    return box_loss.sum(), sem_loss.sum(), end_points


def get_loss_weak(end_points, config, num_decoder_layers, query_points_generator_loss_coef,
                  obj_loss_coef, box_loss_coef, sem_cls_loss_coef, query_points_obj_topk=5,
                  center_loss_type='smoothl1', center_delta=1.0, size_loss_type='smoothl1',
                  size_delta=1.0, heading_loss_type='smoothl1', heading_delta=1.0):
    """loss_helper.py:557-606.  The reference also evaluates every fully supervised term here
    and adds it with weight 0.000; those are not evaluated (same `loss`, same values under
    every key the weak functions set; the zero-weighted strong-only statistics
    `*heading_cls_loss`, `*heading_reg_loss`, `*size_reg_loss` are not filled)."""
    if 'seeds_obj_cls_logits' in end_points:
        gen_loss = compute_points_obj_cls_loss_hard_topk_weak(end_points, query_points_obj_topk)
        end_points['query_points_generation_loss'] = gen_loss
    else:
        gen_loss = 0.0
    prefixes = head_prefixes(num_decoder_layers)
    if fused_loss.can_fuse_weak(end_points, config, prefixes, center_loss_type):
        # the targets as the reference makes them (nearest labelled centre, positive within
        # 0.3 m), then all heads' objectness / centre / size-class / semantic terms and their
        # gradient in three launches (csrc/gf_loss.hip, weak mode)
        xyz = end_points['query_points_xyz']
        gt_center = end_points['center_label'][:, :, 0:3]
        B, K = xyz.shape[:2]
        d2 = torch.sum((xyz.unsqueeze(2) - gt_center.unsqueeze(1)) ** 2, dim=-1)      # (B,K,K2)
        dist1, assignment = torch.min(d2, dim=2)
        label = (torch.sqrt(dist1 + 1e-6) < 0.3).long()
        weights = torch.full((B, K), 1.0 / K, device=xyz.device)
        heads_total = fused_loss.weak_heads_loss(
            end_points, config, prefixes, (obj_loss_coef, box_loss_coef, sem_cls_loss_coef),
            center_delta, _mean_size(config, xyz.device), label, assignment, weights)
        loss = query_points_generator_loss_coef * gen_loss * 10 + heads_total
        end_points['loss'] = loss
        return loss, end_points
    obj_sum, end_points = compute_objectness_loss_based_on_query_points_weak(end_points,
                                                                             num_decoder_layers)
    end_points['sum_heads_objectness_loss'] = obj_sum
    box_sum, sem_sum, end_points = compute_center_and_sem_cls_loss(
        end_points, config, num_decoder_layers, center_loss_type, center_delta=center_delta)
    end_points['sum_heads_box_loss'] = box_sum
    end_points['sum_heads_sem_cls_loss'] = sem_sum
    loss = query_points_generator_loss_coef * gen_loss + 1.0 / (num_decoder_layers + 1) * (
        obj_loss_coef * obj_sum + box_loss_coef * box_sum + sem_cls_loss_coef * sem_sum)
    loss = loss * 10
    end_points['loss'] = loss
    return loss, end_points


def softmax_focal_loss(inputs, targets, gamma):
    """FocalLoss(class_num=2, gamma) with alpha = 1 (loss_helper.py:609-670): mean over the
    batch of -(1 - p_t)^gamma log p_t."""
    probs = torch.gather(torch.softmax(inputs, dim=-1), 1, targets.view(-1, 1))
    return (-torch.pow(1 - probs, gamma) * probs.log()).mean()


def get_loss_DA(end_points_S, end_points_T, config, num_decoder_layers,
                query_points_generator_loss_coef, obj_loss_coef, box_loss_coef,
                sem_cls_loss_coef, query_points_obj_topk=5, center_loss_type='smoothl1',
                center_delta=1.0, size_loss_type='smoothl1', size_delta=1.0,
                heading_loss_type='smoothl1', heading_delta=1.0):
    """The Back-to-Reality loss of GroupFree3D (loss_helper.py:673-712): half the fully
    supervised loss on the source (virtual) scenes + the weakly supervised loss on the target
    (real) scenes + 10 x the domain losses (global: focal, gamma 3; local: squared, weighted
    by the objectness label of the last head)."""
    args = (config, num_decoder_layers, query_points_generator_loss_coef, obj_loss_coef,
            box_loss_coef, sem_cls_loss_coef, query_points_obj_topk, center_loss_type,
            center_delta, size_loss_type, size_delta, heading_loss_type, heading_delta)
    loss = 0.5 * get_loss(end_points_S, *args)[0] + get_loss_weak(end_points_T, *args)[0]
    if end_points_S['global_d_pred'].is_cuda:
        from ..votenet import fused_loss as _fl
        from . import fused_loss as _gfl
        if _gfl.enabled() and _fl.domain_loss_fusable(end_points_S, end_points_T, 'last_'):
            # (the same terms as VoteNet's domain loss, unweighted: one launch each way)
            da_loss = _fl.domain_loss(end_points_S, end_points_T, 3.0, 1.0, 'last_')
            end_points_S['DA_loss'] = da_loss
            return loss + 10 * da_loss, end_points_S, end_points_T
    g_S, g_T = end_points_S['global_d_pred'], end_points_T['global_d_pred']
    source_dloss = softmax_focal_loss(g_S, torch.zeros(g_S.size(0), dtype=torch.long,
                                                       device=g_S.device), 3)
    target_dloss = softmax_focal_loss(g_T, torch.ones(g_T.size(0), dtype=torch.long,
                                                      device=g_T.device), 3)
    l_S = end_points_S['last_local_d_pred'].transpose(1, 2).contiguous().squeeze(-1)
    l_T = end_points_T['last_local_d_pred'].transpose(1, 2).contiguous().squeeze(-1)
    source_dloss = source_dloss + torch.mean(l_S ** 2 * end_points_S['last_objectness_label'])
    target_dloss = target_dloss + torch.mean((1 - l_T) ** 2 *
                                             end_points_T['last_objectness_label'])
    da_loss = source_dloss + target_dloss
    end_points_S['DA_loss'] = da_loss
    loss = loss + 10 * da_loss
    return loss, end_points_S, end_points_T


def compute_jitter_loss(end_points):
    """Mean squared error of the predicted centre displacement (loss_helper.py:715-720)."""
    loss = ((end_points['center_jitter'] -
             end_points['jitter_pred'].transpose(1, 2).contiguous()) ** 2).mean()
    end_points['jitter_loss'] = loss
    return loss


def get_loss_DA_jitter(end_points_S, end_points_T, epoch, config, num_decoder_layers,
                       query_points_generator_loss_coef, obj_loss_coef, box_loss_coef,
                       sem_cls_loss_coef, query_points_obj_topk=5, **kw):
    """CenterRefine loss of GroupFree3D (loss_helper.py:723-774): the centre labels are first
    moved back by the known (source) / predicted (target, detached) displacement, ramped in
    over 120 epochs; then get_loss_DA with 0.5 x the source jitter-regression loss inside the
    domain term.  (The reference edits the batch tensors in place; here the corrected centres
    replace the end_points entries.)"""
    if epoch > -1:
        ramp = min(epoch / 120.0, 1.0)
        end_points_S['center_label'] = end_points_S['center_label'] - \
            ramp * end_points_S['center_jitter']
        corr_T = end_points_T['jitter_pred'].transpose(1, 2) * \
            end_points_T['box_label_mask'].unsqueeze(-1)
        end_points_T['center_label'] = (end_points_T['center_label'] - ramp * corr_T).detach()
    jitter_loss_S = compute_jitter_loss(end_points_S)
    loss, end_points_S, end_points_T = get_loss_DA(
        end_points_S, end_points_T, config, num_decoder_layers,
        query_points_generator_loss_coef, obj_loss_coef, box_loss_coef, sem_cls_loss_coef,
        query_points_obj_topk, **kw)
    loss = loss + 10 * (0.5 * jitter_loss_S)
    return loss, end_points_S, end_points_T
