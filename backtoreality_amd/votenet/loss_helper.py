"""VoteNet loss: device-agnostic counterpart of detection/Votenet/models/loss_helper.py
(`get_loss` :336-400 and the three terms it sums) and detection/Votenet/utils/nn_distance.py.
Same arithmetic, same `end_points` keys; the reference's hard-coded `.cuda()` /
`torch.cuda.FloatTensor` allocations become allocations on the inputs' device.
Stock torch ops here; `get_loss` on GPU tensors dispatches to the fused HIP kernels of
fused_loss.py (the caller right after the hot path).
"""
import numpy as np
import torch
import torch.nn as nn

FAR_THRESHOLD = 0.6
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3  # GT votes per point
OBJECTNESS_CLS_WEIGHTS = [0.2, 0.8]  # more weight on positive objectness


def huber_loss(error, delta=1.0):
    """0.5*x^2 for |x| <= delta, else 0.5*delta^2 + delta*(|x|-delta) (nn_distance.py:15-32)."""
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return 0.5 * quadratic ** 2 + delta * linear


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """Brute-force chamfer terms between pc1 (B,N,C) and pc2 (B,M,C) (nn_distance.py:34-61):
    -> dist1 (B,N), idx1 (B,N), dist2 (B,M), idx2 (B,M)."""
    pc_diff = pc1.unsqueeze(2) - pc2.unsqueeze(1)  # (B,N,M,C), broadcast instead of repeat
    if l1smooth:
        pc_dist = torch.sum(huber_loss(pc_diff, delta), dim=-1)
    elif l1:
        pc_dist = torch.sum(torch.abs(pc_diff), dim=-1)
    else:
        pc_dist = torch.sum(pc_diff ** 2, dim=-1)
    dist1, idx1 = torch.min(pc_dist, dim=2)
    dist2, idx2 = torch.min(pc_dist, dim=1)
    return dist1, idx1, dist2, idx2


_OBJ_W = {}


def _objectness_weights(like):
    key = (like.device, like.dtype)
    if key not in _OBJ_W:
        _OBJ_W[key] = torch.tensor(OBJECTNESS_CLS_WEIGHTS, dtype=like.dtype, device=like.device)
    return _OBJ_W[key]


def _masked_mean(values, mask):
    return torch.sum(values * mask) / (torch.sum(mask) + 1e-6)


def compute_vote_loss(end_points):
    """Seeds on objects must vote for (one of) their GT centres (loss_helper.py:24-69)."""
    B, num_seed = end_points['seed_xyz'].shape[0], end_points['seed_xyz'].shape[1]
    vote_xyz = end_points['vote_xyz']             # (B, num_seed*vote_factor, 3)
    seed_inds = end_points['seed_inds'].long()    # (B, num_seed) into the input cloud

    seed_gt_votes_mask = torch.gather(end_points['vote_label_mask'], 1, seed_inds)
    inds9 = seed_inds.view(B, num_seed, 1).repeat(1, 1, 3 * GT_VOTE_FACTOR)
    seed_gt_votes = torch.gather(end_points['vote_label'], 1, inds9)
    seed_gt_votes = seed_gt_votes + end_points['seed_xyz'].repeat(1, 1, 3)

    vote_xyz_r = vote_xyz.view(B * num_seed, -1, 3)
    gt_r = seed_gt_votes.view(B * num_seed, GT_VOTE_FACTOR, 3)
    _, _, dist2, _ = nn_distance(vote_xyz_r, gt_r, l1=True)
    votes_dist, _ = torch.min(dist2, dim=1)
    votes_dist = votes_dist.view(B, num_seed)
    return _masked_mean(votes_dist, seed_gt_votes_mask.float())


def compute_objectness_loss(end_points):
    """Label proposals by distance to the nearest GT centre (loss_helper.py:111-152)."""
    agg = end_points['aggregated_vote_xyz']
    gt_center = end_points['center_label'][:, :, 0:3]
    B, K = gt_center.shape[0], agg.shape[1]
    dist1, ind1, _, _ = nn_distance(agg, gt_center)

    euclidean_dist1 = torch.sqrt(dist1 + 1e-6)
    objectness_label = torch.zeros((B, K), dtype=torch.long, device=agg.device)
    objectness_mask = torch.zeros((B, K), device=agg.device)
    objectness_label[euclidean_dist1 < NEAR_THRESHOLD] = 1
    objectness_mask[euclidean_dist1 < NEAR_THRESHOLD] = 1
    objectness_mask[euclidean_dist1 > FAR_THRESHOLD] = 1

    scores = end_points['objectness_scores']
    weight = _objectness_weights(scores)
    criterion = nn.CrossEntropyLoss(weight, reduction='none')
    loss = criterion(scores.transpose(2, 1), objectness_label)
    loss = _masked_mean(loss, objectness_mask)
    return loss, objectness_label, objectness_mask, ind1


def compute_box_and_sem_cls_loss(end_points, config):
    """Centre / heading / size / semantic-class terms (loss_helper.py:154-228)."""
    num_heading_bin = config.num_heading_bin
    num_size_cluster = config.num_size_cluster
    mean_size_arr = config.mean_size_arr

    object_assignment = end_points['object_assignment']
    B = object_assignment.shape[0]
    dev = object_assignment.device
    objectness_label = end_points['objectness_label'].float()
    ce = nn.CrossEntropyLoss(reduction='none')

    # centre: chamfer in both directions
    pred_center = end_points['center']
    gt_center = end_points['center_label'][:, :, 0:3]
    dist1, _, dist2, _ = nn_distance(pred_center, gt_center)
    center_loss = (_masked_mean(dist1, objectness_label) +
                   _masked_mean(dist2, end_points['box_label_mask']))

    # heading
    heading_class_label = torch.gather(end_points['heading_class_label'], 1, object_assignment)
    heading_class_loss = ce(end_points['heading_scores'].transpose(2, 1), heading_class_label)
    heading_class_loss = _masked_mean(heading_class_loss, objectness_label)

    heading_residual_label = torch.gather(end_points['heading_residual_label'], 1,
                                          object_assignment)
    heading_residual_normalized_label = heading_residual_label / (np.pi / num_heading_bin)
    K = heading_class_label.shape[1]
    heading_one_hot = torch.zeros((B, K, num_heading_bin), device=dev)
    heading_one_hot.scatter_(2, heading_class_label.unsqueeze(-1), 1)
    heading_reg_loss = huber_loss(
        torch.sum(end_points['heading_residuals_normalized'] * heading_one_hot, -1) -
        heading_residual_normalized_label, delta=1.0)
    heading_reg_loss = _masked_mean(heading_reg_loss, objectness_label)

    # size
    size_class_label = torch.gather(end_points['size_class_label'], 1, object_assignment)
    size_class_loss = ce(end_points['size_scores'].transpose(2, 1), size_class_label)
    size_class_loss = _masked_mean(size_class_loss, objectness_label)

    size_residual_label = torch.gather(end_points['size_residual_label'], 1,
                                       object_assignment.unsqueeze(-1).repeat(1, 1, 3))
    size_one_hot = torch.zeros((B, K, num_size_cluster), device=dev)
    size_one_hot.scatter_(2, size_class_label.unsqueeze(-1), 1)
    size_one_hot_tiled = size_one_hot.unsqueeze(-1).repeat(1, 1, 1, 3)
    predicted_size_residual_normalized = torch.sum(
        end_points['size_residuals_normalized'] * size_one_hot_tiled, 2)
    cache = getattr(config, "_mean_size_dev", None)  # the reference re-uploads it every step
    if cache is None or cache.device != dev:
        cache = torch.from_numpy(mean_size_arr.astype(np.float32)).to(dev)
        try:
            config._mean_size_dev = cache
        except AttributeError:
            pass
    mean_size = cache.unsqueeze(0).unsqueeze(0)
    mean_size_label = torch.sum(size_one_hot_tiled * mean_size, 2)
    size_residual_label_normalized = size_residual_label / mean_size_label
    size_reg_loss = torch.mean(
        huber_loss(predicted_size_residual_normalized - size_residual_label_normalized,
                   delta=1.0), -1)
    size_reg_loss = _masked_mean(size_reg_loss, objectness_label)

    # semantic class
    sem_cls_label = torch.gather(end_points['sem_cls_label'], 1, object_assignment)
    sem_cls_loss = ce(end_points['sem_cls_scores'].transpose(2, 1), sem_cls_label)
    sem_cls_loss = _masked_mean(sem_cls_loss, objectness_label)

    return (center_loss, heading_class_loss, heading_reg_loss, size_class_loss, size_reg_loss,
            sem_cls_loss)


def get_loss(end_points, config):
    """Total VoteNet loss; fills end_points with every term (loss_helper.py:336-400).
    On the GPU the whole loss and its gradient run as three HIP kernels (fused_loss.py);
    the composition below is the definition (and the path for CPU tensors, vote_factor > 1
    or BTR_FUSED_LOSS=0)."""
    if end_points['seed_xyz'].is_cuda:
        from . import fused_loss
        if fused_loss.can_fuse(end_points, config):
            return fused_loss.get_loss(end_points, config)
    vote_loss = compute_vote_loss(end_points)
    end_points['vote_loss'] = vote_loss

    objectness_loss, objectness_label, objectness_mask, object_assignment = \
        compute_objectness_loss(end_points)
    end_points['objectness_loss'] = objectness_loss
    end_points['objectness_label'] = objectness_label
    end_points['objectness_mask'] = objectness_mask
    end_points['object_assignment'] = object_assignment
    total = float(objectness_label.shape[0] * objectness_label.shape[1])
    end_points['pos_ratio'] = torch.sum(objectness_label.float()) / total
    end_points['neg_ratio'] = torch.sum(objectness_mask.float()) / total - end_points['pos_ratio']

    (center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss,
     sem_cls_loss) = compute_box_and_sem_cls_loss(end_points, config)
    end_points['center_loss'] = center_loss
    end_points['heading_cls_loss'] = heading_cls_loss
    end_points['heading_reg_loss'] = heading_reg_loss
    end_points['size_cls_loss'] = size_cls_loss
    end_points['size_reg_loss'] = size_reg_loss
    end_points['sem_cls_loss'] = sem_cls_loss
    box_loss = (center_loss + 0.1 * heading_cls_loss + heading_reg_loss + 0.1 * size_cls_loss +
                size_reg_loss)
    end_points['box_loss'] = box_loss

    loss = vote_loss + 0.5 * objectness_loss + box_loss + 0.1 * sem_cls_loss
    loss = loss * 10
    end_points['loss'] = loss

    obj_pred_val = torch.argmax(end_points['objectness_scores'], 2)
    obj_acc = torch.sum((obj_pred_val == objectness_label.long()).float() * objectness_mask) / \
        (torch.sum(objectness_mask) + 1e-6)
    end_points['obj_acc'] = obj_acc
    return loss, end_points


# ------------------------------------------------------------------ Back-to-Reality (DA) loss
def compute_weak_vote_loss(end_points):
    """Votes only need to land on SOME object centre (loss_helper.py:71-109): chamfer in both
    directions between the votes and the GT centres, L1."""
    B, num_seed = end_points['seed_xyz'].shape[0], end_points['seed_xyz'].shape[1]
    vote_xyz = end_points['vote_xyz']
    gt_center = end_points['center_label'][:, :, 0:3]
    dist1, _, dist2, _ = nn_distance(vote_xyz, gt_center, l1=True)
    votes_dist, _ = torch.min(dist1.view(B, num_seed, -1), dim=2)
    box_label_mask = end_points['box_label_mask']
    object_weight = torch.ones_like(end_points['sem_cls_label'])
    return torch.mean(votes_dist) + torch.sum(dist2 * object_weight * box_label_mask) / \
        (torch.sum(box_label_mask) + 1e-6)


def compute_center_and_sem_cls_loss(end_points, config):
    """Centre, size-class and semantic-class terms only (weak labels) (loss_helper.py:242-308)."""
    object_assignment = end_points['object_assignment']
    objectness_label = end_points['objectness_label'].float()
    ce = nn.CrossEntropyLoss(reduction='none')
    pred_center = end_points['center']
    gt_center = end_points['center_label'][:, :, 0:3]
    dist1, _, dist2, _ = nn_distance(pred_center, gt_center)
    center_loss = (_masked_mean(dist1, objectness_label) +
                   _masked_mean(dist2, end_points['box_label_mask']))
    size_class_label = torch.gather(end_points['size_class_label'], 1, object_assignment)
    size_class_loss = ce(end_points['size_scores'].transpose(2, 1), size_class_label)
    size_class_loss = _masked_mean(size_class_loss, objectness_label)
    sem_cls_label = torch.gather(end_points['sem_cls_label'], 1, object_assignment)
    sem_cls_loss = ce(end_points['sem_cls_scores'].transpose(2, 1), sem_cls_label)
    sem_cls_loss = _masked_mean(sem_cls_loss, objectness_label)
    return center_loss, size_class_loss, sem_cls_loss


def focal_loss(inputs, targets, gamma):
    """mean over the batch of -(1 - p_t)^gamma * log(p_t), p = softmax(inputs)
    (FocalLoss with alpha = 1, loss_helper.py:466-545)."""
    P = torch.softmax(inputs, dim=-1)
    probs = torch.gather(P, 1, targets.view(-1, 1))
    return (-torch.pow(1 - probs, gamma) * probs.log()).mean()


def _objectness_bookkeeping(end_points):
    loss, label, mask, assignment = compute_objectness_loss(end_points)
    end_points['objectness_loss'] = loss
    end_points['objectness_label'] = label
    end_points['objectness_mask'] = mask
    end_points['object_assignment'] = assignment
    total = float(label.shape[0] * label.shape[1])
    end_points['pos_ratio'] = torch.sum(label.float()) / total
    end_points['neg_ratio'] = torch.sum(mask.float()) / total - end_points['pos_ratio']
    return loss


def get_loss_DA(end_points_S, end_points_T, config):
    """Loss of one Back-to-Reality step over a source (virtual, fully labelled) and a target
    (real, weakly labelled) branch (loss_helper.py:548-664): source terms weighted 0.1, focal
    + squared domain losses weighted 0.5 through the gradient-reversal layers."""
    if end_points_S['seed_xyz'].is_cuda:
        from . import fused_loss
        if fused_loss.can_fuse(end_points_S, config) and fused_loss.can_fuse(end_points_T, config):
            return _get_loss_DA_fused(end_points_S, end_points_T, config, fused_loss)
    source_coefficient = 0.1
    vote_loss_S = compute_weak_vote_loss(end_points_S)
    vote_loss_T = compute_weak_vote_loss(end_points_T)
    end_points_S['vote_loss'] = vote_loss_S
    end_points_T['vote_loss'] = vote_loss_T
    vote_loss = source_coefficient * vote_loss_S + vote_loss_T

    objectness_loss = source_coefficient * _objectness_bookkeeping(end_points_S) + \
        _objectness_bookkeeping(end_points_T)

    (center_loss_S, heading_cls_loss, heading_reg_loss, size_cls_loss_S, size_reg_loss,
     sem_cls_loss_S) = compute_box_and_sem_cls_loss(end_points_S, config)
    end_points_S['center_loss'] = center_loss_S
    end_points_S['heading_cls_loss'] = heading_cls_loss
    end_points_S['heading_reg_loss'] = heading_reg_loss
    end_points_S['size_cls_loss'] = size_cls_loss_S
    end_points_S['size_reg_loss'] = size_reg_loss
    end_points_S['sem_cls_loss'] = sem_cls_loss_S
    box_loss_S = (center_loss_S + 0.1 * heading_cls_loss + heading_reg_loss +
                  0.1 * size_cls_loss_S + size_reg_loss)
    end_points_S['box_loss'] = box_loss_S

    center_loss_T, size_cls_loss_T, sem_cls_loss_T = \
        compute_center_and_sem_cls_loss(end_points_T, config)
    end_points_T['center_loss'] = center_loss_T
    end_points_T['size_cls_loss'] = size_cls_loss_T
    end_points_T['sem_cls_loss'] = sem_cls_loss_T
    box_loss_T = center_loss_T + 0.1 * size_cls_loss_T

    box_loss = source_coefficient * box_loss_S + box_loss_T
    sem_cls_loss = source_coefficient * sem_cls_loss_S + sem_cls_loss_T

    DA_loss = _domain_loss(end_points_S, end_points_T)
    end_points_S['DA_loss'] = DA_loss

    loss = vote_loss + 0.5 * objectness_loss + box_loss + 0.1 * sem_cls_loss + DA_loss
    loss = loss * 10
    end_points_S['loss'] = loss

    obj_pred_val = torch.argmax(end_points_S['objectness_scores'], 2)
    label_S, mask_S = end_points_S['objectness_label'], end_points_S['objectness_mask']
    end_points_S['obj_acc'] = torch.sum((obj_pred_val == label_S.long()).float() * mask_S) / \
        (torch.sum(mask_S) + 1e-6)
    return loss, end_points_S, end_points_T


def _domain_loss(end_points_S, end_points_T):
    """Focal loss on the global domain classifier + squared loss on the local one, both
    through the gradient-reversal layers (loss_helper.py:618-650)."""
    if end_points_S['global_d_pred'].is_cuda:
        from . import fused_loss
        if fused_loss.domain_loss_fusable(end_points_S, end_points_T):
            return fused_loss.domain_loss(end_points_S, end_points_T, 3.0)   # (one launch each way)
    da_coefficient = 0.5
    g_S = end_points_S['global_d_pred']
    l_S = end_points_S['local_d_pred'].transpose(1, 2).contiguous()
    domain_S = torch.zeros(g_S.size(0), dtype=torch.long, device=g_S.device)
    w_S = end_points_S['objectness_label'].unsqueeze(-1)
    source_dloss = da_coefficient * torch.mean(l_S ** 2 * w_S) + \
        da_coefficient * focal_loss(g_S, domain_S, 3)
    g_T = end_points_T['global_d_pred']
    l_T = end_points_T['local_d_pred'].transpose(1, 2).contiguous()
    domain_T = torch.ones(g_T.size(0), dtype=torch.long, device=g_T.device)
    w_T = end_points_T['objectness_label'].unsqueeze(-1)
    target_dloss = da_coefficient * torch.mean((1 - l_T) ** 2 * w_T) + \
        da_coefficient * focal_loss(g_T, domain_T, 3)
    return source_dloss + target_dloss


def _get_loss_DA_fused(end_points_S, end_points_T, config, fused_loss):
    """get_loss_DA with each branch's detection terms (forward + gradient) in the fused HIP
    kernels; only the domain losses stay torch ops.  Same end_points keys as above."""
    keys_S = ('vote_loss', 'objectness_loss', 'center_loss', 'heading_cls_loss',
              'heading_reg_loss', 'size_cls_loss', 'size_reg_loss', 'sem_cls_loss', 'box_loss',
              'pos_ratio', 'neg_ratio', 'obj_acc')
    keys_T = ('vote_loss', 'objectness_loss', 'center_loss', 'size_cls_loss', 'sem_cls_loss',
              'pos_ratio', 'neg_ratio')
    loss_S = fused_loss.get_loss_branch(end_points_S, config, fused_loss.W_DA_SOURCE, keys_S)
    loss_T = fused_loss.get_loss_branch(end_points_T, config, fused_loss.W_DA_TARGET, keys_T)
    DA_loss = _domain_loss(end_points_S, end_points_T)
    end_points_S['DA_loss'] = DA_loss
    loss = loss_S + loss_T + DA_loss * 10
    end_points_S['loss'] = loss
    return loss, end_points_S, end_points_T


def get_loss_weak(end_points, config):
    """Loss of the weakly supervised baseline (train_Votenet_WSB.py:170; loss_helper.py:403-464):
    centre labels only -- the weak vote loss, objectness, centre regression with its dead
    zone, size-class and semantic cross-entropy; the target-branch terms of get_loss_DA on
    their own.  Fused kernels on the GPU, torch composition otherwise."""
    if end_points['seed_xyz'].is_cuda:
        from . import fused_loss
        if fused_loss.can_fuse(end_points, config):
            keys = ('vote_loss', 'objectness_loss', 'center_loss', 'size_cls_loss',
                    'sem_cls_loss', 'pos_ratio', 'neg_ratio', 'obj_acc')
            loss = fused_loss.get_loss_branch(end_points, config, fused_loss.W_DA_TARGET, keys)
            end_points['loss'] = loss
            return loss, end_points
    vote_loss = compute_weak_vote_loss(end_points)
    end_points['vote_loss'] = vote_loss
    _objectness_bookkeeping(end_points)
    center_loss, size_cls_loss, sem_cls_loss = compute_center_and_sem_cls_loss(end_points, config)
    end_points['center_loss'] = center_loss
    end_points['size_cls_loss'] = size_cls_loss
    end_points['sem_cls_loss'] = sem_cls_loss
    box_loss = center_loss + 0.1 * size_cls_loss
    loss = vote_loss + 0.5 * end_points['objectness_loss'] + box_loss + 0.1 * sem_cls_loss
    loss = loss * 10
    end_points['loss'] = loss
    obj_pred_val = torch.argmax(end_points['objectness_scores'], 2)
    label, mask = end_points['objectness_label'], end_points['objectness_mask']
    end_points['obj_acc'] = torch.sum((obj_pred_val == label.long()).float() * mask) / \
        (torch.sum(mask) + 1e-6)
    return loss, end_points


def compute_jitter_loss(end_points):
    """Mean squared error between the displacement applied to the GT centres and the
    regressed one (loss_helper.py:667-672)."""
    return ((end_points['center_jitter'] -
             end_points['jitter_pred'].transpose(1, 2).contiguous()) ** 2).mean()


def get_loss_DA_jitter(end_points_S, end_points_T, epoch, config):
    """CenterRefine loss (loss_helper.py:675-803): the GT centres are first moved back by the
    known (source) / predicted (target, detached) displacement, ramped in over 60 epochs; then
    get_loss_DA plus 0.1 x the source jitter-regression loss.  (The reference edits the batch
    tensors in place; here the corrected centres replace the end_points entries instead.)"""
    if epoch > -1:
        ramp = min(epoch / 60.0, 1.0)
        end_points_S['center_label'] = (end_points_S['center_label'] -
                                        ramp * end_points_S['center_jitter'])
        corr_T = end_points_T['jitter_pred'].transpose(1, 2) * \
            end_points_T['box_label_mask'].unsqueeze(-1)
        end_points_T['center_label'] = (end_points_T['center_label'] - ramp * corr_T).detach()
    jitter_loss_S = compute_jitter_loss(end_points_S)
    end_points_S['jitter_loss'] = jitter_loss_S
    loss, end_points_S, end_points_T = get_loss_DA(end_points_S, end_points_T, config)
    loss = loss + jitter_loss_S * (0.1 * 10)
    end_points_S['loss'] = loss
    return loss, end_points_S, end_points_T
