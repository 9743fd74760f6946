"""Training step of GroupFree3D (detection/GroupFree3D/train_GF_FSB.py:196-322): model,
AdamW with the decoder's own learning rate, loss with the script's default coefficients,
gradient clipping."""
import os

import torch

from ..pointnet2 import fused_backbone
from ..votenet.train import FastAdamW, _sync_grads, _zero_grad, backward, grad_sinks
from .detector import GroupFreeDetector, GroupFreeDetector_DA, GroupFreeDetector_DA_jitter
from . import fused_attention
from .loss_helper import get_loss

# train_GF_FSB.py:42-52
LOSS_ARGS = dict(num_decoder_layers=6, query_points_generator_loss_coef=0.8, obj_loss_coef=0.1,
                 box_loss_coef=1, sem_cls_loss_coef=0.1, query_points_obj_topk=4,
                 center_loss_type='smoothl1', center_delta=1.0, size_loss_type='smoothl1',
                 size_delta=1.0, heading_loss_type='smoothl1', heading_delta=1.0)


def build_model(cfg, device, input_feature_dim=0, num_proposal=256, seed=0,
                domain_adaptation=False, center_refine=False, **kw):
    """Random-init GroupFreeDetector with the script defaults (train_GF_FSB.py:26-34,196-217:
    no height channel unless --use_height, 256 query points, KPS sampling, six decoder layers,
    dropout 0.1, 'loc_learned' / 'xyz_learned' position embeddings)."""
    torch.manual_seed(seed)
    # (the scripts' default, train_GF_FSB.py:36; the class's own default is 'xyz_learned')
    kw.setdefault('self_position_embedding', 'loc_learned')
    cls = GroupFreeDetector_DA_jitter if center_refine else (
        GroupFreeDetector_DA if domain_adaptation else GroupFreeDetector)
    net = cls(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                            cfg.mean_size_arr, input_feature_dim=input_feature_dim,
                            num_proposal=num_proposal, **kw)
    return net.to(device)


def make_optimizer(net, lr=0.004, decoder_lr=0.0004, weight_decay=0.0005, capturable=False):
    """AdamW, decoder parameters at their own learning rate (train_GF_FSB.py:233-244); the
    fused multi-tensor implementation on the GPU (`capturable`: for GraphedTrainStep)."""
    named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if "decoder" not in n]},
              {"params": [p for n, p in named if "decoder" in n], "lr": decoder_lr}]
    fused = all(p.is_cuda for _, p in named)
    # (capturable: the step counters live on the device and the learning rates may be tensors;
    # the stock step() handles that)
    fast = fused and not capturable and os.environ.get("BTR_FAST_ADAM", "1") != "0"
    cls = FastAdamW if fast else torch.optim.AdamW
    return cls(groups, lr=lr, weight_decay=weight_decay, fused=fused,
               capturable=bool(capturable and fused))


def clip_and_step(net, optimizer, clip_norm):
    """Gradient clipping + optimizer step (train_GF_FSB.py:316-319); FastAdamW does both in its
    update kernels."""
    if clip_norm > 0 and isinstance(optimizer, FastAdamW):
        optimizer.step(clip_norm=clip_norm)
        return
    if clip_norm > 0:
        torch.nn.utils.clip_grad_norm_(net.parameters(), clip_norm, foreach=True)
    optimizer.step()


def get_scheduler(optimizer, n_iter_per_epoch, lr_scheduler="step", max_epoch=400,
                  lr_decay_epochs=(280, 340), lr_decay_rate=0.1, warmup_epoch=-1,
                  warmup_multiplier=100):
    """Per-ITERATION learning-rate schedule of train_GF_*.py (utils/lr_scheduler.py:65-87;
    defaults train_GF_FSB.py:66-82): step decay at the given epochs or cosine annealing, with
    an optional linear warm-up from lr / multiplier over `warmup_epoch` epochs; call `.step()`
    after every optimizer step."""
    from torch.optim.lr_scheduler import (CosineAnnealingLR, LinearLR, MultiStepLR,
                                          SequentialLR)
    if "cosine" in lr_scheduler:
        sched = CosineAnnealingLR(optimizer, eta_min=0.000001,
                                  T_max=(max_epoch - warmup_epoch) * n_iter_per_epoch)
    elif "step" in lr_scheduler:
        if isinstance(lr_decay_epochs, int):
            lr_decay_epochs = [lr_decay_epochs]
        sched = MultiStepLR(optimizer, gamma=lr_decay_rate,
                            milestones=[(m - warmup_epoch) * n_iter_per_epoch
                                        for m in lr_decay_epochs])
    else:
        raise NotImplementedError("scheduler %s not supported" % lr_scheduler)
    if warmup_epoch > 0:
        warm = LinearLR(optimizer, start_factor=1.0 / warmup_multiplier, end_factor=1.0,
                        total_iters=warmup_epoch * n_iter_per_epoch)
        sched = SequentialLR(optimizer, [warm, sched],
                             milestones=[warmup_epoch * n_iter_per_epoch])
    return sched


def train_step(net, optimizer, batch, cfg, loss_args=None, clip_norm=0.1, criterion=None,
               sampling=None, next_batch=None):
    """One optimisation step (train_GF_FSB.py:287-322) on `batch` (label dict on the model's
    device, GroupFree3D schema: VoteNet's keys + size_gts, point_obj_mask,
    point_instance_label).  Returns (loss, end_points); no host synchronisation.
    `criterion`: `get_loss` (default) or `get_loss_weak` (train_GF_WSB.py:217).
    `sampling` / `next_batch`: software pipelining as in votenet.train.train_step -- the
    sampling pyramid of the NEXT batch (coordinates only) runs on the side stream under this
    step's backward and comes back as end_points['next_sampling']."""
    loss_args = dict(LOSS_ARGS, **(loss_args or {}))
    inputs = {'point_clouds': batch['point_clouds']}
    if sampling is not None:
        inputs['sampling'] = sampling
    if inputs['point_clouds'].is_cuda:   # new attention-dropout masks every step (also when
        fused_attention.bump_step(inputs['point_clouds'].device)   # the step is a replayed graph)
    end_points = net(inputs)
    for key in batch:
        assert key not in end_points
        end_points[key] = batch[key]
    loss, end_points = (criterion or get_loss)(end_points, cfg, **loss_args)
    _zero_grad(net, optimizer)
    if next_batch is not None:
        core = net.module if hasattr(net, "module") else net
        end_points['next_sampling'] = core.backbone_net.prefetch_sampling(
            next_batch['point_clouds'])
    backward(loss)
    _sync_grads(net)          # data parallel: one all-reduce of the flat gradient buffer
    clip_and_step(net, optimizer, clip_norm)
    return loss, end_points


class GraphedTrainStep(object):
    """The whole training step captured once into a HIP graph and replayed.

    A GroupFree3D step is ~3 500 kernel launches (six decoder layers, seven prediction heads
    with eight loss terms each); enqueueing them costs the host 44 ms while the GPU needs 25 ms
    (tools/gf_times.py), so the eager loop is host-bound.  Replaying a captured graph removes
    the enqueue cost.  Conditions: fixed batch shapes (labels are copied into static buffers
    by `__call__`), AdamW with `capturable=True` (step counter on the device), no host reads
    inside the step -- all true for `train_step`.  Learning-rate changes must go through
    tensor-valued `lr`s; BatchNorm momentum changes need a new capture."""

    def __init__(self, net, optimizer, batch, cfg, loss_args=None, clip_norm=0.1, warmup=3,
                 batch_T=None):
        """`batch_T`: capture the Back-to-Reality step (`train_step_br`, source = `batch`)."""
        self.static = {k: v.clone() for k, v in batch.items()}
        self.static_T = None if batch_T is None else {k: v.clone() for k, v in batch_T.items()}
        self.net, self.optimizer = net, optimizer

        def step():
            if self.static_T is None:
                return train_step(net, optimizer, self.static, cfg, loss_args, clip_norm)
            out = train_step_br(net, optimizer, self.static, self.static_T, cfg, loss_args,
                                clip_norm)
            return out[0], out[1:]

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), fused_backbone.layerwise():
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.end_points = step()

    def __call__(self, batch=None, batch_T=None):
        for src, dst in ((batch, self.static), (batch_T, self.static_T)):
            if src is not None:
                for k, v in src.items():
                    if v is not dst[k]:
                        dst[k].copy_(v, non_blocking=True)
        self.graph.replay()
        fused_backbone._ext.RUNNING_STATS_EPOCH[0] += 1   # the replay moves running statistics
        return self.loss, self.end_points


class GraphedPipelinedStep(object):
    """GraphedTrainStep with the NEXT batch's sampling pyramid inside the same graph, on a side
    stream under the backward (votenet.train.GraphedPipelinedStep has the scheme): the four FPS
    levels of a 4 x 50 000-point batch occupy 4 of 256 CUs for 3.3 ms, which this hides.
    `cur` (whole batch) and `nxt_pc` are static buffers filled by __call__; `prime(batch)`
    computes the pyramid of the first batch eagerly."""

    def __init__(self, net, optimizer, batch, next_batch, cfg, loss_args=None, clip_norm=0.1,
                 warmup=3):
        self.net, self.optimizer = net, optimizer
        self.cur = {k: v.clone() for k, v in batch.items()}
        self.nxt_pc = next_batch['point_clouds'].clone()
        bb = net.backbone_net
        pyr = bb.prefetch_sampling(self.cur['point_clouds'])
        torch.cuda.synchronize()
        self.p_out = [inds.clone() for inds, _ in pyr]
        self.p_in = [t.clone() for t in self.p_out]

        def step():
            for dst, src in zip(self.p_in, self.p_out):
                dst.copy_(src)
            loss, end = train_step(net, optimizer, self.cur, cfg, loss_args, clip_norm,
                                   sampling=[(t, None) for t in self.p_in],
                                   next_batch={'point_clouds': self.nxt_pc})
            torch.cuda.current_stream().wait_stream(
                bb._get_side_stream(self.cur['point_clouds'].device, "_prefetch_stream"))
            for dst, (inds, _) in zip(self.p_out, end['next_sampling']):
                dst.copy_(inds)
            return loss, end

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), fused_backbone.layerwise():
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.end_points = step()

    def prime(self, batch):
        pyr = self.net.backbone_net.prefetch_sampling(batch['point_clouds'])
        main = torch.cuda.current_stream()
        for dst, (inds, ev) in zip(self.p_out, pyr):
            if ev is not None:
                main.wait_event(ev)
            dst.copy_(inds)

    def __call__(self, batch, next_batch):
        for k, v in batch.items():
            if v is not self.cur[k]:
                self.cur[k].copy_(v, non_blocking=True)
        if next_batch is not None:
            self.nxt_pc.copy_(next_batch['point_clouds'], non_blocking=True)
        self.graph.replay()
        fused_backbone._ext.RUNNING_STATS_EPOCH[0] += 1   # the replay moves running statistics
        return self.loss, self.end_points


def train_step_br(net, optimizer, batch_S, batch_T, cfg, loss_args=None, clip_norm=0.1,
                  sampling_S=None, next_batch_S=None, sampling_T=None, next_batch_T=None):
    """One Back-to-Reality step of GroupFree3D (train_GF_BR.py:322-365): the SAME
    GroupFreeDetector_DA runs a source (virtual, fully labelled) and a target (real, centre
    labels only) forward, then one `get_loss_DA`, one backward, clipping, one AdamW step.
    `sampling_S/_T`, `next_batch_S/_T`: software pipelining as in votenet.train.train_step_br --
    the target pyramid of this step runs on the side stream under the source forward, the NEXT
    step's two pyramids under this step's forwards / backward; they come back as
    end_points_S/_T['next_sampling']."""
    from ..votenet.train import _prefetch_next, _source_inputs
    from .loss_helper import get_loss_DA
    loss_args = dict(LOSS_ARGS, **(loss_args or {}))
    cuda = batch_S['point_clouds'].is_cuda
    if cuda:
        fused_attention.bump_step(batch_S['point_clouds'].device)
    core = net.module if hasattr(net, "module") else net
    nxt = ({}, {})
    inputs_T = {'point_clouds': batch_T['point_clouds']}
    if cuda:
        if sampling_T is None:
            sampling_T = core.backbone_net.prefetch_sampling(batch_T['point_clouds'])
        inputs_T['sampling'] = sampling_T
        _prefetch_next(core, nxt[0], nxt[1], next_batch_S, next_batch_T)
    with grad_sinks(net):   # (one flat gradient per native node: votenet/train.py)
        end_points_S = net(_source_inputs(batch_S, sampling_S))
        end_points_T = net(inputs_T)
    for key in batch_S:
        assert key not in end_points_S
        end_points_S[key] = batch_S[key]
    for key in batch_T:
        assert key not in end_points_T
        end_points_T[key] = batch_T[key]
    loss, end_points_S, end_points_T = get_loss_DA(end_points_S, end_points_T, cfg, **loss_args)
    end_points_S.update(nxt[0])
    end_points_T.update(nxt[1])
    _zero_grad(net, optimizer)
    backward(loss)
    _sync_grads(net)
    clip_and_step(net, optimizer, clip_norm)
    return loss, end_points_S, end_points_T


def train_step_br_jitter(net, optimizer, batch_S, batch_T, cfg, epoch=0, loss_args=None,
                         clip_norm=0.1):
    """One CenterRefine step of GroupFree3D (train_GF_BR_CenterRefine.py): like train_step_br,
    both forwards also pool features around the (noisy) centre labels and regress their
    displacement; batches need 'center_jitter' (synthetic.make_batch(..., center_jitter=))."""
    from .loss_helper import get_loss_DA_jitter
    loss_args = dict(LOSS_ARGS, **(loss_args or {}))
    with grad_sinks(net):
        end_points_S = net({'point_clouds': batch_S['point_clouds']}, batch_S['center_label'],
                           batch_S['sem_cls_label'])
        end_points_T = net({'point_clouds': batch_T['point_clouds']}, batch_T['center_label'],
                           batch_T['sem_cls_label'])
    for key in batch_S:
        end_points_S[key] = batch_S[key]
    for key in batch_T:
        end_points_T[key] = batch_T[key]
    loss, end_points_S, end_points_T = get_loss_DA_jitter(end_points_S, end_points_T, epoch, cfg,
                                                          **loss_args)
    _zero_grad(net, optimizer)
    backward(loss)
    _sync_grads(net)
    clip_and_step(net, optimizer, clip_norm)
    return loss, end_points_S, end_points_T


def evaluate_one_epoch(net, batches, cfg, config_dict=None, ap_iou_thresholds=(0.25, 0.5),
                       loss_args=None, criterion=None):
    """The evaluation pass of train_GF_FSB.py:354-445: eval-mode forward, loss statistics, and
    for EVERY prediction head (last decoder layer, proposal stage, intermediate layers) the
    boxes through parse_predictions -> AP at each IoU threshold.  Returns
    (mean stats dict, {iou_threshold: {prefix: metrics dict}})."""
    from ..votenet import ap_helper
    from ..votenet.train import EVAL_CONFIG_DICT
    from .loss_helper import head_prefixes
    loss_args = dict(LOSS_ARGS, **(loss_args or {}))
    config_dict = dict(config_dict or dict(EVAL_CONFIG_DICT, conf_thresh=0.0), dataset_config=cfg)
    core = net.module if hasattr(net, "module") else net
    n = core.num_decoder_layers
    prefixes = (['last_', 'proposal_'] + ['%dhead_' % i for i in range(n - 1)]) if n > 0 \
        else ['proposal_']
    assert sorted(prefixes) == sorted(head_prefixes(n))
    # (one calculator per head: the IoUs and the matching candidates do not depend on the
    # threshold, compute_metrics(thresholds) evaluates them once)
    ap_iou_thresholds = list(ap_iou_thresholds)
    calcs = {p: ap_helper.APCalculator(ap_iou_thresh=ap_iou_thresholds[0]) for p in prefixes}
    was_training = net.training
    net.eval()
    stat, nb = {}, 0
    try:
        for batch in batches:
            with torch.no_grad():
                end_points = net({'point_clouds': batch['point_clouds']})
                for key in batch:
                    end_points[key] = batch[key]
                _, end_points = (criterion or get_loss)(end_points, cfg, **loss_args)
            for key, v in end_points.items():
                if ('loss' in key or 'acc' in key or 'ratio' in key) and torch.is_tensor(v):
                    stat[key] = stat.get(key, 0) + v.detach()
            gt = ap_helper.parse_groundtruths(end_points, config_dict)
            for p in prefixes:
                calcs[p].step(ap_helper.parse_predictions(end_points, config_dict, p), gt)
            nb += 1
    finally:
        net.train(was_training)
    stats = {k: float(v) / max(nb, 1) for k, v in sorted(stat.items())}
    per_head = {p: c.compute_metrics(ap_iou_thresholds) for p, c in calcs.items()}
    return stats, {thr: {p: per_head[p][thr] for p in prefixes} for thr in ap_iou_thresholds}
