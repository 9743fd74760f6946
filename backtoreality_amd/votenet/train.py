"""One data-parallel VoteNet training step: the harness counterpart of
detection/Votenet/train_Votenet_FSB.py:211-244 (zero_grad -> forward -> get_loss -> backward
-> Adam step), scaled out the way the reference's GroupFree3D scripts do it
(detection/GroupFree3D/train_GF_FSB.py:450-474, :250): one process per GPU,
`init_process_group('nccl')` (RCCL over xGMI on ROCm), `DistributedDataParallel(...,
broadcast_buffers=False)` so BatchNorm statistics stay per rank, one bucketed gradient
all-reduce per step.  Scenes are independent, so the batch is sharded across ranks and the
only collective is that gradient all-reduce (3.83 MB for VoteNet's 956 408 parameters).
"""
import os

import torch
import torch.distributed as dist

from ..pointnet2 import fused_backbone
from . import loss_helper
from .votenet import VoteNet
from .votenet_da import VoteNet_DA, VoteNet_DA_jitter


def init_distributed(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torchrun); single process if
    those are absent.  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BTR_FORCE_DDP=1: a one-rank process group, so that the DDP wrapper and RCCL run on a
    # single GPU too (what a 1-GPU box can check of the N>1 path: tools/ddp_one_rank.sh)
    force = os.environ.get("BTR_FORCE_DDP", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank,
                                world_size=world)
    return rank, world, local_rank


def build_model(cfg, device, input_feature_dim=1, num_proposal=256, vote_factor=1,
                sampling='vote_fps', seed=0, domain_adaptation=False, center_refine=False):
    """Random-init VoteNet (or VoteNet_DA / VoteNet_DA_jitter) (weights from
    torch.manual_seed(seed), modules constructed in the reference's order) on `device`."""
    torch.manual_seed(seed)
    cls = VoteNet_DA_jitter if center_refine else (VoteNet_DA if domain_adaptation else VoteNet)
    net = cls(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster, cfg.mean_size_arr,
                  input_feature_dim=input_feature_dim, num_proposal=num_proposal,
                  vote_factor=vote_factor, sampling=sampling)
    return net.to(device)


class FlatGradParallel(torch.nn.Module):
    """Data parallelism for this model without the DistributedDataParallel machinery: after the
    backward the gradients are gathered into ONE flat f32 buffer by one multi-tensor copy, the
    only collective of the path (the gradient mean, 3.8 MB) is ONE all-reduce of
    that buffer, and every `.grad` is re-pointed at its slice of the buffer (no copy back).

    Why not DDP: its per-iteration bookkeeping (an autograd hook and an accumulate / copy
    kernel per parameter, two wrapped forwards per Back-to-Reality step) costs 0.25-1.2 ms of
    a 9.6 / 18.2 ms step on ONE rank (tools/ddp_one_rank.sh); overlapping the all-reduce with
    the backward, the thing that machinery buys, hides <0.1 ms here (3.8 MB over xGMI).
    `BTR_DP=ddp` selects DistributedDataParallel instead.  Same interface where the step
    functions touch it: `.module`, call = forward, `state_dict()` keys prefixed `module.`.
    Parameters are broadcast from rank 0 at construction; buffers (BatchNorm statistics)
    are per replica, like the reference (`broadcast_buffers=False`, train_GF_FSB.py:250).
    A parameter without a gradient (the CenterRefine model's jitter_netD) contributes zeros and
    keeps `.grad = None`, so the optimizer skips it on every rank alike."""

    def __init__(self, module, process_group=None):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.world = dist.get_world_size(process_group)
        params = [p for p in module.parameters() if p.requires_grad]
        assert params and all(p.dtype == torch.float32 for p in params)
        dev = params[0].device
        assert all(p.device == dev for p in params)
        with torch.no_grad():
            for p in module.parameters():
                dist.broadcast(p.data, 0, group=process_group)
        # every slice starts on a 16-byte boundary (the optimizer kernel's vector loads; the
        # padding floats stay zero and ride along in the all-reduce)
        pad4 = lambda n: (n + 3) // 4 * 4
        total = sum(pad4(p.numel()) for p in params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self._views = []
        o = 0
        for p in params:
            self._views.append((p, self.flat_grad[o:o + p.numel()].view_as(p)))
            o += pad4(p.numel())
        self._avg = dist.get_backend(process_group) == "nccl"   # RCCL has AVG; gloo only sums

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def sync_gradients(self):
        """Mean of the gradients over the ranks.  Enqueued behind the backward: the collective
        runs on the backend's stream and the current stream waits for it -- no host
        synchronisation."""
        src, dst = [], []
        for p, v in self._views:
            g = p.grad
            if g is None:
                v.zero_()
            elif g is not v:
                src.append(g)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        if self._avg:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.AVG, group=self.process_group)
        else:
            dist.all_reduce(self.flat_grad, group=self.process_group)
            self.flat_grad.div_(self.world)
        for p, v in self._views:
            if p.grad is not None:
                p.grad = v


def wrap_ddp(net, device):
    """The data-parallel wrapper when a process group is up (FlatGradParallel, or
    DistributedDataParallel with BTR_DP=ddp), else the bare module."""
    force = os.environ.get("BTR_FORCE_DDP", "0") == "1"
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        if os.environ.get("BTR_DP", "flat") != "ddp":
            return FlatGradParallel(net)
        ids = [device.index] if device.type == "cuda" else None
        # the CenterRefine model's jitter_netD has no loss term (commented out in the
        # reference, loss_helper.py:776,787), so its parameters never receive a gradient
        unused = bool(getattr(net, "center_refine", False))
        # gradient_as_bucket_view: the gradients ARE the all-reduce buffer (no per-parameter
        # copy kernels into and out of the bucket)
        return torch.nn.parallel.DistributedDataParallel(net, device_ids=ids,
                                                         broadcast_buffers=False,
                                                         find_unused_parameters=unused,
                                                         gradient_as_bucket_view=True)
    return net


_ONES = {}


def backward(loss):
    """loss.backward() with a cached unit gradient: autograd otherwise fills a fresh ones_like
    (loss) -- one launch per step at the head of the backward's dependent chain."""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), device=loss.device, dtype=loss.dtype)
    loss.backward(one if loss.dim() == 0 else None)


def _zero_grad(net, optimizer):
    """optimizer.zero_grad(set_to_none=True) without its per-parameter bookkeeping (0.2 ms of
    host time per step for GroupFree3D's ~400 tensors)."""
    for group in optimizer.param_groups:
        for p in group['params']:
            p.grad = None


def grad_sinks(net):
    """Context for the forwards of a step that runs the same parameters through TWO forwards
    (Back-to-Reality): every native node hands autograd ONE flat gradient (pointnet2/
    grad_sink.py) instead of a view per parameter, so the two branches' contributions are added
    by a launch per node, not by 117 per-parameter launches.  Off (a null context) under torch's
    DistributedDataParallel, whose reducer listens on the parameters' own gradient hooks, and
    with BTR_GRAD_SINK=0."""
    import contextlib
    from ..pointnet2 import grad_sink
    if os.environ.get("BTR_GRAD_SINK", "1") == "0" or \
            isinstance(net, torch.nn.parallel.DistributedDataParallel):
        return contextlib.nullcontext()
    return grad_sink.scope()


_HEAD_STREAMS = {}


def two_forwards(net, args_S, args_T):
    """(end_points_S, end_points_T) of the two forwards of a Back-to-Reality step through the
    same model.  On the GPU the source branch's head (voting, vote aggregation, proposal head,
    domain classifiers: small launches) runs on a side stream BESIDE the target branch's
    backbone (large kernels) -- the two do not depend on each other until the loss -- and the
    target branch's head waits for it, so shared state (BatchNorm running statistics, updated
    source first, then target) is touched in the reference's order.  Autograd runs every
    node's backward on its forward's stream, so the backward overlaps the same way.  Same
    kernels, same results (tests/test_grad_sink_gpu.py).  Off (two plain calls) for models without
    the two-stage forward, under torch's DistributedDataParallel (its forward prepares the
    reducer), on the CPU, while a HIP graph is captured, and with BTR_BR_OVERLAP=0."""
    core = net.module if hasattr(net, "module") else net
    pc = args_S[0]['point_clouds']
    if not (pc.is_cuda and hasattr(core, "forward_backbone") and
            os.environ.get("BTR_BR_OVERLAP", "1") != "0" and
            not isinstance(net, torch.nn.parallel.DistributedDataParallel) and
            not torch.cuda.is_current_stream_capturing()):
        return net(*args_S), net(*args_T)
    dev = pc.device
    main = torch.cuda.current_stream(dev)
    side = _HEAD_STREAMS.get(dev)
    if side is None:
        side = _HEAD_STREAMS[dev] = torch.cuda.Stream(device=dev)
    cS = args_S[1] if len(args_S) > 1 else None
    cT = args_T[1] if len(args_T) > 1 else None
    end_S = core.forward_backbone(*args_S)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        end_S = core.forward_head(end_S, cS)
    end_T = core.forward_backbone(*args_T)
    main.wait_stream(side)
    end_T = core.forward_head(end_T, cT)
    return end_S, end_T


def _sync_grads(net):
    if isinstance(net, FlatGradParallel):
        net.sync_gradients()


class _FastStep(object):
    """step() of torch.optim.Adam / AdamW (same state, same state_dict, same fused update
    kernels) without the Python bookkeeping of the stock optimizer once the state exists: the
    per-step `_init_group` walk, the device / dtype grouping and the profiler hooks cost
    0.45 ms of host time per step for ~100 parameter tensors (tools/host_profile.py) -- a tenth
    of the whole VoteNet step, 1.2 ms for GroupFree3D's ~400.  Covers what the training scripts
    use (fused kernels on one device, no amsgrad / maximize / closure / grad scaler); anything
    else goes through the stock step().

    `step(clip_norm=c)`: torch.nn.utils.clip_grad_norm_(parameters, c) folded into the update
    (train_GF_FSB.py:316-318 clips, then steps): one multi-tensor norm, and the fused kernel
    divides every gradient by max(1, (total_norm + 1e-6) / c) on its way in (its `grad_scale`
    operand) -- no Python loop over the parameters, no second pass over the gradients.
    Returns the total norm (a tensor) in that case."""
    _decoupled = False

    @torch.no_grad()
    def step(self, closure=None, clip_norm=None):
        if closure is not None or not self._fast_ok():
            return self._stock_step(closure, clip_norm)
        work = []
        for group in self.param_groups:
            cache = group.get('_btr_fast')
            # one walk over the parameters per step: their gradients; the cached lists stand when
            # the same parameters (by identity) have one
            every = group['params']
            grads = [p.grad for p in every]
            if any(g is None for g in grads):   # (`None in grads` would call Tensor.__eq__ per entry)
                params = [p for p, g in zip(every, grads) if g is not None]
                grads = [g for g in grads if g is not None]
            else:
                params = every
            if cache is None or len(cache[0]) != len(params) or \
                    any(a is not b for a, b in zip(cache[0], params)):
                if any(len(self.state[p]) == 0 for p in params):
                    return self._stock_step(None, clip_norm)   # first step: builds the state
                params = list(params)
                cache = group['_btr_fast'] = (
                    params, [self.state[p]['exp_avg'] for p in params],
                    [self.state[p]['exp_avg_sq'] for p in params],
                    [self.state[p]['step'] for p in params])
            if cache[0]:
                work.append((group, cache, grads))
        if work:
            done = self._library_step(work, clip_norm)
            if done is not None:
                return done[0]
        total = scale = None
        if clip_norm is not None and work:
            norms = torch._foreach_norm([g for _, _, grads in work for g in grads])
            total = torch.linalg.vector_norm(torch.stack(norms))
            scale = ((total + 1e-6) / float(clip_norm)).clamp_(min=1.0)
        kernel = torch._fused_adamw_ if self._decoupled else torch._fused_adam_
        for group, (params, exp_avgs, exp_avg_sqs, steps), grads in work:
            beta1, beta2 = group['betas']
            torch._foreach_add_(steps, 1)
            kernel(params, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False,
                   lr=group['lr'], beta1=beta1, beta2=beta2, weight_decay=group['weight_decay'],
                   eps=group['eps'], maximize=False, grad_scale=scale, found_inf=None)
        return total

    # ---- the update of ALL groups as one launch of the library (csrc/optimizer.hip)
    def _library_step(self, work, clip_norm):
        """(total norm | None,) when btr_adam_multi did the update, None when it cannot: CUDA f32
        contiguous parameters / gradients on one device, equal betas / eps in all groups, at most
        ADAM_MAX_GROUPS groups, not while a HIP graph is captured (`BTR_ADAM_KERNEL=0`: torch's
        fused kernels).  With `clip_norm` the total gradient norm comes from two more launches
        over the same table (btr_grad_sumsq_multi / btr_grad_norm_final) and its clip factor is
        the update's grad_scale operand.  The per-tensor table (parameter, moments, the tensor's
        own DEVICE step counter, size, group index) and the chunk map live on the device and are
        rebuilt only when the parameter set changes; the gradient pointers and the groups'
        learning rates / weight decays ride in the kernel arguments, so a scheduler that moves
        lr every iteration (train_GF_FSB.py:322) never touches the table.  Nothing here reads
        the device: the bias corrections come from the step tensors torch keeps in the state
        (incremented below exactly like the stock step does), so steps that other kernels took
        in between -- a HIP-graph replay, a fallback, BTR_ADAM_KERNEL toggled -- and parameters
        whose counts differ are all handled (tests/test_optimizer_gpu.py)."""
        import ctypes
        import numpy as np
        from ..pointnet2 import _ext
        if os.environ.get("BTR_ADAM_KERNEL", "1") == "0":
            return None
        g0 = work[0][0]
        first = work[0][1][0][0]
        if not first.is_cuda or torch.cuda.is_current_stream_capturing() or \
                len(work) > _ext.ADAM_MAX_GROUPS:
            return None
        dev = first.device
        key = tuple((id(cache[0]), group['betas'], group['eps']) for group, cache, _ in work)
        st = getattr(self, '_btr_lib', None)
        if st is None or st['key'] != key:
            for group, (params, exp_avgs, exp_avg_sqs, stp), _ in work:   # checked once per table
                if group['betas'] != g0['betas'] or group['eps'] != g0['eps']:
                    return None
                for t in list(params) + list(exp_avgs) + list(exp_avg_sqs):
                    if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
                        return None
                for t in stp:      # fused / capturable state: an f32 scalar on the device
                    if t.device != dev or t.dtype != torch.float32 or t.numel() != 1:
                        return None
            n_t = sum(len(cache[0]) for _, cache, _ in work)
            items = (_ext.AdamItem * n_t)()
            chunk = _ext._lib.btr_adam_chunk()
            cmap, steps, i = [], [], 0
            for gi, (group, (params, exp_avgs, exp_avg_sqs, stp), _) in enumerate(work):
                for p, m, v, t in zip(params, exp_avgs, exp_avg_sqs, stp):
                    it = items[i]
                    it.p, it.m, it.v, it.n = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                    it.step, it.group = t.data_ptr(), gi
                    # (the gradient's alignment is checked per step: see below)
                    it.vec = int(p.numel() % 4 == 0 and all(
                        t.data_ptr() % 16 == 0 for t in (p, m, v)))
                    cmap += [(i, e) for e in range(0, p.numel(), chunk)]
                    i += 1
                steps += stp
            raw = np.frombuffer(items, dtype=np.uint8).copy()
            # launches: blocks of ADAM_MAX_TENSORS tensors = contiguous runs of the chunk map
            blocks, c0 = [], 0
            for t0 in range(0, n_t, _ext.ADAM_MAX_TENSORS):
                t1 = min(n_t, t0 + _ext.ADAM_MAX_TENSORS)
                c1 = c0
                while c1 < len(cmap) and cmap[c1][0] < t1:
                    c1 += 1
                blocks.append((t0, t1, c0, c1 - c0, _ext.AdamGrads()))
                c0 = c1
            st = self._btr_lib = {
                'key': key, 'n': n_t, 'steps': steps, 'blocks': blocks,
                'groups': _ext.AdamGroups(),
                'items': torch.from_numpy(raw).to(dev),
                'cmap': torch.tensor(cmap, dtype=torch.int32).reshape(-1, 2).contiguous().to(dev),
                'vec': [bool(items[j].vec) for j in range(n_t)],
                'partial': torch.empty((max(len(cmap), 1),), dtype=torch.float32, device=dev)}
        ptrs = []
        for _, _, grads in work:
            for g in grads:
                if g.dtype != torch.float32 or not g.is_contiguous() or g.device != dev:
                    return None        # torch's kernels take this step
                ptrs.append(g.data_ptr())
        if any(v and (q & 15) for v, q in zip(st['vec'], ptrs)):
            return None                # a gradient the 16-byte loads cannot take
        torch._foreach_add_(st['steps'], 1)     # the state's step tensors: what torch keeps, and
        beta1, beta2 = g0['betas']              # what the kernel reads its bias corrections from
        cmap_ptr = st['cmap'].data_ptr()
        gr = st['groups']
        for gi, (group, _, _) in enumerate(work):
            gr.lr[gi], gr.wd[gi] = float(group['lr']), float(group['weight_decay'])
        norm = None
        _ext.RUNNING_STATS_EPOCH[0] += 1   # parameters move through raw pointers
        with _ext._on(first) as dv:
            stream = _ext._stream(dv)
            for t0, t1, c0, nchunks, gp in st['blocks']:
                gp.g[0:t1 - t0] = ptrs[t0:t1]
            if clip_norm is not None:
                norm = torch.empty((2,), dtype=torch.float32, device=dev)   # (total, clip factor)
                part = st['partial'].data_ptr()
                for t0, t1, c0, nchunks, gp in st['blocks']:
                    _ext._call(_ext._lib.btr_grad_sumsq_multi, nchunks, t0, _ext._p(st['items']),
                               ctypes.addressof(gp), cmap_ptr + 8 * c0, part + 4 * c0, stream)
                _ext._call(_ext._lib.btr_grad_norm_final, st['cmap'].shape[0], part,
                           float(clip_norm), _ext._p(norm), stream)
            scale = norm.data_ptr() + 4 if norm is not None else None
            for t0, t1, c0, nchunks, gp in st['blocks']:
                _ext._call(_ext._lib.btr_adam_multi, nchunks, t0, _ext._p(st['items']),
                           ctypes.addressof(gp), ctypes.addressof(gr), cmap_ptr + 8 * c0,
                           float(beta1), float(beta2), float(g0['eps']), int(self._decoupled),
                           scale, stream)
        return (norm[0] if norm is not None else None,)

    def _stock_step(self, closure, clip_norm):
        total = None
        if clip_norm is not None:
            total = torch.nn.utils.clip_grad_norm_(
                [p for g in self.param_groups for p in g['params']], clip_norm, foreach=True)
        out = super().step(closure)
        return total if clip_norm is not None else out

    def _fast_ok(self):
        for g in self.param_groups:
            if not g.get('fused') or g.get('amsgrad') or g.get('maximize') or \
                    g.get('differentiable') or \
                    bool(g.get('decoupled_weight_decay')) != self._decoupled or \
                    isinstance(g['lr'], torch.Tensor) or isinstance(g['betas'][0], torch.Tensor):
                return False
        return getattr(self, 'grad_scale', None) is None and getattr(self, 'found_inf', None) is None

    def __setstate__(self, state):
        super().__setstate__(state)
        self._btr_lib = None
        for g in self.param_groups:
            g.pop('_btr_fast', None)

    def state_dict(self):
        sd = super().state_dict()
        for g in sd['param_groups']:
            g.pop('_btr_fast', None)
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._btr_lib = None
        for g in self.param_groups:     # the loaded state tensors are new objects
            g.pop('_btr_fast', None)


class FastAdam(_FastStep, torch.optim.Adam):
    """torch.optim.Adam with the lean step() of _FastStep."""


class FastAdamW(_FastStep, torch.optim.AdamW):
    """torch.optim.AdamW (decoupled weight decay) with the lean step() of _FastStep."""
    _decoupled = True


def make_optimizer(net, lr=1e-3, weight_decay=0.0, capturable=False):
    """Adam, lr 1e-3 (train_Votenet_FSB.py:172).  On the GPU the fused multi-tensor
    implementation (same update rule, two launches instead of ~10 per step) behind FastAdam.
    `capturable`: step counter on the device, needed inside a HIP graph (GraphedPipelinedStep)."""
    params = list(net.parameters())
    fused = bool(params) and all(p.is_cuda for p in params)
    cls = FastAdam if fused and os.environ.get("BTR_FAST_ADAM", "1") != "0" else torch.optim.Adam
    return cls(params, lr=lr, weight_decay=weight_decay, fused=fused,
               capturable=bool(capturable and fused))


# ------------------------------------------------ epoch-level schedules of the training scripts
def get_current_lr(epoch, base_lr=0.001, decay_steps=(80, 120, 160), decay_rates=(0.1, 0.1, 0.1)):
    """Step schedule of train_Votenet_*.py (:191-196; defaults :55-60)."""
    lr = base_lr
    for step, rate in zip(decay_steps, decay_rates):
        if epoch >= step:
            lr *= rate
    return lr


def adjust_learning_rate(optimizer, epoch, **schedule):
    """train_Votenet_FSB.py:198-201."""
    lr = get_current_lr(epoch, **schedule)
    for group in optimizer.param_groups:
        group['lr'] = lr
    return lr


def make_bn_momentum_scheduler(net, start_epoch=0, bn_decay_step=20, bn_decay_rate=0.5,
                               momentum_init=0.5, momentum_max=0.001):
    """The BatchNorm-momentum decay of train_Votenet_*.py (:186-189): 0.5 halved every 20
    epochs, floor 0.001; `.step()` once per epoch.  The fused set-abstraction layers read
    `bn.momentum` at every call, so they follow the scheduler like torch's own modules."""
    from ..pointnet2.pytorch_utils import BNMomentumScheduler

    def bn_lambda(it):
        return max(momentum_init * bn_decay_rate ** (int(it / bn_decay_step)), momentum_max)
    return BNMomentumScheduler(net, bn_lambda=bn_lambda, last_epoch=start_epoch - 1)


def enable_conv_autotune():
    """`torch.backends.cudnn.benchmark = True` like the reference's GroupFree3D scripts
    (train_GF_FSB.py:454-455): on ROCm MIOpen then times its solvers for every new convolution
    shape the first time it sees it and uses the fastest from then on (20-60 s of search per
    process on a fresh machine).  No eager step of this package runs a stock convolution any
    more -- the 1x1 conv chains, domain classifiers and heads run on the point-wise chain
    kernels -- so this only matters for a step captured into a HIP graph (`bench.py --graph`),
    which keeps its chains of < 2 048 rows on the stock ops.  (Rounds 1 - 5 shipped a MIOpen
    find-db and a TunableOp file for those paths; removed in round 6 with the eager loop the
    default everywhere.)  Call before the first convolution runs."""
    torch.backends.cudnn.enabled = True
    torch.backends.cudnn.benchmark = True


def freeze_gc():
    """Call once after the model, the optimizer state and the first (warm-up) steps exist.
    A step creates thousands of short-lived Python objects; every ~70 000 of them CPython runs
    a full (generation-2) collection that walks every tracked object of the process -- model,
    optimizer state, module dicts -- and costs several milliseconds of pure host time: 6 ms
    per Back-to-Reality step (26.3 -> 20.2 ms/step, tools/br_times.py).  `gc.freeze()` moves
    everything alive now into the permanent generation, so later collections only look at
    what the steps themselves allocate."""
    import gc
    gc.collect()
    gc.freeze()


def train_step(net, optimizer, batch, cfg, sampling=None, next_batch=None, criterion=None):
    """One optimisation step on `batch` (dict of tensors already on the model's device).
    Returns (loss tensor, end_points).  No host synchronisation inside (the reference's
    `.item()` statistics, FSB:234-237, are left to the caller).

    Optional software pipelining for loops that already hold the NEXT batch (what a data
    loader with prefetch gives): pass `next_batch` and the sampling pyramid of that batch
    (coordinates only, independent of the weights) is launched on the side stream under this
    step's backward; the returned end_points['next_sampling'] is then passed as `sampling` to
    the next call.  Results are identical to the unpipelined loop (bench.py times this loop and
    reports the strictly sequential one beside it).

    `criterion`: `loss_helper.get_loss` (default, train_Votenet_FSB.py) or
    `loss_helper.get_loss_weak` (the weakly supervised baseline, train_Votenet_WSB.py:170)."""
    _zero_grad(net, optimizer)
    inputs = {'point_clouds': batch['point_clouds']}
    if sampling is not None:
        inputs['sampling'] = sampling
    where = os.environ.get("BTR_PREFETCH_AT", "forward")
    early = next_batch is not None and where == "forward"
    nxt_sampling = None
    core = net.module if hasattr(net, "module") else net
    if early:   # the next pyramid is issued before this step's forward: 4 ms of dependent FPS
        # steps on 8 CUs then have the whole step to hide under (issued at the backward it ended
        # 0.2 ms before the next forward needed it; BTR_PREFETCH_AT=backward: 6.05 vs 6.00 ms)
        nxt_sampling = core.backbone_net.prefetch_sampling(next_batch['point_clouds'])
    fork = None
    if next_batch is not None and where == "sa2" and sampling is not None:
        # BTR_PREFETCH_AT=sa2 (round 4, measured, NOT the default): the next pyramid starts once
        # THIS forward is past SA level BTR_FORK_LEVEL (2) -- the idea: its FPS chain wants L2
        # latency, SA1 / SA2 want HBM bandwidth, and beside each other both lose (FPS 2.1 ->
        # 2.8 ms, the GEMMs +19 %).  Same box, 20 steps: before the forward 4.69 ms, behind SA1
        # 4.68, SA2 4.72, SA3 4.75, SA4 4.80, before the backward 5.00 (tools/ab_prefetch.sh):
        # the FPS runs 0.13 ms faster the later it starts, the step does not -- what the two
        # streams cost each other is paid wherever they overlap
        fork = core.backbone_net.arm_fork_event(batch['point_clouds'])
        if fork is None:
            nxt_sampling = core.backbone_net.prefetch_sampling(next_batch['point_clouds'])
    try:
        end_points = net(inputs)
    finally:
        # the library holds the event's raw handle until a native forward consumes it: never
        # leave it armed past this forward (an exception / another path would leave a dangling
        # handle for the thread's next forward to record)
        if fork is not None:
            core.backbone_net.disarm_fork_event(batch['point_clouds'])
    if fork is not None:
        nxt_sampling = core.backbone_net.prefetch_sampling(next_batch['point_clouds'], after=fork)
    for key in batch:
        assert key not in end_points
        end_points[key] = batch[key]
    loss, end_points = (criterion or loss_helper.get_loss)(end_points, cfg)
    if next_batch is not None and nxt_sampling is None:   # (BTR_PREFETCH_AT=backward)
        nxt_sampling = core.backbone_net.prefetch_sampling(next_batch['point_clouds'])
    if nxt_sampling is not None:
        end_points['next_sampling'] = nxt_sampling
    backward(loss)
    _sync_grads(net)
    optimizer.step()
    return loss, end_points


def pyramids_in_flight(pointcloud):
    """How many sampling pyramids train_one_epoch keeps in flight for clouds of this size: 1, or 2
    for scenes of more than 65 536 points.  A pyramid is a chain of ~2 000 dependent arg-max steps
    on ONE CU per scene, so two of them run beside each other at the speed of one; that pays
    exactly when the pyramid, not the step's own kernels, paces the loop -- the Matterport-shaped
    workload (4 x 80 000 points: 3.3 ms of FPS in the loop against 2.9 ms of main-stream work):
    3.90 -> 3.00 ms per step with two in flight; at 8 x 40 000 the main stream paces the loop and a
    second pyramid only costs (3.64 -> 3.73 ms; tools/ab_two_pyramids.py, profiles/
    r06_w_two_pyramids.txt)."""
    return 2 if pointcloud.is_cuda and pointcloud.shape[1] > 65536 else 1


def train_one_epoch(net, optimizer, batches, cfg, criterion=None, on_step=None, depth=None):
    """The training loop of train_Votenet_FSB.py:211-244 over `batches` (an iterable of batch
    dicts resident on the model's device), software-pipelined: while step i runs, the sampling
    pyramids of the next `depth` batches -- coordinates only, independent of the weights -- are
    computed on side streams (backbone_net.prefetch_sampling(slot=), train_step(sampling=)); the
    first `depth` pyramids are computed at the head of the loop.  `depth`: None = by the cloud
    size (pyramids_in_flight).  Same results as calling train_step batch by batch.  This is the
    loop bench.py times.  `on_step(i, loss, end_points)`: the caller's statistics hook (the
    reference's `.item()` logging, FSB:234-237); nothing here synchronises with the device.
    Returns the last (loss, end_points)."""
    import collections
    it = iter(batches)
    first = next(it, None)
    if first is None:
        return None
    core = net.module if hasattr(net, "module") else net
    if depth is None:
        depth = pyramids_in_flight(first['point_clouds'])
    depth = max(1, int(depth))
    ahead, issued = collections.deque(), [0]

    def issue(batch):
        handle = core.backbone_net.prefetch_sampling(batch['point_clouds'],
                                                     slot=issued[0] % depth)
        issued[0] += 1
        ahead.append((batch, handle))

    issue(first)
    while len(ahead) < depth:
        nxt = next(it, None)
        if nxt is None:
            break
        issue(nxt)
    out, i = None, 0
    while ahead:
        cur, sampling = ahead.popleft()
        nxt = next(it, None)
        if nxt is not None:   # issued before this step's forward: the whole step to hide under
            issue(nxt)
        out = train_step(net, optimizer, cur, cfg, sampling=sampling, criterion=criterion)
        if on_step is not None:
            on_step(i, out[0], out[1])
        i += 1
    return out


class GraphedPipelinedStep(object):
    """The software-pipelined training step as ONE HIP graph, replayed per step.

    Why: the eager step is ~400 launches whose enqueue costs the host 6.5 ms, as much as the
    GPU needs for the pipelined step -- the loop is on the edge of being host-bound and every
    further kernel gain would be invisible.  A replay costs the host ~20 us.

    What is captured (main stream unless noted):
        p_in <- p_out                 (the sampling pyramid the PREVIOUS replay produced)
        zero_grad, forward(cur, sampling = p_in), get_loss
        side stream: sampling pyramid of `nxt_pc` (FPS, 4 levels)  || backward, Adam
        join; p_out <- that pyramid
    `cur` (a whole batch dict) and `nxt_pc` (the NEXT batch's point clouds) are static buffers
    filled by __call__ before each replay -- what a prefetching loader would fill.  Nothing
    synchronises with the host; shapes are fixed; Adam is `capturable`.  `prime(batch)`
    computes the pyramid of the first batch eagerly (the head of the loop).
    Single-process only: under FlatGradParallel / DDP the eager pipelined loop is used.

    Frozen at capture (a replay runs the kernels with the arguments they were recorded with):
    python-float hyper-parameters -- the learning rate `adjust_learning_rate` sets and the
    BatchNorm momentum (a kernel argument of the fused layers): the reference's per-epoch
    decays (train_Votenet_FSB.py:186-201) need a re-capture (a new object) at each change.
    Constructing the object runs `warmup` + 1 REAL optimisation steps on `batch` (weights, Adam
    state and BatchNorm running statistics move): a caller that needs the initial state back
    must restore it (tests/test_configs_gpu.py does)."""

    def __init__(self, net, optimizer, batch, next_batch, cfg, warmup=3, criterion=None):
        assert not hasattr(net, "module"), "graph capture of the data-parallel step: not supported"
        self.net, self.optimizer, self.cfg = net, optimizer, cfg
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
            loss, end = train_step(net, optimizer, self.cur, cfg,
                                   sampling=[(t, None) for t in self.p_in],
                                   next_batch={'point_clouds': self.nxt_pc}, criterion=criterion)
            torch.cuda.current_stream().wait_stream(
                bb._get_side_stream(self.cur['point_clouds'].device, "_prefetch_stream"))   # join the side stream
            for dst, (inds, _) in zip(self.p_out, end['next_sampling']):
                dst.copy_(inds)
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), fused_backbone.layerwise():
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step()

    def prime(self, batch):
        """Head of the loop: the sampling pyramid of the first batch (eager, current stream)."""
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
        return self.loss


def _source_inputs(batch_S, sampling_S):
    inputs = {'point_clouds': batch_S['point_clouds']}
    if sampling_S is not None:
        inputs['sampling'] = sampling_S
    return inputs


def _prefetch_next(core, end_points_S, end_points_T, next_batch_S, next_batch_T):
    """next step's pyramids on the side stream, source first (its forward runs first)"""
    if next_batch_S is not None:
        end_points_S['next_sampling'] = core.backbone_net.prefetch_sampling(
            next_batch_S['point_clouds'])
    if next_batch_T is not None:
        end_points_T['next_sampling'] = core.backbone_net.prefetch_sampling(
            next_batch_T['point_clouds'])


def train_step_br(net, optimizer, batch_S, batch_T, cfg, sampling_S=None, next_batch_S=None,
                  sampling_T=None, next_batch_T=None):
    """One Back-to-Reality step (detection/Votenet/train_Votenet_BR.py:267-289): the SAME
    VoteNet_DA runs a source (virtual scenes) and a target (real scenes) forward -- BatchNorm
    running statistics are updated twice -- then one `get_loss_DA`, one backward, one Adam
    step.  Two scenes batches = 2 x batch scenes of hot-path work per step.
    `sampling_S/_T`, `next_batch_S/_T`: software pipelining across steps as in train_step --
    the NEXT step's pyramids run under this step's backward and come back as
    end_points_S/_T['next_sampling']."""
    _zero_grad(net, optimizer)
    # (unpipelined:) the target branch's sampling pyramid (coordinates only) runs on the side
    # stream under the source branch's forward
    core = net.module if hasattr(net, "module") else net
    if sampling_T is None:
        sampling_T = core.backbone_net.prefetch_sampling(batch_T['point_clouds'])
    early = os.environ.get("BTR_PREFETCH_AT", "forward") == "forward"
    nxt = ({}, {})
    if early:
        _prefetch_next(core, nxt[0], nxt[1], next_batch_S, next_batch_T)
    with grad_sinks(net):
        end_points_S, end_points_T = two_forwards(
            net, (_source_inputs(batch_S, sampling_S),),
            ({'point_clouds': batch_T['point_clouds'], 'sampling': sampling_T},))
    for key in batch_S:
        end_points_S[key] = batch_S[key]
    for key in batch_T:
        end_points_T[key] = batch_T[key]
    loss, end_points_S, end_points_T = loss_helper.get_loss_DA(end_points_S, end_points_T, cfg)
    if not early:
        _prefetch_next(core, nxt[0], nxt[1], next_batch_S, next_batch_T)
    end_points_S.update(nxt[0])
    end_points_T.update(nxt[1])
    backward(loss)
    _sync_grads(net)
    optimizer.step()
    return loss, end_points_S, end_points_T


def train_step_br_jitter(net, optimizer, batch_S, batch_T, cfg, epoch=0, sampling_S=None,
                         next_batch_S=None, sampling_T=None, next_batch_T=None):
    """One CenterRefine step (detection/Votenet/train_Votenet_BR_CenterRefine.py:254-276): like
    train_step_br, but both forwards also pool features around the (noisy) GT centres and
    regress their displacement; batches need 'center_jitter' (synthetic.make_batch(...,
    center_jitter=0.1))."""
    _zero_grad(net, optimizer)
    core = net.module if hasattr(net, "module") else net
    if sampling_T is None:
        sampling_T = core.backbone_net.prefetch_sampling(batch_T['point_clouds'])
    early = os.environ.get("BTR_PREFETCH_AT", "forward") == "forward"
    nxt = ({}, {})
    if early:
        _prefetch_next(core, nxt[0], nxt[1], next_batch_S, next_batch_T)
    with grad_sinks(net):
        end_points_S, end_points_T = two_forwards(
            net, (_source_inputs(batch_S, sampling_S), batch_S['center_label'],
                  batch_S['sem_cls_label']),
            ({'point_clouds': batch_T['point_clouds'], 'sampling': sampling_T},
             batch_T['center_label'], batch_T['sem_cls_label']))
    for key in batch_S:
        end_points_S[key] = batch_S[key]
    for key in batch_T:
        end_points_T[key] = batch_T[key]
    loss, end_points_S, end_points_T = loss_helper.get_loss_DA_jitter(
        end_points_S, end_points_T, epoch, cfg)
    if not early:
        _prefetch_next(core, nxt[0], nxt[1], next_batch_S, next_batch_T)
    end_points_S.update(nxt[0])
    end_points_T.update(nxt[1])
    backward(loss)
    _sync_grads(net)
    optimizer.step()
    return loss, end_points_S, end_points_T


# ------------------------------------------------------------------ checkpoints
EVAL_CONFIG_DICT = {'remove_empty_box': False, 'use_3d_nms': True, 'nms_iou': 0.25,
                    'use_old_type_nms': False, 'cls_nms': True, 'per_class_proposal': True,
                    'conf_thresh': 0.05}  # train_Votenet_FSB.py:204-207


def evaluate_one_epoch(net, batches, cfg, config_dict=None, ap_iou_thresh=0.25,
                       class2type_map=None):
    """The evaluation pass of the training scripts (train_Votenet_FSB.py:246-293): eval-mode
    forward, loss statistics, parse_predictions / parse_groundtruths, AP.  `batches`: an
    iterable of label dicts on the model's device.  Returns (mean stats dict, metrics dict).
    The statistics stay on the device until the end (one host read per key instead of the
    reference's `.item()` per key per batch)."""
    from . import ap_helper
    config_dict = dict(config_dict or EVAL_CONFIG_DICT, dataset_config=cfg)
    calc = ap_helper.APCalculator(ap_iou_thresh=ap_iou_thresh, class2type_map=class2type_map)
    was_training = net.training
    net.eval()
    stat, nb = {}, 0
    try:
        for batch in batches:
            with torch.no_grad():
                end_points = net({'point_clouds': batch['point_clouds']})
                for key in batch:
                    assert key not in end_points
                    end_points[key] = batch[key]
                _, end_points = loss_helper.get_loss(end_points, cfg)
            for key, v in end_points.items():
                if ('loss' in key or 'acc' in key or 'ratio' in key) and torch.is_tensor(v):
                    stat[key] = stat.get(key, 0) + v.detach()
            calc.step(ap_helper.parse_predictions(end_points, config_dict),
                      ap_helper.parse_groundtruths(end_points, config_dict))
            nb += 1
    finally:
        net.train(was_training)
    stats = {k: float(v) / max(nb, 1) for k, v in sorted(stat.items())}
    return stats, calc.compute_metrics()


def save_checkpoint(path, net, optimizer, epoch, loss=None):
    """`checkpoint.tar` in the reference's wire format (train_Votenet_FSB.py:310-318): keys
    'epoch' (the NEXT epoch to run), 'optimizer_state_dict', 'loss', 'model_state_dict' (of the
    bare module when wrapped in DataParallel / DistributedDataParallel)."""
    core = net.module if hasattr(net, "module") else net
    torch.save({'epoch': int(epoch) + 1,
                'optimizer_state_dict': optimizer.state_dict(),
                'loss': loss,
                'model_state_dict': core.state_dict()}, path)


def load_checkpoint(path, net, optimizer=None, map_location="cpu"):
    """Counterpart of train_Votenet_FSB.py:174-181; accepts checkpoints written by the
    reference (same parameter / buffer names, see the golden state signatures).  Returns the
    epoch to resume from."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    core = net.module if hasattr(net, "module") else net
    core.load_state_dict(ckpt['model_state_dict'])
    if optimizer is not None and ckpt.get('optimizer_state_dict') is not None:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return int(ckpt.get('epoch', 0))
