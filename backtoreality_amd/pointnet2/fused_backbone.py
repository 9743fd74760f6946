"""A PointNet++ encoder-decoder (set-abstraction levels + feature-propagation modules) as THREE
library calls per training step: `btr_backbone_sampling` / `_forward` / `_backward`
(csrc/backbone.hip), instead of one autograd node, a ball query, a gather and a dozen
allocator calls per layer.

Computes what `Pointnet2Backbone.forward` computes in the reference
(detection/Votenet/models/backbone_module.py:83-133) through its `PointnetSAModuleVotes`
(pointnet2/pointnet2_modules.py:210-272) and `PointnetFPModule` (:469-514) layers, train-mode
BatchNorm, forward and backward.  The layer-by-layer path (pointnet2_modules.py of this
package: one `FusedSALayer` / `PointwiseChain` node per layer) stays the readable statement of
the same launches and the test oracle of this one: identical kernels, identical results
(tests/test_backbone_gpu.py).  `BTR_NATIVE_BACKBONE=0` selects it.

Why it exists: after the per-layer C calls the VoteNet step still cost the host ~5 ms of
Python per 4.5 ms of GPU work (profiles/r02_h_bench_steps20.json: host_enqueue 5.3 of 5.6 ms).
"""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function

if __package__:
    from . import _ext, fused_mlp, fused_sa, grad_sink
else:
    import pointnet2._ext as _ext
    import fused_mlp
    import fused_sa
    import grad_sink
_p = _ext._p


_LAYERWISE = [0]


class layerwise(object):
    """`with layerwise():` -- the whole-backbone calls are off inside (the layer-by-layer path
    runs, as it does while a HIP graph is captured).  The warm-up steps in front of a capture
    run under it, so that every stock op of the captured path (MIOpen / hipBLASLt solution
    searches, allocator pools) has been seen before the capture begins."""

    def __enter__(self):
        _LAYERWISE[0] += 1
        fused_mlp._CAPTURE_PATH[0] += 1
        return self

    def __exit__(self, *exc):
        _LAYERWISE[0] -= 1
        fused_mlp._CAPTURE_PATH[0] -= 1
        return False


def enabled():
    return (os.environ.get("BTR_NATIVE_BACKBONE", "1") != "0" and not _LAYERWISE[0] and
            fused_sa.enabled() and fused_mlp.enabled() and fused_sa.native_enabled())


def _momentum(bn):
    if bn.momentum is not None:
        return float(bn.momentum)
    return 1.0 / float(bn.num_batches_tracked.item() + 1)


class Entry(object):
    """Description + plan of one backbone at one input shape (cached by the module)."""

    def __init__(self, sa_modules, fp_modules, B, N, C):
        lib = _ext._idx
        d = _ext.Backbone()
        d.b, d.n, d.c = B, N, C
        d.levels, d.fps = len(sa_modules), len(fp_modules)
        n, c = N, C
        self.sa_bns, self.fp_bns = [], []
        self.sa_params, self.fp_params = [], []
        for l, m in enumerate(sa_modules):
            g = m.grouper
            s = d.sa[l]
            s.b, s.n, s.m, s.s, s.c = B, n, m.npoint, g.nsample, c
            s.use_xyz = 1 if g.use_xyz else 0
            s.radius_div = float(g.radius if g.normalize_xyz else 1.0)
            d.radius[l] = float(g.radius)
            layers = list(m.mlp_module)
            s.layers = len(layers)
            for i, layer in enumerate(layers):
                s.width[i] = layer.conv.weight.shape[0]
                s.eps[i] = float(layer.bn.bn.eps)
            s.need_dxyz = s.need_dnew_xyz = 0
            s.need_dfeat = 1 if l > 0 else 0
            s.options = fused_sa._sa_options()
            self.sa_bns.append([layer.bn.bn for layer in layers])
            self.sa_params.append([t for layer in layers for t in
                                   (layer.conv.weight, layer.bn.bn.weight, layer.bn.bn.bias)])
            n, c = m.npoint, s.width[len(layers) - 1]
        self.level_n = [N] + [m.npoint for m in sa_modules]
        self.level_c = [C] + [d.sa[l].width[d.sa[l].layers - 1] for l in range(d.levels)]
        L = d.levels
        for j, m in enumerate(fp_modules):
            f = d.fp[j]
            layers = list(m.mlp)
            c_known = self.level_c[L] if j == 0 else d.fp[j - 1].width[d.fp[j - 1].layers - 1]
            f.b, f.n = B, self.level_n[L - j - 1]
            f.c = c_known + self.level_c[L - j - 1]
            f.layers = len(layers)
            for i, layer in enumerate(layers):
                f.width[i] = layer.conv.weight.shape[0]
                f.has_bn[i] = 1
                f.eps[i] = float(layer.bn.bn.eps)
            f.need_dx = 1
            self.fp_bns.append([layer.bn.bn for layer in layers])
            self.fp_params.append([t for layer in layers for t in
                                   (layer.conv.weight, layer.conv.bias, layer.bn.bn.weight,
                                    layer.bn.bn.bias)])
        self.fp_c = [d.fp[j].width[d.fp[j].layers - 1] for j in range(d.fps)]
        plan = _ext.BackbonePlan()
        _ext._call(lib.btr_backbone_plan, ctypes.addressof(d), ctypes.addressof(plan))
        self.d, self.plan, self.lib = d, plan, lib
        self.B, self.N, self.C = B, N, C
        # how the flat gradient buffer splits into the parameters' gradients
        self.grad_views = []   # (float offset, padded shape, parameter shape) per parameter slot
        for l in range(d.levels):
            sp, s = plan.sa[l], d.sa[l]
            for i in range(s.layers):
                w = self.sa_params[l][3 * i]
                base = plan.gr_sa[l]
                # (a first layer's gradient is written as dense [n][3 + c] rows, without the
                # columns its 4-aligned input width added: csrc/sa_layer.hip reduce_unpad_next)
                kin = w.shape[1] if (i == 0 and not sp.recompute) else sp.kin[i]
                self.grad_views += [(base + sp.dw[i], (s.width[i], kin), tuple(w.shape)),
                                    (base + sp.dgamma[i], (s.width[i],), None),
                                    (base + sp.dbeta[i], (s.width[i],), None)]
        for j in range(d.fps):
            fp, f = plan.fp[j], d.fp[j]
            for i in range(f.layers):
                w, bias = self.fp_params[j][4 * i], self.fp_params[j][4 * i + 1]
                base = plan.gr_fp[j]
                self.grad_views += [(base + fp.dw[i], (fp.np[i], fp.kin[i]), tuple(w.shape)),
                                    (base + fp.dbias[i], (fp.np[i],), None) if bias is not None
                                    else None,
                                    (base + fp.dgamma[i], (fp.np[i],), None),
                                    (base + fp.dbeta[i], (fp.np[i],), None)]
        self.params = [t for ps in self.sa_params for t in ps] + \
                      [t for ps in self.fp_params for t in ps]
        self._sink = None

    def views(self, grads):
        """The parameters' gradients as views of the call's flat gradient buffer."""
        res = []
        for view in self.grad_views:
            if view is None:
                res.append(None)
                continue
            off, padded, shape = view
            if len(padded) == 1:
                res.append(grads.as_strided(padded, (1,), off))
                continue
            n, k = (shape[0], shape[1]) if shape is not None else padded
            res.append(grads.as_strided((n, k), (padded[1], 1), off).reshape(shape))
            # (a level's first layer arrives as dense [n][k] rows -- padded[1] == k in
            # grad_views -- so no view here is copied by the reshape)
        return res

    def sink(self, device):
        """The flat gradient sink of this backbone (grad_sink.py), or None outside its scope."""
        if not grad_sink.active() or not grad_sink.all_leaves(self.params):
            return None
        if self._sink is None or self._sink.tensor.device != device:
            self._sink = grad_sink.Sink(self.plan.grads_floats, device, self.views)
        return self._sink.bind(self.params)

    def bind(self):
        """Refresh the parameter / buffer pointers and BatchNorm momenta of the description
        (parameters may have been moved or replaced since the last call)."""
        d = self.d
        for l in range(d.levels):
            s, ps, bns = d.sa[l], self.sa_params[l], self.sa_bns[l]
            for i, bn in enumerate(bns):
                s.w[i] = ps[3 * i].data_ptr()
                s.gamma[i] = ps[3 * i + 1].data_ptr()
                s.beta[i] = ps[3 * i + 2].data_ptr()
                track = bn.track_running_stats and bn.running_mean is not None
                s.running_mean[i] = bn.running_mean.data_ptr() if track else None
                s.running_var[i] = bn.running_var.data_ptr() if track else None
                s.num_batches_tracked[i] = bn.num_batches_tracked.data_ptr() if track else None
                s.momentum[i] = _momentum(bn)
        for j in range(d.fps):
            f, ps, bns = d.fp[j], self.fp_params[j], self.fp_bns[j]
            for i, bn in enumerate(bns):
                f.w[i] = ps[4 * i].data_ptr()
                f.bias[i] = _p(ps[4 * i + 1])
                f.gamma[i] = ps[4 * i + 2].data_ptr()
                f.beta[i] = ps[4 * i + 3].data_ptr()
                track = bn.track_running_stats and bn.running_mean is not None
                f.running_mean[i] = bn.running_mean.data_ptr() if track else None
                f.running_var[i] = bn.running_var.data_ptr() if track else None
                f.num_batches_tracked[i] = bn.num_batches_tracked.data_ptr() if track else None
                f.momentum[i] = _momentum(bn)


_STATIC_OK = weakref.WeakKeyDictionary()   # first SA module -> {training flags: bool}


def _bns(sa_modules, fp_modules):
    for m in sa_modules:
        for layer in m.mlp_module:
            yield layer.bn.bn
    for m in fp_modules:
        for layer in m.mlp:
            yield layer.bn.bn


def _static_ok(sa_modules, fp_modules, probe):
    """The part of supported() that only depends on the modules: evaluated once per
    combination of training flags (it walks every layer: ~0.2 ms per call otherwise)."""
    import torch.nn as nn
    if not 1 <= len(sa_modules) <= _ext.MAX_LEVELS or len(fp_modules) >= len(sa_modules):
        return False
    try:
        key = tuple(m.training for m in sa_modules) + \
            tuple(bn.training for bn in _bns(sa_modules, fp_modules))
    except AttributeError:   # a layer without conv / bn children: not a covered configuration
        return False
    cache = _STATIC_OK.get(sa_modules[0])
    if cache is None:
        cache = _STATIC_OK[sa_modules[0]] = {}
    ok = cache.get(key)
    if ok is not None:
        return ok
    ok = True
    for l, m in enumerate(sa_modules):
        if not m.training or m.npoint is None or getattr(m, "ret_unique_cnt", False) or \
                not fused_sa.can_fuse(m, probe, None) or len(m.mlp_module) > _ext.MAX_LAYERS or \
                any(layer.conv.weight.shape[0] % 4 for layer in m.mlp_module):
            ok = False
    for m in fp_modules:
        chain = fused_mlp.shared_mlp_chain(m.mlp)
        if chain is None or len(chain) > _ext.MAX_LAYERS:
            ok = False
            continue
        for conv, bn, _ in chain:
            if not isinstance(bn, (nn.BatchNorm1d, nn.BatchNorm2d)) or bn.weight is None or \
                    not bn.training or conv.out_channels % 4 or conv.out_channels > 512 or \
                    conv.kernel_size not in ((1,), (1, 1)) or conv.groups != 1:
                ok = False
    cache[key] = ok
    return ok


def supported(sa_modules, fp_modules, pointcloud):
    """True when every layer is in the configuration the fused kernels cover (training mode,
    ball-query grouping + max-pool, conv -> BatchNorm -> ReLU MLPs) and the cloud is a
    contiguous CUDA f32 tensor that needs no gradient."""
    pc = pointcloud
    if not (enabled() and pc.is_cuda and pc.dtype == torch.float32 and pc.dim() == 3 and
            pc.is_contiguous() and pc.size(-1) >= 3 and not pc.requires_grad):
        return False
    if not torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
        return False
    if not sa_modules[0].grouper.use_xyz and pc.size(-1) == 3:
        return False
    return _static_ok(sa_modules, fp_modules, pc[:, :1, :3])


class Sampling(object):
    """What `btr_backbone_sampling` computed for one cloud: FPS indices, centre coordinates,
    ball-query lists per level, 3-NN blend weights per propagation module -- one arena.
    Iterating yields (inds, ready event) per level, the format the layer-by-layer path takes."""

    def __init__(self, entry, pointcloud, geom, event, wait_side):
        self.entry, self.geom, self.event, self.wait_side = entry, geom, event, wait_side
        self.cloud, self.version = pointcloud, pointcloud._version
        p, d = entry.plan, entry.d
        B = entry.B
        self.inds, self.xyz = [], []
        for l in range(d.levels):
            m = d.sa[l].m
            self.inds.append(geom.as_strided((B, m), (m, 1), p.g_inds[l] // 4))
            self.xyz.append(geom.as_strided((B, m, 3), (3 * m, 3, 1),
                                            p.g_new_xyz[l] // 4).view(torch.float32))

    def idx(self, level):
        """Ball-query lists (B, m, nsample) i32 of SA level `level` (0-based)."""
        s = self.entry.d.sa[level]
        return self.geom.as_strided((self.entry.B, s.m, s.s), (s.m * s.s, s.s, 1),
                                    self.entry.plan.g_idx[level] // 4)

    def three_nn(self, module):
        """(idx (B, n, 3) i32, weight (B, n, 3) f32) of feature-propagation module `module`."""
        n = self.entry.d.fp[module].n
        shape, stride = (self.entry.B, n, 3), (3 * n, 3, 1)
        p = self.entry.plan
        return (self.geom.as_strided(shape, stride, p.g_nn_idx[module] // 4),
                self.geom.as_strided(shape, stride, p.g_nn_w[module] // 4).view(torch.float32))

    def valid_for(self, entry, pointcloud):
        return (self.entry is entry and self.cloud is pointcloud and
                self.version == pointcloud._version)

    def __iter__(self):
        return iter([(i, self.event) for i in self.inds])

    def __len__(self):
        return len(self.inds)

    def __getitem__(self, i):
        return (self.inds[i], self.event)


def sample(entry, pointcloud, stream=None, side=None):
    """Run the sampling of `pointcloud` on `stream` (default: the current one).  `side`: a
    torch stream for the levels behind the first (sequential mode, see the header)."""
    dev = pointcloud.device
    geom = torch.empty((entry.plan.geom_bytes // 4,), dtype=torch.int32, device=dev)
    with _ext._on(pointcloud) as dv:
        cur = stream if stream is not None else torch.cuda.current_stream(dv)
        if _ext._TIMING is not None:   # bench.py: event pairs around the large-scene FPS kernel
            s0 = entry.d.sa[0]         # and the first level's ball query, recorded by the library
            for op, key, hook in (
                    ("fps_kernel", (entry.B, s0.n, s0.m), entry.lib.btr_fps_time_next_kernel),
                    ("ball_query_buckets" if entry.plan.bq_buckets[0] else "ball_query",
                     (entry.B, s0.n, s0.m, s0.s), entry.lib.btr_ball_query_time_next)):
                if s0.n > 4096 and (_ext._TIMING_FILTER is None or _ext._TIMING_FILTER(op, key)):
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(cur):
                        e0.record()   # (creates the handles; the library records them in place)
                        e1.record()
                    hook(e0.cuda_event, e1.cuda_event)
                    _ext._TIMING.append((op, key, e0, e1))
        _ext._call(entry.lib.btr_backbone_sampling, ctypes.addressof(entry.d),
                   ctypes.addressof(entry.plan), _p(pointcloud), _p(geom), cur.cuda_stream,
                   side.cuda_stream if side is not None else None)
    if side is not None:
        geom.record_stream(side)
    return Sampling(entry, pointcloud, geom, None, side is not None)


class FusedBackboneFn(Function):
    """forward(cloud, sampling, entry, sink, *params) -> (SA outputs..., FP outputs...) as (B, C, M)
    views of one arena; their channel-last twins ((B, M, C) for the levels, (B*N, C) for the
    modules, as the layer-by-layer path attaches them) are left in `entry.last_twins`."""

    @staticmethod
    def forward(ctx, cloud, sampling, entry, sink, *params):
        _ext.RUNNING_STATS_EPOCH[0] += 1   # running statistics move through raw pointers
        d, plan = entry.d, entry.plan
        dev = cloud.device
        ctx.to_sink = sink is not None
        entry.bind()
        out = torch.empty((plan.out_bytes // 4,), dtype=torch.float32, device=dev)
        saved = torch.empty((plan.saved_bytes,), dtype=torch.uint8, device=dev)
        scratch = torch.empty((max(plan.fwd_scratch_bytes, 1),), dtype=torch.uint8, device=dev)
        with _ext._on(cloud) as dv:
            _ext._call(entry.lib.btr_backbone_forward, ctypes.addressof(d),
                       ctypes.addressof(plan), _p(cloud), _p(sampling.geom), _p(out), _p(saved),
                       _p(scratch), 1 if sampling.wait_side else 0, _ext._stream(dv))
        B = entry.B
        outs, twins = [], []
        for l in range(d.levels):
            c, m = entry.level_c[l + 1], entry.level_n[l + 1]
            outs.append(out.as_strided((B, c, m), (c * m, m, 1), plan.o_sa[l] // 4))
            twins.append(out.as_strided((B, m, c), (c * m, c, 1), plan.o_sa_cl[l] // 4))
        for j in range(d.fps):
            c, n = entry.fp_c[j], d.fp[j].n
            outs.append(out.as_strided((B, c, n), (c * n, n, 1), plan.o_fp[j] // 4))
            twins.append(out.as_strided((B * n, c), (c, 1), plan.o_fp_cl[j] // 4))
        ctx.entry = entry
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(sampling.geom, out, saved)
        entry.last_twins = twins   # channel-last twins of the outputs, for the caller
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        entry = ctx.entry
        d, plan = entry.d, entry.plan
        geom, out, saved = ctx.saved_tensors
        dev = out.device
        L, F = d.levels, d.fps
        douts = [None if g is None else g.contiguous() for g in douts]
        dsa = (ctypes.c_void_p * L)(*[_p(g) for g in douts[:L]])
        dfp = (ctypes.c_void_p * max(F, 1))(*[_p(g) for g in douts[L:L + F]])
        grads = torch.empty((plan.grads_floats,), dtype=torch.float32, device=dev)
        scratch = torch.empty((plan.bwd_scratch_bytes,), dtype=torch.uint8, device=dev)
        with _ext._on(out) as dv:
            _ext._call(entry.lib.btr_backbone_backward, ctypes.addressof(d),
                       ctypes.addressof(plan), _p(geom), _p(out), dsa, dfp, _p(saved), _p(grads),
                       _p(scratch), _ext._stream(dv))
        if ctx.to_sink:   # (one flat gradient for the sink, grad_sink.py)
            return (None, None, None, grads) + (None,) * len(entry.grad_views)
        return (None, None, None, None) + tuple(entry.views(grads))
