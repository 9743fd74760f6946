"""Fused point-wise MLP chains on MI355X: the 1x1-convolution (+bias) -> BatchNorm -> ReLU
sequences around the set-abstraction stack -- the SharedMLP of the feature-propagation
modules (reference pointnet2_modules.py:469-514, pytorch_utils.py:11-36), the vote generator
(models/voting_module.py:37-56) and the proposal head (models/proposal_module.py:75-113).

The reference runs them as stock conv / batch-norm / relu ops on (B, C, N) tensors: per layer a
GEMM per batch sample, four BatchNorm launches, an activation, and twice that in the backward
(~270 launches and ~1.1 ms of the benchmark step, as much host time as GPU time).  Here a
chain runs on channel-last rows (B*N, C) through the kernels of the fused set-abstraction
path (csrc/sa_mlp.hip): f32-MFMA GEMM on 64-row tiles with the BatchNorm statistics in its
epilogue, BN + ReLU applied while the NEXT layer's operand is staged, BN backward as two
streaming passes, weight gradient as a TN GEMM.  A conv bias in front of a train-mode
BatchNorm cancels exactly (it only moves the running mean, which is corrected), so it is
skipped and its gradient is the exact zero the mathematics gives.

`PointwiseMLP.apply(x, meta, *params)`: x (B, C, N) f32 -> (B, C_out, N); the result carries
its channel-last twin as `._btr_channel_last`.  `BTR_FUSED_MLP=0` disables the path.
"""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function

if __package__:
    from . import _ext, grad_sink
else:
    import pointnet2._ext as _ext
    import grad_sink
_call, _lib, _on, _p, _stream = _ext._call, _ext._lib, _ext._on, _ext._p, _ext._stream


def enabled():
    return os.environ.get("BTR_FUSED_MLP", "1") != "0"


def _ceil4(v):
    return (v + 3) // 4 * 4


def _gemm_nt(rows, n, k, A, lda, W, Y, pa, pb, part, bias, st):
    """Y (rows, n) = f(A) . W^T for W (n, k) contiguous: the small-M kernel on the weight's bf16
    planes where it applies (as csrc/sa_layer.hip pm_gemm_nt_auto does), else gemm_nt_kernel."""
    if _lib.btr_pm_gemm_nt_sm_supported(rows, n, k):
        planes = torch.empty((int(_lib.btr_pm_weight_planes_bytes(n, k)),), dtype=torch.uint8,
                             device=A.device)
        _call(_lib.btr_pm_weight_planes, n, k, _p(W), k, _p(planes), st)
        _call(_lib.btr_pm_gemm_nt_sm, rows, n, k, _p(A), lda, _p(planes), _p(Y), n, _p(pa), _p(pb),
              _p(part), _p(bias), st, key=(rows, n, k))
    else:
        _call(_lib.btr_pm_gemm_nt, rows, n, k, _p(A), lda, _p(W), k, _p(Y), n, _p(pa), _p(pb),
              _p(part), _p(bias), st, key=(rows, n, k))


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


class PointwiseMLP(Function):
    """meta: dict(layers=[dict(bn=nn.BatchNorm*|None, relu=bool)]); params: per layer
    (W (C_out, C_in[,1[,1]]), bias|None, gamma|None, beta|None)."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        _ext.RUNNING_STATS_EPOCH[0] += 1   # running statistics move through raw pointers
        layers = meta["layers"]
        L = len(layers)
        dev = x.device
        B, K0, N = x.shape
        rows = B * N
        assert K0 % 4 == 0, "input width must be a multiple of 4"
        with _on(x) as d:
            st = _stream(d)
            A = _ext.twin_of(x)
            if A is None or A.shape != (rows, K0):
                A = _f32((rows, K0), dev)
                _call(_lib.btr_pm_rows, B, N, K0, K0, _p(x.contiguous()), _p(A), st)
            X0 = A
            grid = _lib.btr_pm_gemm_grid(rows)
            Ys, Ws, stats, widths = [], [], [], []
            lda, K = K0, K0
            pa = pb = None
            counters = []
            for l, spec in enumerate(layers):
                W, bias, gamma, beta = params[4 * l:4 * l + 4]
                Nl = W.shape[0]
                Np = _ceil4(Nl)
                W2 = W.reshape(Nl, -1)
                assert W2.shape[1] == K
                if Np != Nl:
                    Wp = torch.zeros((Np, K), dtype=torch.float32, device=dev)
                    Wp[:Nl] = W2
                    W2 = Wp
                W2 = W2.contiguous()
                bn = spec["bn"]
                Y = _f32((rows, Np), dev)
                if bn is not None:
                    part = _f32((grid, 2, Np), dev)
                    _gemm_nt(rows, Np, K, A, lda, W2, Y, pa, pb, part, None, st)
                    scale, shift, mean, invstd = (_f32((Np,), dev) for _ in range(4))
                    mom = float(bn.momentum) if bn.momentum is not None else \
                        1.0 / float(bn.num_batches_tracked.item() + 1)
                    track = bn.track_running_stats and bn.running_mean is not None
                    _call(_lib.btr_sa_bn_finalize, Np, grid, float(rows), float(bn.eps), mom,
                          _p(part), _p(gamma), _p(beta), _p(scale), _p(shift), _p(mean),
                          _p(invstd), _p(bn.running_mean if track else None),
                          _p(bn.running_var if track else None), st)
                    if track:
                        if bias is not None:   # the skipped bias only moves the running mean
                            bn.running_mean.add_(bias.detach(), alpha=mom)
                        counters.append(bn.num_batches_tracked)
                    stats.append((scale, shift, mean, invstd))
                    assert spec["relu"], "BatchNorm without ReLU is not covered"
                    pa, pb = scale, shift
                else:
                    assert l == L - 1 and not spec["relu"], "only the last layer may lack BN"
                    bp = None
                    if bias is not None:
                        bp = bias if Np == Nl else torch.cat(
                            [bias, torch.zeros(Np - Nl, device=dev)])
                    _gemm_nt(rows, Np, K, A, lda, W2, Y, pa, pb, None, bp, st)
                    stats.append(None)
                    pa = pb = None
                Ys.append(Y)
                Ws.append(W2)
                widths.append(Nl)
                A, lda, K = Y, Np, Np
            NL = widths[-1]
            out = _f32((B, NL, N), dev)
            out_cl = _f32((rows, NL), dev)
            last = stats[-1]
            _call(_lib.btr_pm_out, B, N, NL, Ys[-1].shape[1], _p(Ys[-1]),
                  _p(last[0]) if last else None, _p(last[1]) if last else None,
                  1 if last else 0, _p(out), _p(out_cl), st)
            if counters:
                torch._foreach_add_(counters, 1)
        _ext.attach_twin(out, out_cl)
        ctx.dims = (B, N, K0, L)
        ctx.widths = widths
        ctx.pshapes = [None if p is None else p.shape for p in params]
        ctx.has_bn = [s is not None for s in stats]
        flat = [t for s4 in stats if s4 is not None for t in s4]
        ctx.save_for_backward(X0, *Ys, *Ws, *flat)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, K0, L = ctx.dims
        rows = B * N
        saved = ctx.saved_tensors
        X0 = saved[0]
        Ys = list(saved[1:1 + L])
        Ws = list(saved[1 + L:1 + 2 * L])
        flat = list(saved[1 + 2 * L:])
        stats = []
        for has in ctx.has_bn:
            stats.append(tuple(flat[:4]) if has else None)
            if has:
                flat = flat[4:]
        dev = dout.device
        grads = [None] * (4 * L)
        need_x = ctx.needs_input_grad[0]
        dx = None
        with _on(dout) as d:
            st = _stream(d)
            NL = ctx.widths[-1]
            Np = Ys[-1].shape[1]
            G = _f32((rows, Np), dev)
            _call(_lib.btr_pm_rows, B, N, NL, Np, _p(dout.contiguous()), _p(G), st)
            # `lazy` / `fusable`: the same sequence as csrc/sa_layer.hip btr_pm_chain_backward -- a
            # hidden layer behind a BatchNorm runs its whole backward as one btr_sa_bwd_fused call
            fused_on = os.environ.get("BTR_CHAIN_FUSED", "1") != "0"

            def fusable(j):
                return fused_on and j >= 1 and stats[j] is not None and bool(
                    _lib.btr_sa_bwd_fused_supported(rows, Ws[j].shape[0], Ws[j].shape[1]))
            lazy = None
            if stats[-1] is not None:
                sc, sh, mu, isd = stats[-1]
                part = _f32((1024, 2, Np), dev)
                m1, m2, dg, db = (_f32((Np,), dev) for _ in range(4))
                if fusable(L - 1):
                    _call(_lib.btr_sa_bn_relu_bwd_sums, rows, Np, Np, _p(G), _p(Ys[-1]), _p(sc),
                          _p(sh), _p(mu), _p(isd), _p(part), _p(m1), _p(m2), _p(dg), _p(db), st)
                    lazy = (m1, m2)
                else:
                    _call(_lib.btr_sa_bn_relu_bwd, rows, Np, Np, _p(G), _p(Ys[-1]), _p(sc),
                          _p(sh), _p(mu), _p(isd), _p(part), _p(m1), _p(m2), _p(dg), _p(db), st)
                grads[4 * (L - 1) + 2], grads[4 * (L - 1) + 3] = dg[:NL], db[:NL]
                if ctx.pshapes[4 * (L - 1) + 1] is not None:
                    grads[4 * (L - 1) + 1] = torch.zeros(NL, device=dev)
            elif ctx.pshapes[4 * (L - 1) + 1] is not None:
                grads[4 * (L - 1) + 1] = G.sum(0)[:NL]
            dY = G
            for l in range(L - 1, -1, -1):
                W2 = Ws[l]
                Np, K = W2.shape
                Nl = ctx.widths[l]
                if l == 0:
                    Xsrc, ldx, pa, pb = X0, K0, None, None
                else:
                    Xsrc, ldx = Ys[l - 1], Ys[l - 1].shape[1]
                    pa, pb = stats[l - 1][0], stats[l - 1][1]
                if lazy is not None and fusable(l):
                    chunks = _lib.btr_sa_bwd_fused_chunks(rows, Np, K)
                    pw = _f32((chunks, Np, K), dev)
                    dW = _f32((Np, K), dev)
                    Wt = W2.t().contiguous()
                    Gn = _f32((rows, K), dev)
                    part = _f32((chunks, 2, K), dev)
                    m1, m2, dg, db = (_f32((K,), dev) for _ in range(4))
                    scl, shl, mul, isl = stats[l]
                    scp, shp, mup, isp = stats[l - 1]
                    _call(_lib.btr_sa_bwd_fused, rows, Np, K, _p(dY), Np, _p(Ys[l]), _p(scl),
                          _p(shl), _p(mul), _p(isl), _p(lazy[0]), _p(lazy[1]), 0, None, None, None,
                          None, _p(Xsrc), ldx, None, _p(pa), _p(pb), _p(mup), _p(isp), _p(Wt), Np,
                          _p(Gn), K, _p(pw), _p(dW), _p(part), _p(m1), _p(m2), _p(dg), _p(db), st,
                          key=(rows, Np, K))
                    grads[4 * l] = dW[:Nl].reshape(ctx.pshapes[4 * l])
                    wprev = ctx.widths[l - 1]
                    grads[4 * (l - 1) + 2], grads[4 * (l - 1) + 3] = dg[:wprev], db[:wprev]
                    if ctx.pshapes[4 * (l - 1) + 1] is not None:
                        grads[4 * (l - 1) + 1] = torch.zeros(wprev, device=dev)
                    dY = Gn
                    lazy = (m1, m2)
                    continue
                if lazy is not None:   # a consumer that wants dY_l itself
                    scl, shl, mul, isl = stats[l]
                    _call(_lib.btr_sa_bn_relu_bwd_apply, rows, Np, Np, _p(dY), _p(Ys[l]), _p(scl),
                          _p(shl), _p(mul), _p(isl), _p(lazy[0]), _p(lazy[1]), st)
                    lazy = None
                chunks = _lib.btr_sa_gemm_tn_chunks(rows, Np, K)
                pw = _f32((chunks, Np, K), dev)
                dW = _f32((Np, K), dev)
                _call(_lib.btr_sa_gemm_tn, rows, Np, K, _p(dY), Np, _p(Xsrc), ldx, _p(pa), _p(pb),
                      _p(pw), _p(dW), st, key=(rows, Np, K))
                grads[4 * l] = dW[:Nl].reshape(ctx.pshapes[4 * l])
                if l > 0 or need_x:
                    Wt = W2.t().contiguous()
                    Gn = _f32((rows, K), dev)
                    _gemm_nt(rows, K, Np, dY, Np, Wt, Gn, None, None, None, None, st)
                    if l > 0:
                        sc, sh, mu, isd = stats[l - 1]
                        part = _f32((1024, 2, K), dev)
                        m1, m2, dg, db = (_f32((K,), dev) for _ in range(4))
                        if fusable(l - 1):   # the next layer applies the sums itself
                            _call(_lib.btr_sa_bn_relu_bwd_sums, rows, K, K, _p(Gn), _p(Ys[l - 1]),
                                  _p(sc), _p(sh), _p(mu), _p(isd), _p(part), _p(m1), _p(m2),
                                  _p(dg), _p(db), st)
                            lazy = (m1, m2)
                        else:
                            _call(_lib.btr_sa_bn_relu_bwd, rows, K, K, _p(Gn), _p(Ys[l - 1]),
                                  _p(sc), _p(sh), _p(mu), _p(isd), _p(part), _p(m1), _p(m2),
                                  _p(dg), _p(db), st)
                        wprev = ctx.widths[l - 1]
                        grads[4 * (l - 1) + 2], grads[4 * (l - 1) + 3] = dg[:wprev], db[:wprev]
                        if ctx.pshapes[4 * (l - 1) + 1] is not None:
                            grads[4 * (l - 1) + 1] = torch.zeros(wprev, device=dev)
                        dY = Gn
                    else:
                        dx = _f32((B, K0, N), dev)
                        _call(_lib.btr_pm_out, B, N, K0, K0, _p(Gn), None, None, 0, _p(dx), None,
                              st)
        return (dx, None) + tuple(grads)


def native_enabled():
    """BTR_NATIVE_LAYERS=0 (or inside bench.py's instrumented steps): the Python sequence
    (PointwiseMLP) instead of one btr_pm_chain_forward / _backward call per chain."""
    return os.environ.get("BTR_NATIVE_LAYERS", "1") != "0" and not _ext.timing_detail()


_CHAIN_CACHE = weakref.WeakKeyDictionary()   # first conv of a chain -> {shape: (desc, plan, ..)}


def _u8(nbytes, dev):
    return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=dev)


def chain_static(d, B, K0, N, layers, params, need_dx):
    """The shape part of a btr_pm_chain_t: layers = [{"bn": BatchNorm | None, ...}], params =
    (W, bias, gamma, beta) per layer."""
    d.b, d.n, d.c, d.layers, d.need_dx = B, N, K0, len(layers), int(need_dx)
    for l, spec in enumerate(layers):
        d.width[l] = params[4 * l].shape[0]
        d.has_bn[l] = 1 if spec["bn"] is not None else 0
        d.eps[l] = float(spec["bn"].eps) if spec["bn"] is not None else 0.0


def chain_pointers(d, layers, params):
    """The per-call part: parameter / running-statistics pointers and the BatchNorm momentum."""
    for l, spec in enumerate(layers):
        W, bias, gamma, beta = params[4 * l:4 * l + 4]
        bn = spec["bn"]
        assert W.is_contiguous()
        d.w[l] = W.data_ptr()
        d.bias[l] = _p(bias)
        d.gamma[l], d.beta[l] = _p(gamma), _p(beta)
        track = bn is not None and bn.track_running_stats and bn.running_mean is not None
        d.running_mean[l] = bn.running_mean.data_ptr() if track else None
        d.running_var[l] = bn.running_var.data_ptr() if track else None
        d.num_batches_tracked[l] = bn.num_batches_tracked.data_ptr() if track else None
        if bn is not None:
            d.momentum[l] = float(bn.momentum) if bn.momentum is not None else \
                1.0 / float(bn.num_batches_tracked.item() + 1)


def chain_sizes(plan, L):
    """Block sizes of a chain's flat gradient buffer (btr_pm_plan_t dw / dgamma / dbeta, dbias)."""
    sizes = []
    for l in range(L):
        sizes += [plan.np[l] * plan.kin[l], plan.np[l], plan.np[l]]
    return sizes + [plan.np[l] for l in range(L)]


def chain_grad_views(d, plan, sizes, pshapes, grads):
    """[dW, dbias, dgamma, dbeta] per layer as views of the chain's flat gradient buffer."""
    L = d.layers
    parts = grads.split(sizes)
    res = []
    for l in range(L):
        Nl = d.width[l]
        wshape, bshape = pshapes[4 * l], pshapes[4 * l + 1]
        dW = parts[3 * l].view(plan.np[l], plan.kin[l])[:Nl]
        if plan.kin[l] != wshape[1]:    # first layer of an input padded to 4 columns
            dW = dW[:, :wshape[1]]
        dW = dW.reshape(wshape)
        dbias = parts[3 * L + l][:Nl] if bshape is not None else None
        if d.has_bn[l]:
            res += [dW, dbias, parts[3 * l + 1][:Nl], parts[3 * l + 2][:Nl]]
        else:
            res += [dW, dbias, None, None]
    return res


class PointwiseChain(Function):
    """PointwiseMLP with the launch sequence in C++ (btr_pm_chain_forward / _backward,
    csrc/sa_layer.hip).  Same kernels; the bias gradient of a bare last layer is summed by
    a kernel of the library instead of torch.sum (rounding-level difference)."""

    @staticmethod
    def forward(ctx, x, meta, sink, *params):
        _ext.RUNNING_STATS_EPOCH[0] += 1   # running statistics move through raw pointers
        layers = meta["layers"]
        L = len(layers)
        dev = x.device
        B, K0, N = x.shape
        rows = B * N
        need_dx = bool(ctx.needs_input_grad[0])
        ent = chain_entry(meta, x, params, need_dx)
        ctx.to_sink = sink is not None
        d, plan, sizes = ent[:3]
        chain_pointers(d, layers, params)
        x_cl = _ext.twin_of(x)
        if x_cl is not None and (x_cl.shape != (rows, K0) or not x_cl.is_contiguous() or K0 % 4):
            x_cl = None
        xb = x.contiguous() if x_cl is None else None
        NL = d.width[L - 1]
        out = _f32((B, NL, N), dev)
        out_cl = _f32((rows, NL), dev)
        saved = _u8(plan.saved_bytes, dev)
        scratch = _u8(plan.fwd_scratch_bytes, dev)
        with _on(x) as dv:
            _call(_lib.btr_pm_chain_forward, ctypes.addressof(d), ctypes.addressof(plan), _p(xb),
                  _p(x_cl), _p(out), _p(out_cl), _p(saved), _p(scratch), _stream(dv))
        _ext.attach_twin(out, out_cl)
        ctx.plan = tuple(ent[:3])
        ctx.has_x_cl = x_cl is not None
        ctx.pshapes = [None if p is None else p.shape for p in params]
        if x_cl is not None:
            ctx.save_for_backward(saved, x_cl)
        else:
            ctx.save_for_backward(saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        d, plan, sizes = ctx.plan
        saved = ctx.saved_tensors[0]
        x_cl = ctx.saved_tensors[1] if ctx.has_x_cl else None
        dev = dout.device
        L = d.layers
        dout = dout.contiguous()
        grads = _f32((plan.grads_floats,), dev)
        scratch = _u8(plan.bwd_scratch_bytes, dev)
        dx = _f32((d.b, d.c, d.n), dev) if d.need_dx else None
        with _on(dout) as dv:
            _call(_lib.btr_pm_chain_backward, ctypes.addressof(d), ctypes.addressof(plan),
                  _p(x_cl), _p(dout), _p(saved), _p(grads), _p(dx), _p(scratch), _stream(dv))
        if ctx.to_sink:   # (one flat gradient for the sink, grad_sink.py)
            return (dx, None, grads) + (None,) * len(ctx.pshapes)
        return (dx, None, None) + tuple(chain_grad_views(d, plan, sizes, ctx.pshapes, grads))


def chain_entry(meta, x, params, need_dx):
    """[description, plan, gradient block sizes, sink | None] of a chain at this input shape."""
    B, K0, N = x.shape
    key = (B, K0, N, bool(need_dx))
    cache = meta["cache"]
    ent = cache.get(key)
    if ent is None:
        layers = meta["layers"]
        d = _ext.PmChain()
        chain_static(d, B, K0, N, layers, params, need_dx)
        plan = _ext.PmPlan()
        _call(_lib.btr_pm_chain_plan, ctypes.addressof(d), ctypes.addressof(plan))
        ent = cache[key] = [d, plan, chain_sizes(plan, len(layers)), None]
    return ent


def chain_sink(meta, x, params):
    """The flat gradient sink of this chain call (grad_sink.py), or None outside its scope."""
    if not grad_sink.active() or not grad_sink.all_leaves(params):
        return None
    ent = chain_entry(meta, x, params, x.requires_grad)
    if ent[3] is None or ent[3].tensor.device != x.device:
        d, plan, sizes = ent[:3]
        pshapes = [None if p is None else p.shape for p in params]
        ent[3] = grad_sink.Sink(plan.grads_floats, x.device,
                                lambda g: chain_grad_views(d, plan, sizes, pshapes, g))
    return ent[3].bind(params)


def _layer_ok(conv, bn, K, first):
    import torch.nn as nn
    if conv.kernel_size not in ((1,), (1, 1)) or conv.stride not in ((1,), (1, 1)) or \
            conv.groups != 1 or conv.in_channels != K:
        return False
    if K % 4 != 0 and not (first and native_enabled()):   # the C sequence pads the input rows
        return False
    if bn is not None:
        if not isinstance(bn, (nn.BatchNorm1d, nn.BatchNorm2d)) or bn.weight is None or \
                not bn.training or conv.out_channels % 4 != 0 or conv.out_channels > 512:
            return False
    return True


def shared_mlp_chain(mlp):
    """[(conv, bn, relu)] of a pytorch_utils.SharedMLP whose layers are conv -> bn -> ReLU
    (the feature-propagation MLPs), else None."""
    import torch.nn as nn
    chain = []
    for layer in mlp:
        if [n for n, _ in layer.named_children()] != ["conv", "bn", "activation"]:
            return None
        if not isinstance(layer.activation, nn.ReLU):
            return None
        chain.append((layer.conv, layer.bn.bn, True))
    return chain or None


_CAPTURE_PATH = [0]   # > 0: take the decisions a HIP-graph capture takes (fused_backbone.layerwise)


def _min_rows():
    """Chains of fewer rows stay on the stock ops.  Below ~2 000 rows a chain is a string of
    5-15 us launches either way: issued as one library call it costs the host less than the
    stock ops (GroupFree3D's 256-query heads and position embeddings, eager step: 20.5 ->
    16.2 ms with them on this path); as nodes of a replayed HIP graph the stock ones are
    fewer (15.5 vs 16.2 ms per replay)."""
    v = os.environ.get("BTR_CHAIN_MIN_ROWS")
    if v is not None:
        return int(v)
    if native_enabled() and not _CAPTURE_PATH[0] and not torch.cuda.is_current_stream_capturing():
        return 0
    return 2048


PATHS = {"library": 0, "library_eval": 0, "python": 0, "stock": 0, "stock_small": 0}   # run_chain decisions (bench.py reports them)


def run_chain(x, chain):
    """chain: [(conv, bn | None, relu: bool)].  Returns None when the fused path does not
    cover the configuration (CPU tensors, eval mode, disabled): the caller then runs the
    stock ops."""
    out = _run_chain(x, chain)
    if out is None:
        PATHS["stock"] += 1
    return out


def chain_spec(K, chain):
    """([{"bn", "relu"}], [W, bias, gamma, beta per layer]) of a chain on K input channels that
    the library's chain entry points cover, else None."""
    metas, params = [], []
    for i, (conv, bn, relu) in enumerate(chain):
        if not _layer_ok(conv, bn, K, i == 0):
            return None
        if bn is None and (relu or i != len(chain) - 1):
            return None
        if bn is not None and not relu:
            return None
        metas.append({"bn": bn, "relu": relu})
        params += [conv.weight, conv.bias, bn.weight if bn is not None else None,
                   bn.bias if bn is not None else None]
        K = conv.out_channels
    if K > 512 and chain[-1][1] is not None:
        return None
    return metas, params


def _eval_ok(K, chain):
    """The chain in inference mode (running statistics, no graph recorded) is covered."""
    import torch.nn as nn
    for i, (conv, bn, relu) in enumerate(chain):
        if conv.kernel_size not in ((1,), (1, 1)) or conv.stride not in ((1,), (1, 1)) or \
                conv.groups != 1 or conv.in_channels != K:
            return False
        if bn is None:
            if relu or i != len(chain) - 1:
                return False
        elif not relu or not isinstance(bn, (nn.BatchNorm1d, nn.BatchNorm2d)) or bn.training or \
                bn.weight is None or bn.running_mean is None or conv.out_channels % 4 != 0 or \
                conv.out_channels > 512:
            return False
        K = conv.out_channels
    return True


def _eval_chain_constants(chain, k0p):
    """Per layer [W (ceil4(width), k_in) zero-padded, a, b, bias, weight planes | None] of a
    chain in inference mode: BatchNorm on its running statistics after a convolution with bias
    is  a * (W x) + b  with  a = gamma / sqrt(running_var + eps),
    b = beta + (bias - running_mean) * a;  a layer without BatchNorm keeps its bias.  Cached
    per chain under the tensors' version counters and _ext.RUNNING_STATS_EPOCH (the library's
    writes through raw pointers), as fused_sa._eval_constants."""
    tensors = []
    for conv, bn, _ in chain:
        tensors += [conv.weight, conv.bias]
        if bn is not None:
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    key = (k0p, _ext.RUNNING_STATS_EPOCH[0]) + tuple(
        None if t is None else (t.data_ptr(), t._version) for t in tensors)
    cache = _CHAIN_CACHE.get(chain[0][0])
    if cache is None:
        cache = _CHAIN_CACHE[chain[0][0]] = {}
    hit = cache.get("eval")
    if hit is not None and hit[0] == key:
        return hit[1]
    res, K = [], k0p
    with torch.no_grad():
        for conv, bn, _ in chain:
            W, bias = conv.weight, conv.bias
            Nl = W.shape[0]
            Np = _ceil4(Nl)
            W2 = W.reshape(Nl, -1)
            if tuple(W2.shape) != (Np, K):
                Wp = torch.zeros((Np, K), dtype=torch.float32, device=W.device)
                Wp[:Nl, :W2.shape[1]] = W2
                W2 = Wp
            a = b = bp = None
            if bn is not None:
                a = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
                shift = bn.running_mean if bias is None else bn.running_mean - bias
                b = (bn.bias - shift * a).contiguous()
            elif bias is not None:
                bp = bias if Np == Nl else torch.cat(
                    [bias, torch.zeros(Np - Nl, dtype=torch.float32, device=W.device)])
                bp = bp.contiguous()
            res.append([W2.contiguous(), a, b, bp, None])
            K = Np
    cache["eval"] = (key, res)
    return res


def _eval_chain(x, chain):
    """Inference-mode forward of a chain (module.eval() under no_grad, the evaluation pass of
    the reference, train_Votenet_FSB.py:246-293): one GEMM launch per layer with the previous
    layer's BatchNorm (running statistics) + ReLU applied while its operand is staged, between
    one layout launch on either side -- instead of conv, batch_norm and relu per layer on the
    stock ops."""
    dev = x.device
    B, K0, N = x.shape
    rows = B * N
    K0p = _ceil4(K0)
    consts = _eval_chain_constants(chain, K0p)
    with _on(x) as d:
        st = _stream(d)
        A = _ext.twin_of(x) if K0p == K0 else None
        if A is not None and A.is_contiguous() and A.numel() == rows * K0 and A.shape[-1] == K0:
            A = A.view(rows, K0)
        else:
            A = _f32((rows, K0p), dev)
            _call(_lib.btr_pm_rows, B, N, K0, K0p, _p(x.contiguous()), _p(A), st)
        lda, K = K0p, K0p
        pa = pb = None
        for c in consts:
            W2, a, b, bias = c[:4]
            Np = W2.shape[0]
            Y = _f32((rows, Np), dev)
            if _lib.btr_pm_gemm_nt_sm_supported(rows, Np, K):
                if c[4] is None:
                    c[4] = torch.empty((int(_lib.btr_pm_weight_planes_bytes(Np, K)),),
                                       dtype=torch.uint8, device=dev)
                    _call(_lib.btr_pm_weight_planes, Np, K, _p(W2), K, _p(c[4]), st)
                _call(_lib.btr_pm_gemm_nt_sm, rows, Np, K, _p(A), lda, _p(c[4]), _p(Y), Np, _p(pa),
                      _p(pb), None, _p(bias), st, key=(rows, Np, K))
            else:
                _call(_lib.btr_pm_gemm_nt, rows, Np, K, _p(A), lda, _p(W2), K, _p(Y), Np, _p(pa),
                      _p(pb), None, _p(bias), st, key=(rows, Np, K))
            pa, pb = a, b
            A, lda, K = Y, Np, Np
        NL = chain[-1][0].out_channels
        out = _f32((B, NL, N), dev)
        out_cl = _f32((rows, NL), dev)
        _call(_lib.btr_pm_out, B, N, NL, lda, _p(A), _p(pa), _p(pb), 1 if pa is not None else 0,
              _p(out), _p(out_cl), st)
    _ext.attach_twin(out, out_cl)
    return out


def _run_chain(x, chain):
    if not (enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3):
        return None
    if not torch.is_grad_enabled() and _eval_ok(x.shape[1], chain):
        PATHS["library_eval"] += 1
        return _eval_chain(x, chain)
    if x.shape[0] * x.shape[2] < _min_rows():
        PATHS["stock_small"] += 1
        return None
    spec = chain_spec(x.shape[1], chain)
    if spec is None:
        return None
    metas, params = spec
    if native_enabled() and len(chain) <= _ext.MAX_LAYERS:
        cache = _CHAIN_CACHE.get(chain[0][0])
        if cache is None:
            cache = _CHAIN_CACHE[chain[0][0]] = {}
        PATHS["library"] += 1
        meta = {"layers": metas, "cache": cache}
        return PointwiseChain.apply(x, meta, chain_sink(meta, x, params), *params)
    PATHS["python"] += 1
    return PointwiseMLP.apply(x, {"layers": metas}, *params)
