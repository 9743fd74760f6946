"""The decoder loop of GroupFreeDetector.forward as ONE autograd node (csrc/gf_stack.hip).

The reference (detection/GroupFree3D/models/detector.py:161-219) runs, per decoder layer, two
position embeddings (modules.py:50-65), the layer (transformer.py:36-76), the prediction head
(modules.py:107-262) and the detached query-position bookkeeping (detector.py:204-230).  Each of
those is already one library call per direction here (fused_mlp.PointwiseChain,
fused_decoder.DecoderLayerFn, fused_decode.HeadDecode); the training step was nevertheless paced
by the host, because every call sits in its own autograd node: ~0.1 ms of interpreter, allocator
and ctypes work each way around ten launches of 5 - 20 us.  `run(...)` issues the same launches
from two calls, btr_gf_stack_forward / btr_gf_stack_backward, on buffers cut out of one saved /
scratch / gradient allocation: 30 nodes become one.  The results are those of the per-module path
bit for bit (same kernels, same operand order in the gradient sums; tests/test_gf_stack_gpu.py).

Covers what the per-module library paths cover (CUDA f32 tensors, training mode, ReLU layers,
learned position embeddings or none); `BTR_FUSED_GF_STACK=0` disables it, and a HIP-graph capture
keeps the per-module path (see fused_mlp._min_rows).

Round 6: the two calls replay as HIP graphs (csrc/graph_cache.hip: ~700 launches of 5 - 20 us cost
the host 15 us per call instead of ~3 us per launch, and the queue ~1.1 us between two dependent
kernels instead of 2.5 - 3.5).  A graph replays exact pointers, and the caching allocator does not
hand the same addresses to consecutive steps (tools/diag_gf_ptrs.py: 16 distinct argument tuples
in 16 steps), so a call works on a SLOT of buffers that live as long as the detector: inputs are
copied / laid out into it, the library writes its outputs there, and what autograd and the caller
see is ONE clone of the slot's output block -- nothing handed out aliases a slot.  A slot is busy
from its forward to the end of its backward (two forwards before a backward -- the two-branch
steps -- take two slots); `BTR_GF_SLOTS=0` keeps the per-call allocations."""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function

from ..pointnet2 import _ext, fused_mlp
from . import fused_attention, fused_decode, fused_decoder

_call, _lib, _on, _p, _stream = _ext._call, _ext._lib, _ext._on, _ext._p, _ext._stream
_ENTRIES = weakref.WeakKeyDictionary()   # detector -> {shape key: entry}
_VP = ctypes.c_void_p
CALLS = [0]   # forward calls that took this path (bench.py / tests read it)
REFUSED = {}  # reason -> calls that stayed on the module loop


def _no(why):
    REFUSED[why] = REFUSED.get(why, 0) + 1
    return False


def enabled():
    return os.environ.get("BTR_FUSED_GF_STACK", "1") != "0"


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def _u8(nbytes, dev):
    return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=dev)


def _slots_enabled():
    return os.environ.get("BTR_GF_SLOTS", "1") != "0"


_MAX_SLOTS = 4


def _up64(n):
    return (int(n) + 63) // 64 * 64


class _OutBlock(object):
    """The stack's outputs as views of one flat f32 buffer (pieces start on 256-byte bounds)."""

    def __init__(self, flat, L, B, Pq, E, C, Cp, nh, ns, want_last):
        at = [0]

        def take(*shape):
            n = 1
            for v in shape:
                n *= v
            t = flat[at[0]:at[0] + n].view(shape)
            at[0] += _up64(n)
            return t
        self.cls = take(L, B * Pq, Cp)
        self.qpos, self.qpos_t = take(L, B, Pq, 6), take(L, B, 6, Pq)
        self.outs = []
        for i in range(L):
            self.outs.append((take(B, C, Pq), take(B, Pq, 3), take(B, Pq, nh), take(B, Pq, ns, 3),
                              take(B, Pq, 3), self.qpos[i], self.qpos_t[i]))
        self.last = take(B, E, Pq) if want_last else None
        self.last_cl = take(B * Pq, E) if want_last else None

    @staticmethod
    def floats(L, B, Pq, E, C, Cp, nh, ns, want_last):
        n = _up64(L * B * Pq * Cp) + 2 * _up64(L * B * Pq * 6)
        n += L * (_up64(B * C * Pq) + 2 * _up64(B * Pq * 3) + _up64(B * Pq * nh) +
                  _up64(B * Pq * ns * 3))
        if want_last:
            n += 2 * _up64(B * E * Pq)
        return n


class _Slot(object):
    """The buffers of one decoder-stack call in flight (see the module docstring)."""

    def __init__(self, ent, dev, want_last, qd, kd, cat_shapes):
        d, plan = ent.d, ent.plan
        L, B, Pq, Pk, E = d.layers, d.b, d.pq, d.pk, d.e
        C, nh, ns = d.head_c, d.nh, d.ns
        Cp = plan.head[0].np[d.head[0].layers - 1]
        self.busy = False
        self.dims = (L, B, Pq, E, C, Cp, nh, ns, want_last)
        self.x_cl, self.key_cl = _f32((B * Pq, E), dev), _f32((B * Pk, E), dev)
        self.qpos0_t = _f32((B, qd, Pq), dev) if qd else None
        self.key_xyz_t = _f32((B, kd, Pk), dev) if kd else None
        self.base = _f32((B, Pq, 3), dev)
        self.cat = [(_f32(ws, dev), _f32(bs, dev)) for ws, bs in cat_shapes]
        self.cat_views = None   # the seven layers' places in them (made on first use)
        self.out_flat = _f32((_OutBlock.floats(*self.dims),), dev)
        self.out = _OutBlock(self.out_flat, *self.dims)
        self.saved = _u8(plan.saved_bytes, dev)
        self.scratch = _u8(max(plan.fwd_scratch_bytes, plan.bwd_scratch_bytes), dev)
        self.dheads = _f32((L, B, C, Pq), dev)
        self.dlast = _f32((B, E, Pq), dev) if want_last else None
        self.grads = _f32((plan.grads_floats,), dev)
        self.dquery, self.dkey = _f32((B, E, Pq), dev), _f32((B, E, Pk), dev)
        self.seeds = [fused_attention._next_seed() for _ in range(L)]
        self.d = None          # this slot's descriptor (seeds, the heads' concatenated layers)
        self.ptrs = None       # ent.ptrs the descriptor was made from
        self.fwd_arrays = None


class _Lease(object):
    """Frees its slot when the autograd node that holds it goes away (or at the end of the
    backward, whichever is first)."""
    __slots__ = ("slot",)

    def __init__(self, slot):
        self.slot = slot
        slot.busy = True

    def release(self):
        if self.slot is not None:
            self.slot.busy = False
            self.slot = None

    def __del__(self):
        self.release()


def _acquire(ent, dev, stream, want_last, qd, kd, cat_shapes):
    """A free slot of `ent` for calls on `stream`, or None (slots off, or _MAX_SLOTS in flight:
    the call then allocates per call and is issued launch by launch)."""
    if not _slots_enabled():
        return None
    pool = ent.slots.setdefault((dev, stream, want_last), [])
    for s in pool:
        if not s.busy:
            return s
    if len(pool) >= _MAX_SLOTS:
        return None
    pool.append(_Slot(ent, dev, want_last, qd, kd, cat_shapes))
    return pool[-1]


def _transposed(xyz):
    """(B, d, P) contiguous form of (B, P, d) coordinates (cached on the tensor by the head decode
    kernel / the position embedding module)."""
    cached = getattr(xyz, '_btr_t', None)
    if cached is not None and cached[1] == xyz._version:
        return cached[0]
    x = xyz.transpose(1, 2).contiguous()
    if not xyz.requires_grad:
        xyz._btr_t = (x, xyz._version)
    return x


def _embed_chain(m):
    from .modules import PositionEmbeddingLearned
    if not isinstance(m, PositionEmbeddingLearned):
        return None
    head = m.position_embedding_head
    return [(head[0], head[1], True), (head[3], None, False)]


class _Entry(object):
    """What is fixed for a detector and a set of shapes: the descriptor with its sizes filled in,
    the plan, the block sizes of the flat gradient buffer."""
    __slots__ = ("d", "plan", "layer_sizes", "chain_sizes", "ptrs", "metas", "sink", "slots")


def _entry(det, key, B, Pq, Pk, E, qd, kd, p, specs):
    cache = _ENTRIES.get(det)
    if cache is None:
        cache = _ENTRIES[det] = {}
    ent = cache.get(key)
    if ent is not None:
        return ent
    L = det.num_decoder_layers
    d = _ext.GfStack()
    head0 = det.prediction_heads[0]
    d.layers, d.b, d.pq, d.pk, d.e = L, B, Pq, Pk, E
    d.nh, d.ns = head0.num_heading_bin, head0.num_size_cluster
    d.has_qpos, d.has_kpos = int(qd > 0), int(kd > 0)
    for i in range(L):
        layer = det.decoder[i]
        ld = d.layer[i]
        ld.b, ld.pq, ld.pk, ld.e = B, Pq, Pk, E
        ld.heads, ld.ff, ld.dropout = layer.self_attn.num_heads, layer.linear1.out_features, p
        for j, ln in enumerate((layer.norm1, layer.norm2, layer.norm3)):
            ld.ln_eps[j] = float(ln.eps)
        (qm, qp), (km, kp), (hm, hp) = specs[i]
        if qm is not None:
            fused_mlp.chain_static(d.qpos[i], B, qd, Pq, qm, qp, False)
        if km is not None:
            fused_mlp.chain_static(d.kpos[i], B, kd, Pk, km, kp, False)
        fused_mlp.chain_static(d.head[i], B, E, Pq, hm, hp, True)
    d.head_c = d.head[0].width[d.head[0].layers - 1]
    plan = _ext.GfStackPlan()
    assert _lib.btr_gf_stack_sizeof(0) == ctypes.sizeof(d) and \
        _lib.btr_gf_stack_sizeof(1) == ctypes.sizeof(plan), "btr_gf_stack_t mirror out of date"
    _call(_lib.btr_gf_stack_plan, ctypes.addressof(d), ctypes.addressof(plan))
    ent = _Entry()
    ent.d, ent.plan, ent.ptrs, ent.sink, ent.slots = d, plan, None, None, {}
    F = d.layer[0].ff
    ent.layer_sizes = [3 * E * E, 3 * E, E * E, E, 3 * E * E, 3 * E, E * E, E, F * E, F, E * F, E,
                       E, E, E, E, E, E]
    assert sum(ent.layer_sizes) == plan.layer[0].grads_floats
    ent.chain_sizes = []
    for i in range(L):
        (qm, _), (km, _), (hm, _) = specs[i]
        ent.chain_sizes.append((
            fused_mlp.chain_sizes(plan.qpos[i], len(qm)) if qm is not None else None,
            fused_mlp.chain_sizes(plan.kpos[i], len(km)) if km is not None else None,
            fused_mlp.chain_sizes(plan.head[i], len(hm))))
    cache[key] = ent
    return ent


def _specs(det, E, qd, kd, standin_device=None):
    """Per layer ((metas, params) of the query / key position embedding and the head), or None
    when a chain is not covered.  standin_device: the heads' concatenated output layers appear as
    shape-only stand-ins (the slot path fills its own copy from the seven layers, `_head_subs`)."""
    specs = []
    for i in range(det.num_decoder_layers):
        layer = det.decoder[i]
        row = []
        for m, K in ((layer.self_posembed, qd), (layer.cross_posembed, kd)):
            if K == 0:
                if m is not None:
                    return None
                row.append((None, []))
                continue
            chain = _embed_chain(m)
            spec = fused_mlp.chain_spec(K, chain) if chain is not None else None
            if spec is None or chain[-1][0].out_channels != E:
                return None
            row.append(spec)
        head = det.prediction_heads[i]
        last = head.cat_standin(standin_device) if standin_device is not None else None
        spec = fused_mlp.chain_spec(E, head.chain(last))
        if spec is None:
            return None
        row.append(spec)
        specs.append(row)
    return specs


def _head_subs(det):
    """(weight, bias) of the seven output layers of every prediction head, head by head, and the
    rows each takes in the head's concatenated layer."""
    subs, rows = [], []
    for i in range(det.num_decoder_layers):
        heads = det.prediction_heads[i]._heads()
        rows.append([h.out_channels for h in heads])
        for h in heads:
            subs += [h.weight, h.bias]
    return subs, rows


def _flat_params(det, specs):
    """(parameters in descriptor order, positions of the heads' concatenated output layers)."""
    params, cat_at = [], []
    for i in range(det.num_decoder_layers):
        params += list(fused_decoder._params(det.decoder[i]))
        for _m, ps in specs[i]:
            params += ps
        cat_at.append(len(params) - 4)   # (W, bias, None, None) of the head's last layer
    return params, cat_at


def _set_pointers(ent, det, specs, params, cat_at):
    """Parameter pointers into the cached descriptor -- only when one of them moved (they are the
    optimizer's own tensors: fixed unless the model is reloaded or moved).  The heads' last layers
    are concatenations made for this call: always set."""
    ptrs = [0 if t is None else t.data_ptr() for t in params]
    for i, at in enumerate(cat_at):
        hd = ent.d.head[i]
        hd.w[hd.layers - 1], hd.bias[hd.layers - 1] = ptrs[at], ptrs[at + 1]
        ptrs[at] = ptrs[at + 1] = 0
    for i in range(det.num_decoder_layers):   # BatchNorm buffers / momentum
        for m, _ps in specs[i]:
            if m is not None:
                for spec in m:
                    bn = spec["bn"]
                    if bn is not None:
                        ptrs += [bn.running_mean.data_ptr() if bn.running_mean is not None else 0,
                                 bn.momentum]
    if ptrs == ent.ptrs:
        return
    d = ent.d
    at = 0
    for i in range(det.num_decoder_layers):
        ld = d.layer[i]
        lp = params[at:at + 18]
        at += 18
        for name, t in zip(fused_decoder._FIELDS, lp[:12]):
            assert t.is_contiguous()
            setattr(ld, name, t.data_ptr())
        for j in range(3):
            ld.ln_w[j], ld.ln_b[j] = lp[12 + 2 * j].data_ptr(), lp[13 + 2 * j].data_ptr()
        for (m, ps), cd in zip(specs[i], (d.qpos[i], d.kpos[i], d.head[i])):
            if m is not None:
                fused_mlp.chain_pointers(cd, m, params[at:at + len(ps)])
            at += len(ps)
    ent.ptrs = ptrs


def _slot_descriptor(slot, ent, p, dev):
    """The slot's own copy of the descriptor: the entry's (sizes, parameter pointers) with the
    slot's dropout seeds and the heads' last layers read from the slot's copies.  Remade when a
    parameter pointer moved (ent.ptrs is replaced then)."""
    if slot.d is not None and slot.ptrs is ent.ptrs:
        return
    d = _ext.GfStack.from_buffer_copy(ent.d)
    step = fused_attention.step_counter(dev) if p > 0 else None
    for i in range(d.layers):
        if p > 0:
            d.layer[i].seed = slot.seeds[i]
            d.layer[i].step = step.data_ptr()
        hd = d.head[i]
        hd.w[hd.layers - 1] = slot.cat[i][0].data_ptr()
        hd.bias[hd.layers - 1] = slot.cat[i][1].data_ptr()
    slot.d, slot.ptrs = d, ent.ptrs
    slot.fwd_arrays = _fwd_arrays(slot.out)


def _ptr_array(tensors):
    return (_VP * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class DecoderStackFn(Function):
    """(query (B,E,Pq), key (B,E,Pk), qpos0_t (B,d,Pq) | None, key_xyz_t (B,3,Pk) | None,
    base_xyz (B,Pq,3), mean_size, meta, *parameters) -> per layer (head output (B,C,Pq), center,
    heading_residuals, size_residuals, pred_size, query_pos, query_pos_t) [+ the last layer's
    output (B,E,Pq) when meta["want_last"]]."""

    @staticmethod
    def forward(ctx, query, key, qpos0_t, key_xyz_t, base_xyz, mean_size, meta, sink, *params):
        _ext.RUNNING_STATS_EPOCH[0] += 1   # running statistics move through raw pointers
        ent, p, want_last = meta["entry"], meta["p"], meta["want_last"]
        # (slot path: the seven output layers of every head follow the descriptor-order
        # parameters; the concatenated layers among those are shape-only stand-ins)
        nsub = meta.get("nsub", 0)
        subs = params[len(params) - nsub:] if nsub else ()
        if nsub:
            params = params[:len(params) - nsub]
        ctx.subs = (nsub, meta.get("sub_rows"), meta["cat_at"])
        # (flat gradient sink, pointnet2/grad_sink.py: the leaf parameters' gradients leave as
        # ONE buffer; computed operands -- the heads' concatenated last layers -- keep theirs)
        ctx.leaf = [t is not None and t.is_leaf and t.requires_grad for t in params] \
            if sink is not None else None
        plan = ent.plan
        slot = meta.get("slot")
        dev = query.device
        step = fused_attention.step_counter(dev) if p > 0 else None
        if slot is not None:
            ctx.lease = _Lease(slot)
            d = slot.d
        else:
            ctx.lease = None
            d = _ext.GfStack.from_buffer_copy(ent.d)   # this call's own copy (dropout seeds)
            if p > 0:
                for i in range(d.layers):
                    d.layer[i].seed = fused_attention._next_seed()
                    d.layer[i].step = step.data_ptr()
        L, B, Pq, E = d.layers, d.b, d.pq, d.e
        C, nh, ns = d.head_c, d.nh, d.ns
        Cp = plan.head[0].np[d.head[0].layers - 1]
        dims = (L, B, Pq, E, C, Cp, nh, ns, want_last)
        with _on(query) as dv:
            st = _stream(dv)
            if slot is not None:
                # inputs into the slot (a layout launch or a copy each), outputs out of it (one
                # clone): every pointer the library sees is the slot's
                x_cl, key_cl = _rows_into(query, slot.x_cl, st), _rows_into(key, slot.key_cl, st)
                if qpos0_t is not None:
                    qpos0_t = slot.qpos0_t.copy_(qpos0_t)
                if key_xyz_t is not None:
                    key_xyz_t = slot.key_xyz_t.copy_(key_xyz_t)
                base = slot.base.copy_(base_xyz)
                if slot.cat_views is None:
                    slot.cat_views = [v for (W, b), rows in zip(slot.cat, meta["sub_rows"])
                                      for wv, bv in zip(W.split(rows, 0), b.split(rows, 0))
                                      for v in (wv, bv)]
                torch._foreach_copy_(slot.cat_views, list(subs))
                ob, saved, scratch = slot.out, slot.saved, slot.scratch
                arrays = slot.fwd_arrays
            else:
                x_cl, key_cl = fused_decoder._rows(query, st), fused_decoder._rows(key, st)
                base = base_xyz.contiguous()
                ob = _OutBlock(_f32((_OutBlock.floats(*dims),), dev), *dims)
                saved = _u8(plan.saved_bytes, dev)
                scratch = _u8(plan.fwd_scratch_bytes, dev)
                arrays = _fwd_arrays(ob)
            _call(_lib.btr_gf_stack_forward, ctypes.addressof(d), ctypes.addressof(plan),
                  _p(x_cl), _p(key_cl), _p(qpos0_t), _p(key_xyz_t), _p(base), _p(mean_size),
                  *arrays, _p(ob.last), _p(ob.last_cl), _p(saved), _p(scratch), st)
        if slot is not None:   # (the backward reads the slot's own copies)
            ob = _OutBlock(slot.out_flat.clone(), *dims)
            ctx.save_for_backward(mean_size)
        else:
            ctx.save_for_backward(mean_size, saved, x_cl, key_cl, *[o[0] for o in ob.outs])
        outs = ob.outs
        for i in range(L):
            _ext.attach_twin(outs[i][0], ob.cls[i])
        if want_last:
            _ext.attach_twin(ob.last, ob.last_cl)
        ctx.ent = (d, ent, step)
        ctx.want_last = want_last
        ctx.pshapes = [None if t is None else t.shape for t in params]
        ctx.specs = meta["specs"]
        flat = [t for o in outs for t in o]
        ctx.mark_non_differentiable(*[t for o in outs for t in o[5:]])
        ctx.set_materialize_grads(False)
        return tuple(flat) + ((ob.last,) if want_last else ())

    @staticmethod
    def backward(ctx, *g):
        d, ent, _step = ctx.ent
        plan = ent.plan
        L, B, Pq, Pk, E = d.layers, d.b, d.pq, d.pk, d.e
        mean_size = ctx.saved_tensors[0]
        lease = ctx.lease
        slot = lease.slot if lease is not None else None
        if lease is not None and slot is None:
            raise RuntimeError("decoder stack: this call's buffers were released by an earlier "
                               "backward (retain_graph is not supported with BTR_GF_SLOTS=1)")
        head_outs = [o[0] for o in slot.out.outs] if slot is not None else ctx.saved_tensors[4:]
        dev = mean_size.device
        dims = (B, d.head_c, Pq, d.nh, d.ns)
        dheads, g_base = [], None
        for i in range(L):
            gh, gc, ghr, gsr, gps = g[7 * i:7 * i + 5]
            if gc is not None or ghr is not None or gsr is not None or gps is not None:
                fold = fused_decode.fold_grads(head_outs[i], mean_size, dims, gc, ghr, gsr, gps)
                gh = fold if gh is None else gh + fold
                if gc is not None and ctx.needs_input_grad[4]:
                    g_base = gc if g_base is None else g_base + gc
            dheads.append(gh)
        dlast = g[7 * L] if ctx.want_last else None
        need_q, need_k = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if slot is not None:
            have = [i for i in range(L) if dheads[i] is not None]
            if have:
                torch._foreach_copy_([slot.dheads[i] for i in have], [dheads[i] for i in have])
            dheads = [slot.dheads[i] if dheads[i] is not None else None for i in range(L)]
            if dlast is not None:
                dlast = slot.dlast.copy_(dlast)
            saved, x_cl, key_cl = slot.saved, slot.x_cl, slot.key_cl
            grads, scratch = slot.grads, slot.scratch
            dquery = slot.dquery if need_q else None
            dkey = slot.dkey if need_k else None
        else:
            dheads = [None if t is None else t.contiguous() for t in dheads]
            dlast = None if dlast is None else dlast.contiguous()
            saved, x_cl, key_cl = ctx.saved_tensors[1:4]
            grads = _f32((plan.grads_floats,), dev)
            scratch = _u8(plan.bwd_scratch_bytes, dev)
            dquery = _f32((B, E, Pq), dev) if need_q else None
            dkey = _f32((B, E, Pk), dev) if need_k else None
        with _on(saved) as dv:
            _call(_lib.btr_gf_stack_backward, ctypes.addressof(d), ctypes.addressof(plan),
                  _p(x_cl), _p(key_cl), _ptr_array(dheads), _p(dlast), _p(saved), _p(grads),
                  _p(dquery), _p(dkey), _p(scratch), _stream(dv))
        if slot is not None:
            # (autograd may keep what it is handed -- a parameter's .grad -- beyond this slot's
            # next call: the gradient block leaves as a copy)
            grads = grads.clone()
            lease.release()
        res = _grad_views(d, ent, ctx.specs, ctx.pshapes, grads)
        nsub, sub_rows, cat_at = ctx.subs
        gsubs = []
        if nsub:   # the concatenated layers' gradients, cut into the seven layers' own
            for rows, at in zip(sub_rows, cat_at):
                for wv, bv in zip(res[at].split(rows, 0), res[at + 1].split(rows, 0)):
                    gsubs += [wv, bv]
                res[at] = res[at + 1] = None
        if ctx.leaf is not None:
            res = [None if leaf else g for g, leaf in zip(res, ctx.leaf)]
            return (dquery, dkey, None, None, g_base, None, None, grads) + tuple(res) + \
                tuple(gsubs)
        return (dquery, dkey, None, None, g_base, None, None, None) + tuple(res) + tuple(gsubs)


def _fwd_arrays(ob):
    """The eight per-layer pointer arrays of btr_gf_stack_forward for an output block."""
    arrays = [_ptr_array([o[j] for o in ob.outs]) for j in range(7)]
    arrays.insert(1, _ptr_array([ob.cls[i] for i in range(len(ob.outs))]))
    return arrays


def _rows_into(t, dst, st):
    """fused_decoder._rows into a given buffer: a copy of the twin the producer kept, else the
    layout launch."""
    B, C, P = t.shape
    cl = _ext.twin_of(t)
    if cl is not None and cl.numel() == dst.numel() and cl.shape[-1] == C and cl.is_contiguous():
        return dst.copy_(cl.view(B * P, C))
    _call(_lib.btr_pm_rows, B, P, C, C, _p(t.contiguous()), _p(dst), st)
    return dst


def _grad_views(d, ent, specs, pshapes, grads):
    """The parameters' gradients as views of the stack's flat gradient buffer (descriptor order)."""
    plan = ent.plan
    res, at = [], 0
    for i in range(d.layers):
        lg = grads.narrow(0, plan.g_layer[i], plan.layer[i].grads_floats)
        res += [t.view(s) for t, s in zip(lg.split(ent.layer_sizes), pshapes[at:at + 18])]
        at += 18
        for (m, ps), cd, cp, off, sizes in zip(
                specs[i], (d.qpos[i], d.kpos[i], d.head[i]),
                (plan.qpos[i], plan.kpos[i], plan.head[i]),
                (plan.g_qpos[i], plan.g_kpos[i], plan.g_head[i]), ent.chain_sizes[i]):
            if m is not None:
                cg = grads.narrow(0, off, cp.grads_floats)
                res += fused_mlp.chain_grad_views(cd, cp, sizes, pshapes[at:at + len(ps)], cg)
            at += len(ps)
    return res


def _stack_sink(ent, specs, params, device):
    """The flat gradient sink of the decoder stack (one per detector and shape), or None outside
    a grad_sink scope."""
    from ..pointnet2 import grad_sink
    if not grad_sink.active():
        return None
    if ent.sink is None or ent.sink.tensor.device != device:
        d = ent.d
        pshapes = [None if t is None else t.shape for t in params]
        ent.sink = grad_sink.Sink(ent.plan.grads_floats, device,
                                  lambda g: _grad_views(d, ent, specs, pshapes, g))
    return ent.sink.bind(params)


def run(det, query, key, query_pos, key_pos, base_xyz, end_points):
    """The decoder loop of `det` on query (B,E,Pq) / key (B,E,Pk) features, the first query
    position (B,Pq,3|6) | None and the key position (B,Pk,3) | None.  Fills end_points like the
    module loop does and returns True, or returns False when this path does not cover the call."""
    L = det.num_decoder_layers
    if not (enabled() and fused_mlp.enabled() and fused_mlp.native_enabled() and
            fused_decoder.enabled() and fused_decode.enabled() and
            0 < L <= _ext.GF_MAX_DECODER_LAYERS and _hook_last_only(det) and
            query.is_cuda and query.dtype == torch.float32 and key.dtype == torch.float32 and
            query.dim() == 3 and key.dim() == 3 and base_xyz.dtype == torch.float32):
        return _no("disabled, not CUDA f32, or an override of the layer hook")
    if fused_mlp._CAPTURE_PATH[0] or torch.cuda.is_current_stream_capturing() or \
            os.environ.get("BTR_CHAIN_MIN_ROWS") is not None:
        return _no("HIP-graph capture path or BTR_CHAIN_MIN_ROWS")
    B, E, Pq = query.shape
    Pk = key.shape[2]
    if tuple(base_xyz.shape) != (B, Pq, 3):
        return _no("base_xyz shape")
    qd = 0 if query_pos is None else int(query_pos.shape[-1])
    kd = 0 if key_pos is None else int(key_pos.shape[-1])
    if qd not in (0, 6) or kd not in (0, 3):   # the head decode kernel writes (center, size)
        return _no("query position must be (center, size) or none")
    for pos, P in ((query_pos, Pq), (key_pos, Pk)):
        if pos is not None and not (pos.dtype == torch.float32 and pos.dim() == 3 and
                                    pos.shape[0] == B and pos.shape[1] == P and
                                    not pos.requires_grad):
            return _no("position tensors")
    # Slot path (replayed graphs need every pointer to repeat): the heads' concatenated output
    # layers are shape-only stand-ins here, the slot fills its own copy from the seven layers
    slotted = _slots_enabled() and not torch.cuda.is_current_stream_capturing()
    # What follows only depends on the shapes and on the modules' configuration: it is worked out
    # once per (shapes, training / dropout state of every module) and kept with the detector
    # (~0.9 ms of interpreter per call otherwise, on a step some hosts barely keep ahead of)
    sig = (B, E, Pq, Pk, qd, kd, slotted, _state_signature(det))
    static = det.__dict__.get('_btr_stack_static') if slotted else None
    if static is not None and static[0] == sig:
        p, specs, want_last, key_, params, cat_at = static[1]
    else:
        ps = set()
        for i in range(L):
            layer = det.decoder[i]
            if not fused_decoder.covered(layer, query, key, None, None):
                return _no("a decoder layer is not covered")
            ps.add(float(layer.dropout.p) if layer.training else 0.0)
        if len(ps) != 1:
            return _no("dropout rates differ")
        p = ps.pop()
        head0 = det.prediction_heads[0]
        for i in range(L):
            h = det.prediction_heads[i]
            if (h.num_heading_bin, h.num_size_cluster, h.num_class) != \
                    (head0.num_heading_bin, head0.num_size_cluster, head0.num_class) or \
                    h.mean_size_arr is not head0.mean_size_arr and \
                    not (h.mean_size_arr == head0.mean_size_arr).all():
                return _no("prediction heads differ")
        specs = _specs(det, E, qd, kd, query.device if slotted else None)
        if specs is None:
            return _no("a chain is not covered")
        want_last = type(det)._after_decoder_layer is not _base_hook(det)
        key_ = (B, Pq, Pk, E, qd, kd, p,
                tuple(tuple(len(m) if m is not None else 0 for m, _ in row) for row in specs))
        params, cat_at = _flat_params(det, specs)
        if slotted:
            det.__dict__['_btr_stack_static'] = (sig, (p, specs, want_last, key_, params, cat_at))
    head0 = det.prediction_heads[0]
    # momentum=None (cumulative average) changes with num_batches_tracked every call, and the
    # cached descriptor is only refreshed when a pointer moved: the module loop handles it
    for row in specs:
        for m, _ps in row:
            for spec in (m or ()):
                if spec["bn"] is not None and spec["bn"].momentum is None:
                    return _no("BatchNorm with momentum=None")
    ent = _entry(det, key_, B, Pq, Pk, E, qd, kd, p, specs)
    slot = None
    if slotted:
        slot = _acquire(ent, query.device, torch.cuda.current_stream(query.device).cuda_stream,
                        want_last, qd, kd,
                        [(params[at].shape, params[at + 1].shape) for at in cat_at])
        if slot is None:   # every slot in flight: this call concatenates as the module loop does
            specs = _specs(det, E, qd, kd)
            params, cat_at = _flat_params(det, specs)
    _set_pointers(ent, det, specs, params, cat_at)
    mean_size = head0._mean_size_on(query.device)
    meta = {"entry": ent, "p": p, "want_last": want_last, "specs": specs, "cat_at": cat_at}
    subs = []
    if slot is not None:
        _slot_descriptor(slot, ent, p, query.device)
        meta["slot"] = slot
        subs, meta["sub_rows"] = _head_subs(det)
        meta["nsub"] = len(subs)
        # (the slot's copies are made inside the node: transposed views, no cached tensors)
        qpos0_t = query_pos.transpose(1, 2) if query_pos is not None else None
        key_xyz_t = key_pos.transpose(1, 2) if key_pos is not None else None
    else:
        qpos0_t = _transposed(query_pos) if query_pos is not None else None
        key_xyz_t = _transposed(key_pos) if key_pos is not None else None
    outs = DecoderStackFn.apply(query, key, qpos0_t, key_xyz_t, base_xyz, mean_size, meta,
                                _stack_sink(ent, specs, params, query.device), *params, *subs)
    CALLS[0] += 1
    for i in range(L):
        prefix = 'last_' if i == L - 1 else '%dhead_' % i
        out, center, hres, sres, psize, qpos, qpos_t = outs[7 * i:7 * i + 7]
        if i == L - 1 and want_last:
            det._after_decoder_layer(prefix, outs[7 * L], end_points)
        det.prediction_heads[i].publish(out, (center, hres, sres, psize, qpos, qpos_t), base_xyz,
                                        end_points, prefix)
    return True


def _state_signature(det):
    """What the coverage checks of run() read besides shapes: every module's training flag, every
    dropout rate, every parameter object's identity (the list of the modules is made once per
    detector: a sub-module ADDED to a layer or a head later is not seen, one replaced is -- its
    parameters are other objects)."""
    mods = det.__dict__.get('_btr_stack_mods')
    n = (id(det.decoder), len(det.decoder), id(det.prediction_heads), len(det.prediction_heads))
    if mods is None or mods[0] != n:   # (walking modules() costs 0.4 ms: once)
        ms = list(det.decoder.modules()) + list(det.prediction_heads.modules())
        drops = [m for m in ms if isinstance(m, torch.nn.Dropout)]
        mods = det.__dict__['_btr_stack_mods'] = (n, ms, drops)
    # (the identity of every parameter object as well: the cached lists hold the objects)
    return (tuple(m.training for m in mods[1]), tuple(m.p for m in mods[2]),
            tuple(m.dropout for m in mods[1] if isinstance(m, torch.nn.MultiheadAttention)),
            tuple(id(v) for m in mods[1] for v in m._parameters.values()))


def _base_hook(det):
    from .detector import GroupFreeDetector
    return GroupFreeDetector._after_decoder_layer


def _hook_last_only(det):
    """May the layer hook be called for the last layer alone?  True for the base class's no-op
    and for a class that overrides `_after_decoder_layer` AND declares `_hook_last_only = True`
    in its own body: an inherited flag says nothing about a subclass's new hook (a hook that
    acts on every layer would silently see the last one only)."""
    for klass in type(det).__mro__:
        if '_after_decoder_layer' in klass.__dict__:
            return klass.__dict__['_after_decoder_layer'] is _base_hook(det) or \
                klass.__dict__.get('_hook_last_only', False) is True
    return False
