"""TransformerDecoderLayer.forward as ONE call into the library (csrc/decoder.hip), and its
backward as another.

The reference's layer (detection/GroupFree3D/models/transformer.py:36-76) is ~35 torch ops
forward and ~70 backward on (P, B, E) tensors: two nn.MultiheadAttention modules, two linear
layers, three LayerNorms, four dropouts, three residual adds and the position-embedding adds.
`layer_forward(layer, query, key, q_pos, k_pos)` computes the same function from the layer's own
parameters (state-dict keys unchanged) on batch-major channel-last rows -- the layout the
point-wise MLP chains around it (position embeddings, prediction heads) already keep as the
twin of their (B, C, P) outputs, so no transposes happen between them:

    projections / FFN   the bf16x6 NT GEMM of the set-abstraction MLPs (bias epilogue)
    attention cores     csrc/attention.hip on the projection outputs in place
    residual + dropout + LayerNorm      one kernel forward, one backward
    bias / LayerNorm parameter gradients, weight transposes        one launch each per layer

Returns None when the configuration is not covered (CPU tensors, activation other than ReLU,
masks, ...): the caller then runs the op-by-op path.  `BTR_FUSED_DECODER=0` disables it.
"""
import ctypes
import os
import weakref

import torch
import torch.nn as nn
from torch.autograd import Function

from ..pointnet2 import _ext
from . import fused_attention

_call, _lib, _on, _p, _stream = _ext._call, _ext._lib, _ext._on, _ext._p, _ext._stream
_PLANS = weakref.WeakKeyDictionary()   # layer -> {(B, Pq, Pk, p): (desc, plan, grad sizes)}


def enabled():
    return os.environ.get("BTR_FUSED_DECODER", "1") != "0"


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def _u8(nbytes, dev):
    return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=dev)


def _mha_ok(m, E):
    return (isinstance(m, nn.MultiheadAttention) and m._qkv_same_embed_dim and
            m.in_proj_bias is not None and m.bias_k is None and m.bias_v is None and
            not m.add_zero_attn and not getattr(m, "batch_first", False) and
            m.embed_dim == E and m.out_proj.bias is not None and
            _lib.btr_attention_supported(m.head_dim))


def covered(layer, query, key, q_pos, k_pos):
    if not (enabled() and fused_attention.enabled() and query.is_cuda and
            query.dtype == torch.float32 and query.dim() == 3 and key.dim() == 3 and
            key.dtype == torch.float32):
        return False
    B, E, Pq = query.shape
    if key.shape[0] != B or key.shape[1] != E or E % 4 or E > 1024:
        return False
    if layer.activation is not torch.nn.functional.relu:
        return False
    sa, ca = layer.self_attn, layer.multihead_attn
    if not (_mha_ok(sa, E) and _mha_ok(ca, E) and sa.num_heads == ca.num_heads and
            sa.dropout == ca.dropout):
        return False
    ps = {sa.dropout, layer.dropout.p, layer.dropout1.p, layer.dropout2.p, layer.dropout3.p}
    if len(ps) != 1:   # one rate for the whole layer, as the reference constructs it
        return False
    for ln in (layer.norm1, layer.norm2, layer.norm3):
        if not (isinstance(ln, nn.LayerNorm) and ln.elementwise_affine and ln.bias is not None and
                tuple(ln.normalized_shape) == (E,)):
            return False
    if layer.linear1.bias is None or layer.linear2.bias is None or \
            layer.linear1.out_features % 4:
        return False
    for pos, P in ((q_pos, Pq), (k_pos, key.shape[2])):
        if pos is not None and (tuple(pos.shape) != (B, E, P) or pos.dtype != torch.float32):
            return False
    return True


def _params(layer):
    sa, ca = layer.self_attn, layer.multihead_attn
    return (sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
            ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias,
            layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
            layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias,
            layer.norm3.weight, layer.norm3.bias)


def _rows(t, st):
    """(B*P, C) channel-last rows of a (B, C, P) tensor: its twin when the producer kept one."""
    B, C, P = t.shape
    cl = _ext.twin_of(t)
    if cl is not None and tuple(cl.shape) == (B * P, C) and cl.is_contiguous():
        return cl
    cl = _f32((B * P, C), t.device)
    _call(_lib.btr_pm_rows, B, P, C, C, _p(t.contiguous()), _p(cl), st)
    return cl


def _entry(layer, B, Pq, Pk, E, p):
    cache = _PLANS.get(layer)
    if cache is None:
        cache = _PLANS[layer] = {}
    key = (B, Pq, Pk, E, p)
    ent = cache.get(key)
    if ent is None:
        d = _ext.DecoderLayer()
        d.b, d.pq, d.pk, d.e = B, Pq, Pk, E
        d.heads, d.ff, d.dropout = layer.self_attn.num_heads, layer.linear1.out_features, p
        plan = _ext.DecoderPlan()
        _call(_lib.btr_decoder_layer_plan, ctypes.addressof(d), ctypes.addressof(plan))
        F = d.ff
        sizes = [3 * E * E, 3 * E, E * E, E, 3 * E * E, 3 * E, E * E, E, F * E, F, E * F, E,
                 E, E, E, E, E, E]
        assert sum(sizes) == plan.grads_floats
        ent = cache[key] = (d, plan, sizes)
    return ent


_FIELDS = ("sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ca_in_w", "ca_in_b", "ca_out_w",
           "ca_out_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b")


class DecoderLayerFn(Function):
    """(query (B,E,Pq), key (B,E,Pk), q_pos | None, k_pos | None, layer, p, *parameters) ->
    (B, E, Pq); the result carries its channel-last twin."""

    @staticmethod
    def forward(ctx, query, key, q_pos, k_pos, layer, p, *params):
        B, E, Pq = query.shape
        Pk = key.shape[2]
        dev = query.device
        d0, plan, sizes = _entry(layer, B, Pq, Pk, E, p)
        d = _ext.DecoderLayer.from_buffer_copy(d0)   # this call's own copy (seed, pointers)
        for name, t in zip(_FIELDS, params[:12]):
            assert t.is_contiguous()
            setattr(d, name, t.data_ptr())
        for i, ln in enumerate((layer.norm1, layer.norm2, layer.norm3)):
            d.ln_w[i], d.ln_b[i] = params[12 + 2 * i].data_ptr(), params[13 + 2 * i].data_ptr()
            d.ln_eps[i] = float(ln.eps)
        step = None
        if p > 0:
            d.seed = fused_attention._next_seed()
            step = fused_attention.step_counter(dev)
            d.step = step.data_ptr()
        out = _f32((B, E, Pq), dev)
        out_cl = _f32((B * Pq, E), dev)
        saved = _u8(plan.saved_bytes, dev)
        scratch = _u8(plan.fwd_scratch_bytes, dev)
        with _on(query) as dv:
            st = _stream(dv)
            x_cl, key_cl = _rows(query, st), _rows(key, st)
            qpos_cl = _rows(q_pos, st) if q_pos is not None else None
            kpos_cl = _rows(k_pos, st) if k_pos is not None else None
            _call(_lib.btr_decoder_layer_forward, ctypes.addressof(d), ctypes.addressof(plan),
                  _p(x_cl), _p(key_cl), _p(qpos_cl), _p(kpos_cl), _p(out), _p(out_cl), _p(saved),
                  _p(scratch), st)
        _ext.attach_twin(out, out_cl)
        ctx.ent = (d, plan, sizes, step)
        ctx.has_pos = (q_pos is not None, k_pos is not None)
        ctx.pshapes = [t.shape for t in params]
        keep = [saved, x_cl, key_cl] + [t for t in (qpos_cl, kpos_cl) if t is not None]
        ctx.save_for_backward(*keep)
        return out

    @staticmethod
    def backward(ctx, dout):
        d, plan, sizes, _step = ctx.ent
        saved, x_cl, key_cl = ctx.saved_tensors[:3]
        rest = list(ctx.saved_tensors[3:])
        qpos_cl = rest.pop(0) if ctx.has_pos[0] else None
        kpos_cl = rest.pop(0) if ctx.has_pos[1] else None
        dev = dout.device
        dout = dout.contiguous()
        need = ctx.needs_input_grad
        grads = _f32((plan.grads_floats,), dev)
        scratch = _u8(plan.bwd_scratch_bytes, dev)
        dx = _f32((d.b, d.e, d.pq), dev) if need[0] else None
        dkey = _f32((d.b, d.e, d.pk), dev) if (need[1] or need[3]) else None
        dqpos = _f32((d.b, d.e, d.pq), dev) if (need[2] and qpos_cl is not None) else None
        with _on(dout) as dv:
            _call(_lib.btr_decoder_layer_backward, ctypes.addressof(d), ctypes.addressof(plan),
                  _p(x_cl), _p(key_cl), _p(qpos_cl), _p(kpos_cl), _p(dout), _p(saved), _p(grads),
                  _p(dx), _p(dkey), _p(dqpos), _p(scratch), _stream(dv))
        parts = [g.view(s) for g, s in zip(grads.split(sizes), ctx.pshapes)]
        return (dx, dkey if need[1] else None, dqpos,
                dkey if (need[3] and kpos_cl is not None) else None, None, None) + tuple(parts)


def layer_forward(layer, query, key, q_pos, k_pos):
    """query (B,E,Pq), key (B,E,Pk), q_pos / k_pos: position embeddings (B,E,P) or None.
    Returns the layer's output (B,E,Pq), or None when this path does not cover the call."""
    if not covered(layer, query, key, q_pos, k_pos):
        return None
    p = float(layer.dropout.p) if layer.training else 0.0
    return DecoderLayerFn.apply(query, key, q_pos, k_pos, layer, p, *_params(layer))
