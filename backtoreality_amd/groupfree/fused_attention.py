"""nn.MultiheadAttention's forward on the fused attention kernels (csrc/attention.hip).

The reference's decoder layer (detection/GroupFree3D/models/transformer.py:36-76) calls
`MultiheadAttention(query, key, value)[0]` twice per layer -- self-attention over the query
points and cross-attention onto the seed points -- always with key is value, no masks, and only
the output used.  `mha_forward(module, query, key)` computes that output from the module's own
parameters (same state-dict keys: in_proj_weight / in_proj_bias / out_proj.*):

    packed input projection (one GEMM for q, k, v in self-attention; one for q and one for
    k, v in cross-attention)  ->  btr_attention_fwd on the projection outputs IN PLACE (no
    head transposes, nothing of size Lq x Lk in HBM, dropout inside)  ->  output projection

and its backward through btr_attention_bwd (dq / dk / dv written straight into the packed
gradient of the projection output).  Returns None when the configuration is not covered
(CPU tensors, masks, bias_k / add_zero_attn, separate q/k/v widths): the caller then runs the
stock module.  `BTR_FUSED_ATTENTION=0` disables it.

Dropout masks come from a counter-based hash of (a host seed unique per call, a device step
counter, the element index): `bump_step()` once per training step makes a replayed HIP graph
draw fresh masks; the backward of a call reuses the seed its forward drew.
"""
import itertools
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from ..pointnet2 import _ext

_call, _lib, _on, _p, _stream = _ext._call, _ext._lib, _ext._on, _ext._p, _ext._stream
_calls = itertools.count(1)
_STEP = {}   # device -> int64 step counter


def enabled():
    return os.environ.get("BTR_FUSED_ATTENTION", "1") != "0"


def step_counter(device):
    t = _STEP.get(device)
    if t is None:
        t = _STEP[device] = torch.zeros(1, dtype=torch.int64, device=device)
    return t


def bump_step(device):
    """Once per training step (inside the captured region when the step is a HIP graph)."""
    step_counter(device).add_(1)


def _next_seed():
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_calls) * 0xD1B54A32D192ED03) \
        & 0xFFFFFFFFFFFFFFFF


class _AttentionCore(Function):
    """self-attention: qkv (L, B, 3E) packed;  cross: q (Lq, B, E) and kv (Lk, B, 2E) packed."""

    @staticmethod
    def forward(ctx, q_src, kv_src, nhead, dropout_p, seed):
        self_attn = kv_src is None
        Lq, B, Wq = q_src.shape
        E = Wq // 3 if self_attn else Wq
        d = E // nhead
        if self_attn:
            Lk, k_t, v_t = Lq, q_src, q_src
            k_off, v_off, kv_w = E, 2 * E, 3 * E
        else:
            Lk, k_t, v_t = kv_src.shape[0], kv_src, kv_src
            k_off, v_off, kv_w = 0, E, 2 * E
        dev = q_src.device
        out = torch.empty((Lq, B, E), dtype=torch.float32, device=dev)
        lse = torch.empty((B * nhead, Lq), dtype=torch.float32, device=dev)
        scale = float(d) ** -0.5
        step = step_counter(dev) if dropout_p > 0 else None
        with _on(q_src) as dv:
            _call(_lib.btr_attention_fwd, Lq, Lk, B, nhead, d, q_src.data_ptr(), B * Wq, Wq,
                  k_t.data_ptr() + 4 * k_off, v_t.data_ptr() + 4 * v_off, B * kv_w, kv_w,
                  _p(out), _p(lse), scale, float(dropout_p), seed, _p(step), _stream(dv))
        ctx.cfg = (self_attn, nhead, float(dropout_p), seed, Lq, Lk, B, E, d, scale)
        ctx.save_for_backward(q_src, kv_src, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        self_attn, nhead, dropout_p, seed, Lq, Lk, B, E, d, scale = ctx.cfg
        q_src, kv_src, out, lse = ctx.saved_tensors
        dev = dout.device
        dout = dout.contiguous()
        dsum = torch.empty_like(lse)
        dq_src = torch.empty_like(q_src)
        Wq = q_src.shape[2]
        if self_attn:
            k_t = v_t = q_src
            dk_t = dq_src
            k_off, v_off, kv_w = E, 2 * E, 3 * E
            dkv_src = None
        else:
            k_t = v_t = kv_src
            dkv_src = dk_t = torch.empty_like(kv_src)
            k_off, v_off, kv_w = 0, E, 2 * E
        step = step_counter(dev) if dropout_p > 0 else None
        with _on(dout) as dv:
            _call(_lib.btr_attention_bwd, Lq, Lk, B, nhead, d, q_src.data_ptr(), B * Wq, Wq,
                  k_t.data_ptr() + 4 * k_off, v_t.data_ptr() + 4 * v_off, B * kv_w, kv_w,
                  _p(out), _p(dout), _p(lse), _p(dsum), dq_src.data_ptr(), B * Wq, Wq,
                  dk_t.data_ptr() + 4 * k_off, dk_t.data_ptr() + 4 * v_off, B * kv_w, kv_w,
                  scale, dropout_p, seed, _p(step), _stream(dv))
        return dq_src, dkv_src, None, None, None


def _covered(mha, query, key):
    return (enabled() and query.is_cuda and query.dtype == torch.float32 and query.dim() == 3 and
            key.dim() == 3 and mha._qkv_same_embed_dim and mha.in_proj_bias is not None and
            mha.bias_k is None and mha.bias_v is None and not mha.add_zero_attn and
            not getattr(mha, "batch_first", False) and
            _lib.btr_attention_supported(mha.head_dim) and query.shape[2] == mha.embed_dim and
            key.shape[2] == mha.embed_dim and query.shape[1] == key.shape[1])


def mha_forward(mha, query, key):
    """Output of `mha(query, key, value=key)[0]` ((Lq, B, E)), or None when not covered."""
    if not _covered(mha, query, key):
        return None
    E = mha.embed_dim
    p = float(mha.dropout) if mha.training else 0.0
    seed = _next_seed() if p > 0 else 0
    W, bias = mha.in_proj_weight, mha.in_proj_bias
    if key is query:
        qkv = F.linear(query, W, bias)                              # (L, B, 3E)
        core = _AttentionCore.apply(qkv, None, mha.num_heads, p, seed)
    else:
        q = F.linear(query, W[:E], bias[:E])                        # (Lq, B, E)
        kv = F.linear(key, W[E:], bias[E:])                         # (Lk, B, 2E)
        core = _AttentionCore.apply(q, kv, mha.num_heads, p, seed)
    return F.linear(core, mha.out_proj.weight, mha.out_proj.bias)
