"""Decode of a PredictHead's raw output as ONE launch (csrc/gf_loss.hip btr_gf_head_decode).

The reference (detection/GroupFree3D/models/modules.py:233-262) derives the box centre, the
heading / size residuals in metres and the size of the arg-max size class with six elementwise /
arg-max / gather ops per head, and the detector (detector.py:204-230) clones centre and size and
concatenates them into the next decoder layer's query position: ~10 launches per head, seven
heads per step, every one of them at the launch-latency floor.  `decode(...)` returns the same
tensors from one kernel (bit-identical: products and sums are rounded separately like the torch
ops), plus the query position (B, P, 6) and its (B, 6, P) transpose for the position embedding.

The outputs stay differentiable with respect to the head output: the backward is written with
torch ops (it only runs for a loss that reads these tensors; the fused per-head loss reads the raw
head output, so in the training step of train.py it never runs).  `BTR_FUSED_GF_DECODE=0`
disables it.
"""
import math
import os

import torch
from torch.autograd import Function

from ..pointnet2 import _ext

_call, _lib, _on, _p, _stream = _ext._call, _ext._lib, _ext._on, _ext._p, _ext._stream


def enabled():
    return os.environ.get("BTR_FUSED_GF_DECODE", "1") != "0"


class HeadDecode(Function):
    """(out (B, C, P), base_xyz (B, P, 3), mean_size (NS, 3), nh, ns) ->
    center, heading_residuals, size_residuals (B, P, NS, 3), pred_size, query_pos, query_pos_t"""

    @staticmethod
    def forward(ctx, out, base_xyz, mean_size, nh, ns):
        B, C, P = out.shape
        dev = out.device
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        center, hres, sres, psize = f(B, P, 3), f(B, P, nh), f(B, P, ns, 3), f(B, P, 3)
        qpos, qpos_t = f(B, P, 6), f(B, 6, P)
        cl = _ext.twin_of(out)
        if cl is not None and cl.is_contiguous() and cl.shape[0] == B * P:
            src, sb, sp, sc = cl, P * cl.shape[1], cl.shape[1], 1
        else:
            src = out.contiguous()
            sb, sp, sc = C * P, 1, P
        base = base_xyz.contiguous()
        with _on(out) as dv:
            _call(_lib.btr_gf_head_decode, B, P, nh, ns, _p(src), sb, sp, sc, _p(base),
                  _p(mean_size), _p(center), _p(hres), _p(sres), _p(psize), _p(qpos), _p(qpos_t),
                  _stream(dv))
        ctx.dims = (B, C, P, nh, ns)
        ctx.save_for_backward(out, mean_size)
        ctx.mark_non_differentiable(qpos, qpos_t)
        ctx.set_materialize_grads(False)
        return center, hres, sres, psize, qpos, qpos_t

    @staticmethod
    def backward(ctx, g_center, g_hres, g_sres, g_psize, _gq, _gqt):
        out, mean_size = ctx.saved_tensors
        g = fold_grads(out, mean_size, ctx.dims, g_center, g_hres, g_sres, g_psize)
        g_base = g_center if ctx.needs_input_grad[1] else None
        return g, g_base, None, None, None


def fold_grads(out, mean_size, dims, g_center, g_hres, g_sres, g_psize):
    """Gradient w.r.t. the raw head output (B, C, P) of the gradients of the decoded tensors
    (each may be None)."""
    B, C, P, nh, ns = dims
    o_hres, o_ss, o_sr = 4 + nh, 4 + 2 * nh, 4 + 2 * nh + ns
    g = torch.zeros((B, P, C), dtype=out.dtype, device=out.device)
    if g_center is not None:
        g[..., 1:4] += g_center
    if g_hres is not None:
        g[..., o_hres:o_hres + nh] += g_hres * (math.pi / nh)
    if g_sres is not None:
        g[..., o_sr:o_sr + 3 * ns] += (g_sres * mean_size).reshape(B, P, 3 * ns)
    if g_psize is not None:
        pick = torch.argmax(out[:, o_ss:o_ss + ns, :], 1)                    # (B, P)
        cols = o_sr + pick.unsqueeze(-1) * 3 + torch.arange(3, device=out.device)   # (B, P, 3)
        g.scatter_add_(2, cols, g_psize * mean_size[pick])
    return g.transpose(1, 2)


def decode(out, base_xyz, mean_size, nh, ns):
    """None when not covered (CPU tensors, other dtypes, BTR_FUSED_GF_DECODE=0)."""
    if not (enabled() and out.is_cuda and out.dtype == torch.float32 and out.dim() == 3 and
            base_xyz.dtype == torch.float32 and out.shape[1] >= 4 + 2 * nh + 4 * ns and
            tuple(base_xyz.shape) == (out.shape[0], out.shape[2], 3)):
        return None
    return HeadDecode.apply(out, base_xyz, mean_size, nh, ns)
