"""Fused attention core (csrc/attention.hip via groupfree/fused_attention.py) against
torch.nn.MultiheadAttention -- the module the reference vendors a copy of
(detection/GroupFree3D/models/multi_head_attention.py) -- on the GPU: output and every gradient
(query, key, in/out projection weights and biases), self- and cross-attention at the decoder's
sizes and at ragged ones; dropout: keep fraction, scaling, and a backward that matches the
forward's mask (finite differences through the same seed)."""
import copy

import pytest
import torch
import torch.nn as nn

from backtoreality_amd.groupfree import fused_attention

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("Lq,Lk,B,E,H,self_attn", [
    (256, 256, 4, 288, 8, True),      # decoder self-attention (d = 36)
    (256, 1024, 4, 288, 8, False),    # decoder cross-attention
    (77, 77, 2, 64, 4, True),         # ragged: rows not a multiple of the 32 / 64 tiles, d = 16
    (50, 130, 3, 120, 2, False),      # d = 60
    (33, 65, 1, 8, 8, False),         # d = 1
])
def test_matches_multihead_attention(cuda, Lq, Lk, B, E, H, self_attn):
    torch.manual_seed(0)
    mha = nn.MultiheadAttention(E, H, dropout=0.0).to(cuda)
    with torch.no_grad():
        mha.in_proj_bias.uniform_(-0.2, 0.2)
        mha.out_proj.bias.uniform_(-0.2, 0.2)
    ref = copy.deepcopy(mha)
    q0 = torch.randn(Lq, B, E, device=cuda)
    k0 = q0 if self_attn else torch.randn(Lk, B, E, device=cuda)
    w = torch.randn(Lq, B, E, device=cuda)
    res = []
    for mod, fused in ((ref, False), (mha, True)):
        q = q0.clone().requires_grad_(True)
        k = q if self_attn else k0.clone().requires_grad_(True)
        out = fused_attention.mha_forward(mod, q, k) if fused else mod(q, k, value=k)[0]
        assert out is not None
        (out * w).sum().backward()
        g = {"out": out.detach(), "dq": q.grad}
        if not self_attn:
            g["dk"] = k.grad
        g.update({"d" + n: p.grad for n, p in mod.named_parameters()})
        res.append(g)
    want, got = res
    assert set(want) == set(got)
    for key in want:
        assert got[key].shape == want[key].shape, key
        assert _rel(got[key], want[key]) < (1e-5 if key == "out" else 1e-4), (
            key, _rel(got[key], want[key]))


def test_not_covered_configurations_fall_back(cuda):
    mha = nn.MultiheadAttention(64, 4).to(cuda)
    q = torch.randn(10, 2, 64, device=cuda)
    assert fused_attention.mha_forward(mha, q.cpu(), q.cpu()) is None          # CPU tensors
    assert fused_attention.mha_forward(nn.MultiheadAttention(64, 4, kdim=32, vdim=32).to(cuda),
                                       q, q) is None                           # separate widths
    assert fused_attention.mha_forward(nn.MultiheadAttention(64, 4, add_zero_attn=True).to(cuda),
                                       q, q) is None
    assert fused_attention.mha_forward(nn.MultiheadAttention(640, 4).to(cuda),
                                       torch.randn(10, 2, 640, device=cuda),
                                       torch.randn(10, 2, 640, device=cuda)) is None   # d = 160
    assert fused_attention.mha_forward(mha, q, q) is not None


def test_dropout_mask_statistics_and_consistent_backward(cuda):
    """With v = 1 the output row is sum_j keep_ij p_ij / (1 - p): its mean is 1 and its spread
    is the mask's; eval mode is deterministic; the backward uses the forward's mask: the
    directional derivative from two forwards with the same seed equals <grad, direction>."""
    from backtoreality_amd.groupfree.fused_attention import _AttentionCore
    L, B, E, H, p = 256, 2, 64, 4, 0.3
    torch.manual_seed(1)
    qkv = torch.randn(L, B, 3 * E, device=cuda) * 0.3
    qkv[:, :, 2 * E:] = 1.0                                   # v = 1
    fused_attention.bump_step(cuda)
    out = _AttentionCore.apply(qkv, None, H, p, 1234)
    assert abs(float(out.mean()) - 1.0) < 0.02                # E[keep / (1 - p)] = 1
    assert 0.02 < float(out.std()) < 0.3                      # ... but rows do vary
    out2 = _AttentionCore.apply(qkv, None, H, p, 1234)
    assert torch.equal(out, out2)                             # same seed, same step: same mask
    out3 = _AttentionCore.apply(qkv, None, H, p, 99)
    assert not torch.equal(out, out3)
    fused_attention.bump_step(cuda)
    assert not torch.equal(out, _AttentionCore.apply(qkv, None, H, p, 1234))   # new step

    x = (torch.randn(L, B, 3 * E, device=cuda) * 0.5).double()
    direction = torch.randn_like(x)
    w = torch.randn(L, B, E, device=cuda)

    def f(t):
        return float((_AttentionCore.apply(t.float().contiguous(), None, H, p, 77).double()
                      * w.double()).sum())
    xr = x.float().requires_grad_(True)
    (_AttentionCore.apply(xr, None, H, p, 77) * w).sum().backward()
    analytic = float((xr.grad.double() * direction).sum())
    eps = 1e-2
    numeric = (f(x + eps * direction) - f(x - eps * direction)) / (2 * eps)
    assert abs(analytic - numeric) <= 2e-2 * abs(numeric) + 1e-3, (analytic, numeric)


def test_decoder_layer_uses_the_fused_core(cuda, monkeypatch):
    from backtoreality_amd.groupfree.transformer import TransformerDecoderLayer
    torch.manual_seed(2)
    layer = TransformerDecoderLayer(288, 8, 512, dropout=0.0).to(cuda)
    q = torch.randn(2, 288, 64, device=cuda)
    k = torch.randn(2, 288, 200, device=cuda)
    monkeypatch.setenv("BTR_FUSED_DECODER", "0")   # the op-by-op form of the layer
    calls = []
    real = fused_attention._AttentionCore.apply
    monkeypatch.setattr(fused_attention._AttentionCore, "apply",
                        staticmethod(lambda *a: calls.append(1) or real(*a)))
    out = layer(q, k, None, None)
    assert len(calls) == 2
    monkeypatch.setenv("BTR_FUSED_ATTENTION", "0")
    ref = layer(q, k, None, None)
    assert len(calls) == 2
    assert _rel(out, ref) < 1e-5
