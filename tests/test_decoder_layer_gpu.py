"""The whole-layer decoder call (csrc/decoder.hip via groupfree/fused_decoder.py) against the
op-by-op TransformerDecoderLayer (reference: detection/GroupFree3D/models/transformer.py:36-76)
evaluated in float64 with the stock torch modules: output and every gradient."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(E=288, H=8, F=2048, p=0.0, pos=(3, 3)):
    from backtoreality_amd.groupfree.modules import PositionEmbeddingLearned
    from backtoreality_amd.groupfree.transformer import TransformerDecoderLayer
    torch.manual_seed(3)
    sp = PositionEmbeddingLearned(pos[0], E) if pos[0] else None
    cp = PositionEmbeddingLearned(pos[1], E) if pos[1] else None
    layer = TransformerDecoderLayer(E, H, F, p, "relu", self_posembed=sp, cross_posembed=cp)
    for q in layer.parameters():
        if q.dim() > 1:
            torch.nn.init.xavier_uniform_(q)
        else:
            torch.nn.init.normal_(q, 0.0 if q.abs().max() == 0 else 1.0, 0.2)
    return layer.cuda().train()


def _inputs(B, Pq, Pk, E, pos, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g).cuda()
    return (mk(B, E, Pq), mk(B, E, Pk), mk(B, Pq, pos[0]) if pos[0] else None,
            mk(B, Pk, pos[1]) if pos[1] else None, mk(B, E, Pq))


def _run(layer, q, k, qp, kp, w, dtype=torch.float32):
    lay = copy.deepcopy(layer).to(dtype)
    q = q.detach().to(dtype).requires_grad_(True)
    k = k.detach().to(dtype).requires_grad_(True)
    qp = qp.detach().to(dtype) if qp is not None else None
    kp = kp.detach().to(dtype) if kp is not None else None
    out = lay(q, k, qp, kp)
    (out * w.to(dtype)).sum().backward()
    grads = {"query": q.grad, "key": k.grad}
    grads.update({n: t.grad for n, t in lay.named_parameters()})
    return out.detach(), grads


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,Pq,Pk,pos", [(8, 256, 1024, (3, 3)), (2, 100, 300, (6, 3)),
                                         (3, 64, 128, (0, 0))])
def test_layer_matches_float64(B, Pq, Pk, pos):
    from backtoreality_amd.groupfree import fused_decoder
    E = 288
    layer = _layer(E, pos=pos)
    args = _inputs(B, Pq, Pk, E, pos)
    assert fused_decoder.covered(layer, args[0], args[1], None, None)
    o_hip, g_hip = _run(layer, *args)
    os.environ["BTR_FUSED_DECODER"] = "0"
    try:
        o_ops, g_ops = _run(layer, *args)
    finally:
        del os.environ["BTR_FUSED_DECODER"]
    o_ref, g_ref = _run(layer, *args, dtype=torch.float64)
    # same function: both f32 evaluations sit at rounding distance from the f64 one
    assert _rel(o_hip, o_ref) <= 2 * _rel(o_ops, o_ref) + 2e-6
    assert _rel(o_hip, o_ref) < 1e-5
    assert set(g_hip) == set(g_ref)
    for name in g_ref:
        assert g_hip[name] is not None, name
        if float(g_ref[name].abs().max()) < 1e-9:
            # a conv bias in front of a train-mode BatchNorm: the exact gradient is 0 (the chain
            # kernels return that zero, stock f32 ops return rounding noise)
            assert float(g_hip[name].abs().max()) < 1e-3, name
            continue
        e_hip, e_ops = _rel(g_hip[name], g_ref[name]), _rel(g_ops[name], g_ref[name])
        assert e_hip <= 2 * e_ops + 2e-5, (name, e_hip, e_ops)
        assert e_hip < 3e-4, (name, e_hip)


def test_output_carries_channel_last_twin():
    from backtoreality_amd.pointnet2 import _ext
    layer = _layer()
    q, k, qp, kp, _ = _inputs(8, 256, 1024, 288, (3, 3))
    out = layer(q, k, qp, kp)
    twin = _ext.twin_of(out)
    assert twin is not None and twin.shape == (8 * 256, 288)
    assert torch.equal(twin.view(8, 256, 288).transpose(1, 2), out)


def test_dropout_masks_agree_between_forward_and_backward(monkeypatch):
    """With the seed pinned the layer is a deterministic function: its autograd gradient must be
    the derivative of THAT function (a backward that drew other masks would be off by O(p))."""
    from backtoreality_amd.groupfree import fused_attention
    layer = _layer(p=0.1)
    monkeypatch.setattr(fused_attention, "_next_seed", lambda: 0x1234567)
    q, k, qp, kp, w = _inputs(4, 128, 256, 288, (3, 3), seed=5)
    q = q.requires_grad_(True)
    out = layer(q, k, qp, kp)
    assert torch.equal(out, layer(q, k, qp, kp))
    (out * w).sum().backward()
    v = torch.randn_like(q)
    eps = 1e-2
    with torch.no_grad():
        lp = (layer(q + eps * v, k, qp, kp).double() * w).sum()
        lm = (layer(q - eps * v, k, qp, kp).double() * w).sum()
    num = float((lp - lm) / (2 * eps))
    ana = float((q.grad.double() * v).sum())
    assert abs(num - ana) <= 2e-2 * max(abs(num), abs(ana)) + 1e-3, (num, ana)
    # and the masks are real: roughly a tenth of the FFN units that relu keeps are dropped
    monkeypatch.setattr(fused_attention, "_next_seed", lambda: 0x7654321)
    assert not torch.equal(out, layer(q, k, qp, kp))


def test_eval_mode_has_no_dropout():
    layer = _layer(p=0.1).eval()
    q, k, qp, kp, _ = _inputs(2, 64, 128, 288, (3, 3))
    with torch.no_grad():
        a = layer(q, k, qp, kp)
        os.environ["BTR_FUSED_DECODER"] = "0"
        try:
            b = layer(q, k, qp, kp)
        finally:
            del os.environ["BTR_FUSED_DECODER"]
    assert _rel(a, b) < 1e-5
