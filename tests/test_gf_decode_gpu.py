"""PredictHead's decode as one launch (csrc/gf_loss.hip btr_gf_head_decode via
groupfree/fused_decode.py) against the reference's op sequence
(detection/GroupFree3D/models/modules.py:233-262): bit-identical outputs, the query position the
detector builds from them (detector.py:204-230), and the torch-op backward."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(out, base_xyz, mean_size, nh, ns):
    B, C, P = out.shape
    t = out.transpose(2, 1)
    _, cres, _, hrn, ss, srf, _ = torch.split(t, [1, 3, nh, nh, ns, 3 * ns, C - 4 - 2 * nh - 4 * ns],
                                              dim=2)
    center = base_xyz + cres
    hres = hrn * (np.pi / nh)
    ms = mean_size.unsqueeze(0).unsqueeze(0)
    sres = srf.reshape(B, P, ns, 3) * ms
    rec = sres + ms
    pick = torch.argmax(ss, -1).unsqueeze(-1).unsqueeze(-1).expand(-1, -1, 1, 3)
    psize = torch.gather(rec, 2, pick).squeeze(2)
    return center, hres, sres, psize


@pytest.mark.parametrize("B,P,nh,ns,nc,twin", [(4, 256, 1, 18, 18, True), (2, 100, 12, 10, 10, False),
                                               (3, 77, 12, 70, 5, True)])
def test_decode_is_bit_identical_and_differentiable(cuda, B, P, nh, ns, nc, twin):
    from backtoreality_amd.groupfree import fused_decode
    from backtoreality_amd.pointnet2 import _ext
    C = 4 + 2 * nh + 4 * ns + nc
    g = torch.Generator(device="cpu").manual_seed(B * P)
    out = torch.randn(B, C, P, generator=g).to(cuda)
    out[0, 4 + 2 * nh + 3, :5] = out[0, 4 + 2 * nh:4 + 2 * nh + ns, :5].max(0)[0]   # ties -> first
    base = torch.randn(B, P, 3, generator=g).to(cuda)
    mean_size = (torch.rand(ns, 3, generator=g) + 0.2).to(cuda)
    a = out.clone().requires_grad_(True)
    b = out.clone().requires_grad_(True)
    if twin:
        _ext.attach_twin(a, a.detach().transpose(1, 2).reshape(B * P, C).contiguous())
    got = fused_decode.decode(a, base, mean_size, nh, ns)
    assert got is not None
    ref = _reference(b, base, mean_size, nh, ns)
    for x, y in zip(got[:4], ref):
        assert torch.equal(x, y)
    assert torch.equal(got[4], torch.cat([ref[0], ref[3]], -1))
    assert torch.equal(got[5], got[4].transpose(1, 2))
    assert not got[4].requires_grad and not got[5].requires_grad
    ws = [torch.randn(x.shape, generator=g).to(cuda) for x in ref]
    sum((x * w).sum() for x, w in zip(got[:4], ws)).backward()
    sum((x * w).sum() for x, w in zip(ref, ws)).backward()
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-7)
    # partial use: only the centre enters a loss
    a.grad = None
    fused_decode.decode(a, base, mean_size, nh, ns)[0].sum().backward()
    assert float(a.grad[:, 1:4].min()) == 1.0 and float(a.grad.sum()) == 3.0 * B * P


def test_head_and_detector_use_it(cuda, monkeypatch):
    """PredictHead with and without the kernel: same end_points; the detector's query position is
    the kernel's (centre, size) pair."""
    from backtoreality_amd.groupfree.modules import PredictHead
    torch.manual_seed(0)
    mean = np.random.RandomState(0).rand(18, 3).astype(np.float32) + 0.3
    head = PredictHead(18, 1, 18, mean, 256, 288).to(cuda).train()
    feats = torch.randn(2, 288, 256, device=cuda)
    base = torch.randn(2, 256, 3, device=cuda)
    ep_a, ep_b = {}, {}
    torch.manual_seed(1)
    ca, sa = head(feats, base, ep_a, prefix='x_')
    monkeypatch.setenv("BTR_FUSED_GF_DECODE", "0")
    cb, sb = head(feats, base, ep_b, prefix='x_')
    assert getattr(ca, '_btr_query_pos', None) is not None
    assert getattr(cb, '_btr_query_pos', None) is None
    for k in ep_b:
        assert k in ep_a
        if k.endswith('_head_output'):
            continue
        # (two train-mode evaluations of the same BatchNorm chain: identical inputs, same bits)
        assert torch.equal(ep_a[k], ep_b[k]), k
    assert torch.equal(ca._btr_query_pos[0], torch.cat([cb, sb], -1))


def test_cached_transpose_follows_in_place_edits(cuda):
    """PositionEmbeddingLearned keeps the (B, C, P) form of a coordinate tensor on the tensor; an
    in-place edit of the coordinates (its `_version` moves) must not be served the old copy."""
    from backtoreality_amd.groupfree.modules import PositionEmbeddingLearned
    torch.manual_seed(0)
    pe = PositionEmbeddingLearned(3, 288).to(cuda).eval()
    xyz = torch.rand(2, 256, 3, device=cuda)
    with torch.no_grad():
        a = pe(xyz)
        assert getattr(xyz, '_btr_t', None) is not None
        assert torch.equal(pe(xyz), a)
        xyz.mul_(2.0)
        b = pe(xyz)
        assert torch.equal(b, pe(xyz.clone()))
        assert not torch.equal(a, b)
