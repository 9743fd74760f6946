"""The attributes fused layers hand each other (channel-last twins, prefetched geometry) carry
the `_version` of the tensors they were derived from: an in-place edit voids them instead of
feeding the next layer stale data (pointnet2/_twin.py)."""
import pytest
import torch

from backtoreality_amd.pointnet2 import _ext
from backtoreality_amd.pointnet2 import pointnet2_modules as M

pytestmark = pytest.mark.gpu


def test_twin_is_dropped_after_an_in_place_edit(cuda, monkeypatch):
    monkeypatch.setenv("BTR_CHAIN_MIN_ROWS", "0")
    torch.manual_seed(0)
    fp = M.PointnetFPModule(mlp=[64 + 32, 64, 64]).to(cuda).train()
    sa = M.PointnetSAModuleVotes(npoint=64, radius=0.5, nsample=16, mlp=[64, 64, 64],
                                 use_xyz=True, normalize_xyz=True).to(cuda).train()
    unknown, known = torch.rand(2, 256, 3, device=cuda), torch.rand(2, 64, 3, device=cuda)
    out = fp(unknown, known, torch.randn(2, 32, 256, device=cuda),
             torch.randn(2, 64, 64, device=cuda))
    assert _ext.twin_of(out) is not None
    _, ref, _ = sa(unknown, out.detach().clone())          # no twin: transposes itself
    _, fast, _ = sa(unknown, out)                          # uses the twin
    assert torch.equal(ref, fast)
    with torch.no_grad():
        out.mul_(2.0)                                      # a user hook / in-place op
    assert _ext.twin_of(out) is None
    _, got, _ = sa(unknown, out)
    _, want, _ = sa(unknown, out.detach().clone())
    assert torch.equal(got, want), "the stale channel-last twin was consumed"


def test_prefetched_geometry_is_dropped_after_an_in_place_edit(cuda, monkeypatch):
    """Layer-by-layer path: the ball query / sampled coordinates / 3-NN weights attached by
    prefetch_sampling are ignored once the coordinates they came from were modified."""
    from backtoreality_amd.votenet import backbone_module
    monkeypatch.setenv("BTR_NATIVE_BACKBONE", "0")
    torch.manual_seed(0)
    net = backbone_module.Pointnet2Backbone(input_feature_dim=0).to(cuda).train()
    pc = torch.rand(2, 6000, 3, device=cuda) * 4
    pyr = net.prefetch_sampling(pc)
    inds1 = pyr[0][0]
    new_xyz, src = _ext.derived(inds1, "_btr_new_xyz", net._break_up_pc(pc)[0])
    assert _ext.derived(new_xyz, "_btr_ball_query", src) is not None
    torch.cuda.synchronize()
    with torch.no_grad():
        src.add_(0.5)
    assert _ext.derived(inds1, "_btr_new_xyz", src) is None
    assert _ext.derived(new_xyz, "_btr_ball_query", src) is None
