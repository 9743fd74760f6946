"""The whole-backbone library calls (csrc/backbone.hip: btr_backbone_sampling / _forward /
_backward behind pointnet2/fused_backbone.py) against the layer-by-layer path they replace
(one FusedSALayer / PointwiseChain autograd node per layer): the same kernels in the same
order, so every output, gradient and BatchNorm buffer must be BIT-identical.
reference: models/backbone_module.py:83-133."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(native, net0, pc, prefetch, douts_seed=0, extra_grads=False):
    from backtoreality_amd.pointnet2 import fused_backbone
    os.environ["BTR_NATIVE_BACKBONE"] = "1" if native else "0"
    # (the layer-by-layer path leaves chains below 2 048 rows to the stock torch ops; the
    # library calls always run the fused kernels: compare like with like)
    os.environ["BTR_CHAIN_MIN_ROWS"] = "0"
    try:
        net = copy.deepcopy(net0)
        net.train()
        sampling = net.prefetch_sampling(pc) if prefetch else None
        if native:
            assert net._native_entry(pc) is not None
            if prefetch:
                assert isinstance(sampling, fused_backbone.Sampling)
        end = net(pc, sampling=sampling)
        g = torch.Generator(device="cpu").manual_seed(douts_seed)
        keys = ["fp2_features"] + (["sa2_features", "sa4_features"] if extra_grads else [])
        loss = 0
        for k in keys:
            w = torch.randn(end[k].shape, generator=g).to(pc.device)
            loss = loss + (end[k] * w).sum()
        loss.backward()
        torch.cuda.synchronize()
        out = {k: v.detach().clone() for k, v in end.items() if torch.is_tensor(v)}
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        bufs = {n: b.detach().clone() for n, b in net.named_buffers()}
        return out, grads, bufs
    finally:
        os.environ.pop("BTR_NATIVE_BACKBONE", None)
        os.environ.pop("BTR_CHAIN_MIN_ROWS", None)


@pytest.mark.parametrize("shape,feat,prefetch,extra", [
    ((2, 4096), 1, False, False),
    ((2, 4096), 1, True, False),
    ((2, 6000), 0, True, True),     # xyz only (GroupFree3D-style input), bucket FPS + ball query
    ((3, 20000), 1, False, True),
])
def test_native_backbone_equals_layer_by_layer(cuda, shape, feat, prefetch, extra):
    from backtoreality_amd.votenet import backbone_module
    torch.manual_seed(3)
    net = backbone_module.Pointnet2Backbone(input_feature_dim=feat).to(cuda)
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N = shape
    pc = torch.rand((B, N, 3 + feat), generator=g)
    pc[..., :3] = pc[..., :3] * torch.tensor([6.0, 5.0, 2.5])
    pc = pc.to(cuda)
    ref = _run(False, net, pc, prefetch, extra_grads=extra)
    got = _run(True, net, pc, prefetch, extra_grads=extra)
    for name, (r, gt) in (("outputs", (ref[0], got[0])), ("buffers", (ref[2], got[2]))):
        assert sorted(r) == sorted(gt), name
        for k in r:
            assert r[k].shape == gt[k].shape and r[k].dtype == gt[k].dtype, (name, k)
            assert torch.equal(r[k], gt[k]), "%s %s: max diff %g" % (
                name, k, float((r[k].float() - gt[k].float()).abs().max()))
    # gradients: the input-gradient scatter sums a point's neighbour rows in the order an
    # integer atomic cursor filled its list (csr_small_kernel), which varies from run to run
    # even on one path -- rounding-level agreement, not bit equality
    assert sorted(ref[1]) == sorted(got[1])
    for k in ref[1]:
        r, gt = ref[1][k], got[1][k]
        assert r.shape == gt.shape
        err = float((r - gt).abs().max() / (r.abs().max() + 1e-20))
        assert err < 2e-5, "gradient %s: rel err %g" % (k, err)


def test_stale_sampling_handle_is_refused(cuda):
    """A handle computed for one tensor must not be consumed for another, nor after an in-place
    edit of the cloud (the indices would describe other coordinates)."""
    from backtoreality_amd.votenet import backbone_module
    torch.manual_seed(0)
    net = backbone_module.Pointnet2Backbone(input_feature_dim=1).to(cuda).train()
    pc = torch.rand((2, 4096, 4), device=cuda)
    other = pc.clone()
    h = net.prefetch_sampling(pc)
    with pytest.raises(RuntimeError, match="does not belong"):
        net(other, sampling=h)
    pc.mul_(1.5)
    with pytest.raises(RuntimeError, match="does not belong"):
        net(pc, sampling=h)
    end = net(pc, sampling=net.prefetch_sampling(pc))
    assert end["fp2_features"].shape == (2, 256, 1024)


def test_deepcopy_and_pickle_survive_the_cached_plan(cuda):
    import pickle
    from backtoreality_amd.votenet import backbone_module
    net = backbone_module.Pointnet2Backbone(input_feature_dim=1).to(cuda).train()
    pc = torch.rand((2, 4096, 4), device=cuda)
    net(pc)
    net2 = pickle.loads(pickle.dumps(copy.deepcopy(net)))
    assert net2(pc)["fp2_features"].shape == (2, 256, 1024)
