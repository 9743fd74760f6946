"""Flat gradient sinks (pointnet2/grad_sink.py): a Back-to-Reality step that hands autograd one
flat gradient per native node must leave the SAME `.grad` on every parameter as the step that
hands it a view per parameter -- the two branches' contributions are the same element-wise sums
either way."""
import pytest
import torch

from backtoreality_amd.votenet import config, synthetic, train

pytestmark = pytest.mark.gpu


def _br_steps(cuda, monkeypatch, sinks, jitter=False, steps=2, overlap=False):
    monkeypatch.setenv("BTR_GRAD_SINK", "1" if sinks else "0")
    monkeypatch.setenv("BTR_BR_OVERLAP", "1" if overlap else "0")
    cfg = config.scannet_md40()
    kw = dict(center_refine=True) if jitter else dict(domain_adaptation=True)
    net = train.build_model(cfg, cuda, seed=0, **kw)
    opt = train.make_optimizer(net)
    mk = dict(center_jitter=0.1) if jitter else {}
    bS = synthetic.make_batch(3, 2, 8192, cfg, device=cuda, **mk)
    bT = synthetic.make_batch(103, 2, 8192, cfg, device=cuda, **mk)
    losses, grads = [], None
    for _ in range(steps):
        if jitter:
            loss, _, _ = train.train_step_br_jitter(net, opt, bS, bT, cfg, epoch=30)
        else:
            loss, _, _ = train.train_step_br(net, opt, bS, bT, cfg)
        losses.append(float(loss))
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()
                 if p.grad is not None}
    params = {n: p.detach().clone() for n, p in net.named_parameters()}
    return losses, grads, params


@pytest.mark.parametrize("jitter", [False, True])
def test_sinks_leave_the_same_gradients_and_parameters(cuda, monkeypatch, jitter):
    """One step from the same weights: loss identical, every gradient the same element-wise sum
    of the two branches' contributions.  (Not asserted bit for bit: the backward's input-gradient
    scatter sums a point's neighbour list in the order integer atomics built it, DESIGN.md 8, so
    two runs of the SAME configuration already differ in the last bits -- the bound is that
    run-to-run spread, measured here, not a tolerance of the sinks.)"""
    l0, g0, p0 = _br_steps(cuda, monkeypatch, False, jitter, steps=1)
    l0b, g0b, _ = _br_steps(cuda, monkeypatch, False, jitter, steps=1)
    l1, g1, p1 = _br_steps(cuda, monkeypatch, True, jitter, steps=1)
    assert l0 == l1 == l0b
    assert set(g0) == set(g1)
    for n in g0:
        assert g0[n].shape == g1[n].shape and g1[n].is_contiguous(), n
        scale = float(g0[n].abs().max()) + 1e-30
        spread = float((g0[n] - g0b[n]).abs().max()) / scale
        dev = float((g0[n] - g1[n]).abs().max()) / scale
        assert dev <= 4.0 * spread + 2e-5, (n, dev, spread)
    # (the parameters after Adam's first update are NOT compared: an entry whose gradient is at
    # the summation noise gets lr * g / (|g| + eps) with either sign from two runs of the same
    # configuration -- tests/test_parity_fullsize_gpu.py states that for whole training curves)
    assert set(p0) == set(p1)


def test_sinks_are_what_the_two_forward_step_uses(cuda, monkeypatch):
    """The step's nodes really return flat gradients: the per-parameter accumulation launches
    (one element-wise add per parameter tensor reached by both branches) are gone."""
    from backtoreality_amd.pointnet2 import grad_sink
    calls = []
    orig = grad_sink.Sink._distribute

    def spy(self, t):
        calls.append(int(t.numel()))
        return orig(self, t)

    monkeypatch.setattr(grad_sink.Sink, "_distribute", spy)
    _br_steps(cuda, monkeypatch, True, steps=1)
    # backbone, vote generator, vote aggregation, proposal head + the discriminators' chains
    assert len(calls) >= 4, calls
    assert max(calls) > 500000, calls      # the backbone's flat buffer
    calls.clear()
    _br_steps(cuda, monkeypatch, False, steps=1)
    assert calls == []


def test_gradient_accumulation_over_two_backward_calls(cuda, monkeypatch):
    """No zero_grad between two backward calls: the second call's gradients are ADDED to the
    first's, as with per-parameter gradients."""
    from backtoreality_amd.pointnet2 import grad_sink
    from backtoreality_amd.votenet import loss_helper
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(5, 2, 8192, cfg, device=cuda)

    def run(sinks):
        net = train.build_model(cfg, cuda, seed=0)
        import contextlib
        for _ in range(2):
            with (grad_sink.scope() if sinks else contextlib.nullcontext()):
                end = net({'point_clouds': batch['point_clouds']})
            end.update(batch)
            loss, _ = loss_helper.get_loss(end, cfg)
            loss.backward()
        return {n: p.grad.detach().clone() for n, p in net.named_parameters()
                if p.grad is not None}
    a, b = run(False), run(True)
    assert set(a) == set(b)
    for n in a:
        tol = 1e-5 * float(a[n].abs().max()) + 1e-12
        assert float((a[n] - b[n]).abs().max()) <= tol, n


@pytest.mark.parametrize("jitter", [False, True])
def test_source_head_beside_target_backbone_is_the_same_step(cuda, monkeypatch, jitter):
    """train.two_forwards: the source branch's head on a side stream beside the target branch's
    backbone -- same kernels on the same data, so the loss is identical and the gradients sit
    within the run-to-run spread of the unordered scatter sums; BatchNorm running statistics are
    updated source first, target second, as in the sequential order."""
    cfg = config.scannet_md40()

    def run(overlap):
        monkeypatch.setenv("BTR_BR_OVERLAP", "1" if overlap else "0")
        kw = dict(center_refine=True) if jitter else dict(domain_adaptation=True)
        net = train.build_model(cfg, cuda, seed=0, **kw)
        opt = train.make_optimizer(net)
        mk = dict(center_jitter=0.1) if jitter else {}
        bS = synthetic.make_batch(3, 2, 8192, cfg, device=cuda, **mk)
        bT = synthetic.make_batch(103, 2, 8192, cfg, device=cuda, **mk)
        if jitter:
            loss, eS, eT = train.train_step_br_jitter(net, opt, bS, bT, cfg, epoch=30)
        else:
            loss, eS, eT = train.train_step_br(net, opt, bS, bT, cfg)
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()
                 if p.grad is not None}
        bufs = {n: b.detach().clone() for n, b in net.named_buffers()}
        keep = {k: eS[k].detach().clone() for k in ('aggregated_vote_inds', 'vote_xyz')}
        return float(loss), grads, bufs, keep
    l0, g0, b0, k0 = run(False)
    l0b, g0b, _, _ = run(False)
    l1, g1, b1, k1 = run(True)
    assert l0 == l0b == l1
    for k in k0:
        assert torch.equal(k0[k], k1[k]), k
    for n in b0:      # running statistics, num_batches_tracked: exactly the sequential order's
        assert torch.equal(b0[n], b1[n]), n
    assert set(g0) == set(g1)
    for n in g0:
        scale = float(g0[n].abs().max()) + 1e-30
        spread = float((g0[n] - g0b[n]).abs().max()) / scale
        dev = float((g0[n] - g1[n]).abs().max()) / scale
        assert dev <= 4.0 * spread + 2e-5, (n, dev, spread)


def test_groupfree_two_branch_step_with_sinks(cuda, monkeypatch):
    """GroupFree3D's Back-to-Reality step: the decoder stack's node carries computed operands (the
    heads' concatenated last layers) beside ~300 leaf parameters -- the sink takes the leaves, the
    node keeps returning the computed operands' gradients.  Dropout off (its masks are drawn per
    call); one step from the same weights: loss identical, gradients within the run-to-run
    spread, and the per-parameter accumulation launches gone."""
    from backtoreality_amd.groupfree import train as gf_train
    from backtoreality_amd.pointnet2 import grad_sink
    cfg = config.scannet_md40()
    bS = synthetic.make_batch(0, 2, 8192, cfg, use_height=False, device=cuda)
    bT = synthetic.make_batch(50, 2, 8192, cfg, use_height=False, device=cuda)
    calls = []
    orig = grad_sink.Sink._distribute

    def spy(self, t):
        calls.append(int(t.numel()))
        return orig(self, t)

    monkeypatch.setattr(grad_sink.Sink, "_distribute", spy)

    def run(sinks):
        monkeypatch.setenv("BTR_GRAD_SINK", "1" if sinks else "0")
        torch.manual_seed(0)
        net = gf_train.build_model(cfg, cuda, domain_adaptation=True, dropout=0.0)
        opt = gf_train.make_optimizer(net)
        loss, _, _ = gf_train.train_step_br(net, opt, bS, bT, cfg)
        torch.cuda.synchronize()
        assert not [n for n, p in net.named_parameters() if p.grad is None]
        return float(loss), {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    l0, g0 = run(False)
    assert calls == []
    l0b, g0b = run(False)
    l1, g1 = run(True)
    assert len(calls) >= 3 and max(calls) > 1000000, calls   # backbone, chains, the decoder stack
    assert l0 == l0b == l1
    for n in g0:
        assert g1[n].shape == g0[n].shape and g1[n].is_contiguous(), n
        scale = float(g0[n].abs().max()) + 1e-30
        spread = float((g0[n] - g0b[n]).abs().max()) / scale
        dev = float((g0[n] - g1[n]).abs().max()) / scale
        assert dev <= 4.0 * spread + 2e-5, (n, dev, spread)
