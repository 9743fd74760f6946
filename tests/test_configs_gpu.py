"""Parity at the other BASELINE.json configurations: they are test cases, not
bench lines.  Each runs the full-size forward+backward on the GPU through the fused HIP path
and checks (a) every sampling index against the CPU oracle, bit-exact, (b) losses and
gradients against the unfused op-by-op path (the reference's own composition) within 1e-4."""
import os

import numpy as np
import pytest
import torch

import oracle
from backtoreality_amd.votenet import backbone_module, config, loss_helper, synthetic, train

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))


def _worst_grad_dev(g_f, g_u):
    """Largest relative L2 deviation over the parameters whose gradient is not pure rounding
    noise (conv biases in front of a BatchNorm have an exactly-zero true gradient).
    Gradients flow through arg-max / ReLU selections, so two valid f32 implementations may
    differ by more than rounding on individual entries (max-norm deviations of 1e-2 were
    observed between the fused and the op-by-op path); the bound is 1e-2 in relative L2
    (measured: 2e-3 .. 5e-3 on inputs without a near-tie element, tools/diag_c5_grads.py)
    while features and losses are held to 1e-4."""
    gmax = max(float(g.abs().max()) for g in g_u.values())

    def l2(n):
        return float((g_f[n] - g_u[n]).norm() / (g_u[n].norm() + 1e-20))
    return max(l2(n) for n in g_u if float(g_u[n].abs().max()) > 1e-4 * gmax)


def _gradients_as_close_to_float64_as_the_reference_formulation(g_f, g_u, g64, max_events=2):
    """The gradient bar of a full-size step, read against a float64 evaluation of the same step
    (tests/f64_path.py) instead of against the other float32 path.

    Why: at 2 x 8 x 40 000 points the fused HIP path and the nine-op + torch composition agree
    to 1e-4 on every forward quantity and to 1e-6 on the loss, and still differ by 1 - 3 % in
    relative L2 on many gradient tensors.  Measured (tools/diag_c3_f64.py, profiles/
    r06_e_diag_c3_f64.txt): that is not one flipped max-pool decision -- the deviation starts at
    the proposal head's first layer and grows smoothly towards SA1, and the reference's OWN
    float32 formulation sits just as far from float64 (seed pair 0: nine-op 0.45 - 1.7 %, fused
    0.78 - 2.0 %, the two against each other 0.9 - 2.5 % = the two errors in quadrature).  A
    detection loss pulls positives and negatives apart; its parameter gradients are sums that
    cancel, and float32 defines them to ~1e2 x the forward's rounding.  So:
      * every live parameter's gradient must be within 2 x the worst error of the nine-op float32
        path (+ 5e-3) of float64 -- the fused path may be as inexact as the formulation it
        replaces, not more;
      * discrete events (a ReLU gate or a pool tie within rounding of its threshold) move ONE
        tensor by more: at most `max_events` tensors may exceed the bound, none by 0.1.
    A gradient path that is off in one layer moves that layer's weight, scale and shift and
    everything upstream: more than two tensors.  Returns (yardstick, bound, the exceptions)."""
    gmax = max(float(g.abs().max()) for g in g64.values())
    live = [n for n in g64 if float(g64[n].abs().max()) > 1e-4 * gmax]

    def l2(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))
    err_f = {n: l2(g_f[n], g64[n]) for n in live}
    err_u = {n: l2(g_u[n], g64[n]) for n in live}
    yard = max(err_u.values())
    bound = 2.0 * yard + 5e-3
    over = sorted(((err_f[n], n) for n in live if err_f[n] > bound), reverse=True)
    report = sorted(((err_f[n], err_u[n], n) for n in live), reverse=True)[:10]
    assert len(over) <= max_events, (yard, bound, over, report)
    assert all(e < 0.1 for e, _ in over), (yard, bound, over)
    return yard, bound, over


def _votenet_step(cfg, batch, dev, fused, monkeypatch, num_proposal=256, vote_inds=None):
    """`vote_inds`: proposals to aggregate around (PointnetSAModuleVotes' `inds` hook) instead
    of the layer's own FPS over the computed votes."""
    monkeypatch.setenv("BTR_FUSED_SA", "1" if fused else "0")
    net = train.build_model(cfg, dev, input_feature_dim=1, num_proposal=num_proposal, seed=0)
    if vote_inds is not None:
        sa = net.pnet.vote_aggregation
        own = sa.forward
        sa.forward = lambda xyz, features=None, inds=None: own(xyz, features, vote_inds)
    end = net({'point_clouds': batch['point_clouds']})
    end.update(batch)
    loss, end = loss_helper.get_loss(end, cfg)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    return loss.detach(), end, grads


def test_c5_matterport_80k_points(cuda, monkeypatch):
    """C5: 80 000-point scenes (12 x 12 x 3 m), 13 classes / 12 heading bins, batch 4."""
    cfg = config.matterport_md40()
    B = 4   # BASELINE configs[4]: batch 4 per GPU
    # (scenes 8..11: scenes 0..3 hold a max-pool element of SA4 within float32 noise of a tie
    # whose flip moves one row of one weight gradient by 0.3 and everything upstream by 4 %:
    # tools/diag_c5_grads.py)
    batch = synthetic.make_batch(8, B, 80000, cfg, extent_scale=1.7, device=cuda)
    # op-by-op path first; the fused path then aggregates around the SAME proposals (the vote
    # FPS samples computed floats, so it may legitimately differ between two f32 paths), which
    # makes the loss and every gradient comparable unconditionally
    loss_u, end_u, g_u = _votenet_step(cfg, batch, cuda, False, monkeypatch)
    loss_f, end_f, g_f = _votenet_step(cfg, batch, cuda, True, monkeypatch,
                                       vote_inds=end_u['aggregated_vote_inds'])
    # indices vs the oracle (FPS at 80 000 points uses two bucket slots per lane)
    xyz = batch['point_clouds'][..., :3].cpu().numpy()
    ref1 = oracle.furthest_point_sampling(xyz, 2048)
    np.testing.assert_array_equal(end_f['sa1_inds'].cpu().numpy(), ref1)
    np.testing.assert_array_equal(end_f['sa2_inds'].cpu().numpy(),
                                  np.tile(np.arange(1024, dtype=np.int32), (B, 1)))
    assert torch.equal(end_f['sa1_inds'], end_u['sa1_inds'])
    assert torch.equal(end_f['aggregated_vote_inds'], end_u['aggregated_vote_inds'])
    assert _rel(end_f['fp2_features'], end_u['fp2_features']) < 1e-4
    # two float32 paths, each within 1e-4 of the reference's result: 2e-4 between them
    assert _rel(end_f['aggregated_vote_features'], end_u['aggregated_vote_features']) < 2e-4
    assert abs(float(loss_f) - float(loss_u)) / abs(float(loss_u)) < 1e-4
    worst = _worst_grad_dev(g_f, g_u)
    assert worst < 1e-2, worst


def test_c5_scenes_with_a_near_tie_state_the_bound_per_flip(cuda, monkeypatch):
    """The C5 test above uses scenes 8..11 because scenes 0..3 hold ONE max-pool element of SA4
    (layer 2, channel 83) within float32 noise of a tie between two different neighbours
    (tools/diag_c5_grads.py).  Instead of only stepping around it, state what such a flip may
    do, on those very scenes -- two correct f32 implementations may pick either neighbour:
      * nothing in the forward moves (the two candidates hold the same value to rounding): loss
        and features stay at 1e-4;
      * the gradients of every layer the backward reaches BEFORE the flipped pool (proposal
        head, vote aggregation, voting module, fp1 / fp2) do not depend on the choice: they keep
        the normal bound;
      * in the flipped layer's weight gradient exactly the flipped channel's row moves: all but
        <= 2 rows keep the normal bound row by row;
      * everything upstream of it (SA4's first layers, SA3..SA1) receives that one element's
        gradient through another neighbour: 4 % relative L2 was measured, 0.1 is the bound.
    On a box / build where the element does not flip every bound holds trivially."""
    cfg = config.matterport_md40()
    B = 4
    batch = synthetic.make_batch(0, B, 80000, cfg, extent_scale=1.7, device=cuda)
    loss_u, end_u, g_u = _votenet_step(cfg, batch, cuda, False, monkeypatch)
    loss_f, end_f, g_f = _votenet_step(cfg, batch, cuda, True, monkeypatch,
                                       vote_inds=end_u['aggregated_vote_inds'])
    assert torch.equal(end_f['sa1_inds'], end_u['sa1_inds'])
    assert _rel(end_f['fp2_features'], end_u['fp2_features']) < 1e-4
    assert _rel(end_f['aggregated_vote_features'], end_u['aggregated_vote_features']) < 2e-4
    assert abs(float(loss_f) - float(loss_u)) / abs(float(loss_u)) < 1e-4
    gmax = max(float(g.abs().max()) for g in g_u.values())

    def l2(n):
        return float((g_f[n] - g_u[n]).norm() / (g_u[n].norm() + 1e-20))
    live = [n for n in g_u if float(g_u[n].abs().max()) > 1e-4 * gmax]
    upstream = ('backbone_net.sa1.', 'backbone_net.sa2.', 'backbone_net.sa3.', 'backbone_net.sa4.')
    flipped = 'backbone_net.sa4.mlp_module.layer2.conv.weight'
    assert flipped in g_u
    for n in live:
        if not n.startswith(upstream):
            assert l2(n) < 1e-2, (n, l2(n))       # reached before the pool: unaffected
        elif n != flipped:
            assert l2(n) < 0.1, (n, l2(n))        # reached through the flipped element
    wf, wu = g_f[flipped].flatten(1), g_u[flipped].flatten(1)
    # (row deviations relative to the tensor's largest row: a row whose own norm is small is not
    # "moved" by ordinary rounding of the others' scale)
    row_dev = (wf - wu).norm(dim=1) / float(wu.norm(dim=1).max())
    moved = int((row_dev > 1e-2).sum())
    own = (wf - wu).norm(dim=1) / (wu.norm(dim=1) + 1e-20)
    assert moved <= 2, (moved, row_dev.topk(8), own.topk(8), {n: l2(n) for n in live})


@pytest.mark.parametrize("first", [24, 0])
def test_c3_back_to_reality_full_size_step(cuda, monkeypatch, first):
    """C3 (BASELINE configs[2]) at its full per-GPU size: source AND target batch of 8 x 40 000
    points through the same VoteNet_DA (models/votenet_DA.py:123-176), get_loss_DA, one backward
    -- the fused HIP path against the nine-op + torch composition with both branches' proposals
    and vote-ball neighbour lists pinned to the op-by-op run's (the two ops downstream of computed
    floats, as the golden tests pin them): every sampling index equal, features / discriminator
    outputs / every loss term 1e-4 -- and the gradients against a FLOAT64 evaluation of the same
    step on the GPU: the fused path must be as close to it as the reference's own float32
    formulation is (_gradients_as_close_to_float64_as_the_reference_formulation).  Round 5 held the
    two float32 paths to 5e-2 against each other, which could not tell a flipped pool from a
    gradient path that is 3 % off; the float64 yardstick shows what float32 defines (seed pair
    24: nine-op 0.05 - 0.5 %, fused 0.5 - 1.1 % with one BatchNorm shift of SA4's pooled layer at
    3 % -- one gate flipped; pair 0: both 0.5 - 2 %) and bounds the fused path by it."""
    import f64_path
    cfg = config.scannet_md40()
    batch_S = synthetic.make_batch(first, 8, 40000, cfg, device=cuda)
    batch_T = synthetic.make_batch(100000 + first, 8, 40000, cfg, device=cuda)
    loss_u, uS, uT, g_u, idx_u = f64_path.br_step(cfg, batch_S, batch_T, cuda, fused=False)
    pins = (uS['aggregated_vote_inds'], uT['aggregated_vote_inds'])
    loss_f, fS, fT, g_f, idx_f = f64_path.br_step(cfg, batch_S, batch_T, cuda, fused=True,
                                                  vote_inds=pins, vote_idx=idx_u)
    assert len(idx_u) == len(idx_f) == 2
    # (reported, not asserted: how many neighbour slots the fused run's own queries, on its own
    # votes, filled differently)
    print("c3 pair %d: vote-ball neighbour slots that differ between the two f32 paths: %s of %d"
          % (first, [int((a != b).sum()) for a, b in zip(idx_f, idx_u)], idx_u[0].numel()))
    for tag, f, u in (("S", fS, uS), ("T", fT, uT)):
        for k in ('sa1_inds', 'sa2_inds', 'fp2_inds', 'aggregated_vote_inds', 'objectness_label'):
            assert torch.equal(f[k], u[k]), (tag, k)
        assert _rel(f['fp2_features'], u['fp2_features']) < 1e-4, tag
        assert _rel(f['aggregated_vote_features'], u['aggregated_vote_features']) < 2e-4, tag
        for k in ('global_d_pred', 'local_d_pred', 'center'):
            assert _rel(f[k], u[k]) < 2e-4, (tag, k)
        for k in ('vote_loss', 'objectness_loss', 'center_loss', 'size_cls_loss', 'sem_cls_loss'):
            if k in u:
                a, b = float(f[k]), float(u[k])
                assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (tag, k, a, b)
    assert abs(float(loss_f) - float(loss_u)) / abs(float(loss_u)) < 1e-4
    assert set(g_f) == set(g_u)
    del uS, uT, fS, fT
    loss64, _, _, g64, _ = f64_path.br_step(cfg, batch_S, batch_T, cuda, fused=False,
                                            vote_inds=pins, vote_idx=idx_u, float64=True)
    assert set(g64) == set(g_f)
    for name, lv in (("fused", loss_f), ("nine-op", loss_u)):
        assert abs(float(lv) - float(loss64)) <= 1e-5 * abs(float(loss64)), (name, lv, loss64)
    yard, bound, over = _gradients_as_close_to_float64_as_the_reference_formulation(g_f, g_u, g64)
    print("c3 pair %d: nine-op f32 worst gradient error vs float64 %.4f -> bound %.4f; fused "
          "tensors beyond it: %s" % (first, yard, bound, over))


def test_c4_groupfree_backbone_50k_no_features(cuda, monkeypatch):
    """C4: GroupFree3D-style backbone: xyz only (no height channel), 50 000 points, fp2 -> 288
    channels (detection/GroupFree3D/models/backbone_module.py:33-75); configs[3]: batch 4."""
    B = 4
    pc = torch.from_numpy(np.stack([synthetic.make_scene(70 + i, 50000, use_height=False)[
        'point_clouds'] for i in range(B)], 0)).to(cuda)

    def run(fused):
        monkeypatch.setenv("BTR_FUSED_SA", "1" if fused else "0")
        torch.manual_seed(0)
        net = backbone_module.Pointnet2Backbone(input_feature_dim=0, fp2_out=288).to(cuda)
        end = net(pc)
        end['fp2_features'].square().mean().backward()
        return end, {n: p.grad.detach().clone() for n, p in net.named_parameters()}

    end_f, g_f = run(True)
    assert end_f['fp2_features'].shape == (B, 288, 1024)
    ref1 = oracle.furthest_point_sampling(pc.cpu().numpy(), 2048)
    np.testing.assert_array_equal(end_f['sa1_inds'].cpu().numpy(), ref1)
    end_u, g_u = run(False)
    for k in ('sa1_features', 'sa2_features', 'sa4_features', 'fp2_features'):
        assert _rel(end_f[k], end_u[k]) < 1e-4, k
    worst = _worst_grad_dev(g_f, g_u)
    assert worst < 1e-2, worst


def test_ddp_wrapper_on_the_gpu_path(cuda, monkeypatch):
    """DistributedDataParallel (RCCL backend, world_size 1) around the fused path: its
    bucket/allreduce hooks must accept the gradients the custom autograd function returns."""
    import torch.distributed as dist
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29531")
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(0, 2, 8192, cfg, device=cuda)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        net = train.build_model(cfg, cuda)
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[cuda.index or 0],
                                                        broadcast_buffers=False)
        opt = train.make_optimizer(net)
        l0, _ = train.train_step(ddp, opt, batch, cfg)
        l1, _ = train.train_step(ddp, opt, batch, cfg)
        assert torch.isfinite(l0) and torch.isfinite(l1)
        assert all(p.grad is not None for p in net.parameters())
    finally:
        dist.destroy_process_group()


def test_sampling_prefetch_gives_identical_training(cuda):
    """Software-pipelined loop (next batch's FPS pyramid under this step's backward) vs the
    plain loop: the prefetched indices are bit-identical to the inline ones and the first loss
    is bit-identical (the forward is deterministic).  Later steps are compared loosely: the
    backward of the nine ops uses f32 atomics like the reference, so even two PLAIN runs
    diverge (tools/diag_nondeterminism.py: step-2 losses differ by 3e-5, the vote FPS then
    picks different proposals and step-3 losses differ by 5 %)."""
    cfg = config.scannet_md40()
    batches = [synthetic.make_batch(10 * i, 2, 20000, cfg, device=cuda) for i in range(3)]

    def run(pipelined):
        net = train.build_model(cfg, cuda, seed=0)
        opt = train.make_optimizer(net)
        losses, inds, sampling = [], [], None
        for i, b in enumerate(batches):
            nxt = batches[i + 1] if pipelined and i + 1 < len(batches) else None
            loss, end = train.train_step(net, opt, b, cfg, sampling=sampling, next_batch=nxt)
            sampling = end.get('next_sampling')
            losses.append(float(loss))
            inds.append((end['sa1_inds'].clone(), end['sa2_inds'].clone()))
        return losses, inds

    l0, i0 = run(False)
    l1, i1 = run(True)
    for (a1, a2), (b1, b2) in zip(i0, i1):
        assert torch.equal(a1, b1) and torch.equal(a2, b2)
    assert l0[0] == l1[0], (l0, l1)
    np.testing.assert_allclose(l1[1], l0[1], rtol=1e-2)


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_train_one_epoch_with_pyramids_in_flight(cuda, depth):
    """train.train_one_epoch keeps `depth` sampling pyramids in flight on alternating prefetch
    streams (2 by default for scenes of more than 65 536 points): every step must consume the
    pyramid of ITS batch -- indices bit-identical to the plain loop's on five distinct batches,
    first loss bit-identical -- and the default depth must follow the cloud size."""
    cfg = config.scannet_md40()
    batches = [synthetic.make_batch(10 * i, 2, 20000, cfg, device=cuda) for i in range(5)]
    assert train.pyramids_in_flight(batches[0]['point_clouds']) == 1
    assert train.pyramids_in_flight(torch.empty(1, 70000, 4, device=cuda)) == 2

    def plain():
        net = train.build_model(cfg, cuda, seed=0)
        opt = train.make_optimizer(net)
        out = []
        for b in batches:
            loss, end = train.train_step(net, opt, b, cfg)
            out.append((float(loss), end['sa1_inds'].clone(), end['sa2_inds'].clone(),
                        end['aggregated_vote_inds'].shape))
        return out

    def piped():
        net = train.build_model(cfg, cuda, seed=0)
        opt = train.make_optimizer(net)
        out = []
        train.train_one_epoch(net, opt, iter(batches), cfg, depth=depth,
                              on_step=lambda i, loss, end: out.append(
                                  (float(loss), end['sa1_inds'].clone(), end['sa2_inds'].clone(),
                                   end['aggregated_vote_inds'].shape)))
        return out

    a, b = plain(), piped()
    assert len(a) == len(b) == 5
    for (la, i1a, i2a, sa), (lb, i1b, i2b, sb) in zip(a, b):
        assert torch.equal(i1a, i1b) and torch.equal(i2a, i2b) and sa == sb
    assert a[0][0] == b[0][0], (a[0][0], b[0][0])
    np.testing.assert_allclose(b[1][0], a[1][0], rtol=1e-2)


@pytest.mark.parametrize("jitter", [False, True])
def test_back_to_reality_prefetch_gives_identical_training(cuda, jitter):
    """The same for the two-branch steps (train_step_br / train_step_br_jitter): both pyramids
    of the NEXT step under this step's backward; the indices both branches consume and the
    first loss are bit-identical to the plain loop's."""
    cfg = config.scannet_md40()
    jit = 0.1 if jitter else 0.0
    bs = [synthetic.make_batch(10 * i, 2, 12000, cfg, device=cuda, center_jitter=jit)
          for i in range(2)]
    bt = [synthetic.make_batch(500 + 10 * i, 2, 12000, cfg, device=cuda, center_jitter=jit)
          for i in range(2)]
    step = (lambda *a, **k: train.train_step_br_jitter(*a, epoch=30, **k)) if jitter \
        else train.train_step_br

    def run(pipelined):
        net = train.build_model(cfg, cuda, seed=0, domain_adaptation=True, center_refine=jitter)
        opt = train.make_optimizer(net)
        losses, inds, ss, st = [], [], None, None
        for i in range(2):
            more = pipelined and i + 1 < 2
            loss, es, et = step(net, opt, bs[i], bt[i], cfg, sampling_S=ss, sampling_T=st,
                                next_batch_S=bs[i + 1] if more else None,
                                next_batch_T=bt[i + 1] if more else None)
            ss, st = es.get('next_sampling'), et.get('next_sampling')
            losses.append(float(loss))
            inds.append((es['sa1_inds'].clone(), et['sa1_inds'].clone(), et['sa2_inds'].clone()))
        return losses, inds

    l0, i0 = run(False)
    l1, i1 = run(True)
    for a, b in zip(i0, i1):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert l0[0] == l1[0], (l0, l1)
    np.testing.assert_allclose(l1[1], l0[1], rtol=1e-2)


def test_graphed_pipelined_step_replays_the_eager_loop(cuda):
    """train.GraphedPipelinedStep (the software-pipelined step as one HIP graph, what bench.py
    times on one GPU) against the eager pipelined loop: same sampling indices consumed, first
    loss bit-identical, later losses as close as two eager runs are (float atomics in the
    nine-op backward of the unfused layers)."""
    cfg = config.scannet_md40()
    batches = [synthetic.make_batch(10 * i, 2, 20000, cfg, device=cuda) for i in range(2)]

    def eager():
        net = train.build_model(cfg, cuda, seed=0)
        opt = train.make_optimizer(net, capturable=True)
        losses = []
        sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
        for i in range(4):
            loss, end = train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                                         next_batch=batches[(i + 1) % 2])
            sampling = end['next_sampling']
            losses.append(float(loss))
        return losses

    def graphed():
        net = train.build_model(cfg, cuda, seed=0)
        opt = train.make_optimizer(net, capturable=True)
        state = {k: v.clone() for k, v in net.state_dict().items()}
        gs = train.GraphedPipelinedStep(net, opt, batches[0], batches[1], cfg, warmup=2)
        # capture + warm-up stepped the model: back to the initial weights and optimizer state
        net.load_state_dict(state)
        for st in opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        gs.prime(batches[0])
        losses = []
        for i in range(4):
            losses.append(float(gs(batches[i % 2], batches[(i + 1) % 2])))
        return losses

    from backtoreality_amd.pointnet2 import fused_backbone
    lg = graphed()
    # the eager loop on the path a capture takes (backbone layer by layer, chains below 2 048
    # rows on the stock ops): the replay is that loop
    with fused_backbone.layerwise():
        le = eager()
    np.testing.assert_allclose(lg[0], le[0], rtol=1e-6)
    # from the second step on even two eager runs drift apart by percents (run-dependent
    # summation order in the input-gradient scatter, then the vote FPS picks other proposals:
    # tools/diag_nondeterminism.py, tools/diag_step1b.py)
    np.testing.assert_allclose(lg[1], le[1], rtol=6e-2)
    assert all(np.isfinite(lg))
    # the default eager loop (whole-backbone library calls, every chain on the fused kernels)
    # computes the same function with other kernels: its first loss agrees to rounding; after
    # one Adam step (sign-like for noise-level gradients) the vote FPS may pick other proposals
    ld = eager()
    np.testing.assert_allclose(ld[0], lg[0], rtol=1e-5)
    np.testing.assert_allclose(ld[1], lg[1], rtol=6e-2)
