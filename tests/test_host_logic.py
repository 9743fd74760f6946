"""Host-side mirror of the reference interface: names, signatures, state-dict keys, error
behaviour (no GPU needed)."""
import inspect
import subprocess
import sys
import os

import numpy as np
import pytest
import torch

from backtoreality_amd.pointnet2 import pointnet2_modules as M
from backtoreality_amd.pointnet2 import pointnet2_utils as U
from backtoreality_amd.pointnet2 import pytorch_utils as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_public_names_of_pointnet2_utils():
    for name in ("furthest_point_sample", "gather_operation", "three_nn", "three_interpolate",
                 "grouping_operation", "ball_query", "QueryAndGroup", "GroupAll",
                 "FurthestPointSampling", "GatherOperation", "ThreeNN", "ThreeInterpolate",
                 "GroupingOperation", "BallQuery", "RandomDropout"):
        assert hasattr(U, name), name
    sig = inspect.signature(U.QueryAndGroup.__init__)
    assert list(sig.parameters)[1:] == ["radius", "nsample", "use_xyz", "ret_grouped_xyz",
                                        "normalize_xyz", "sample_uniformly", "ret_unique_cnt"]


def test_modules_are_keyword_only_like_the_reference():
    with pytest.raises(TypeError):
        M.PointnetSAModuleVotes([1, 8], 16, 0.2, 8)
    with pytest.raises(TypeError):
        M.PointnetFPModule([4, 4])
    for name in ("PointnetSAModuleVotes", "PointnetFPModule", "PointnetSAModuleCenters",
                 "PointnetSAModuleMSG", "PointnetSAModule", "PointnetSAModuleMSGVotes",
                 "PointnetLFPModuleMSG", "_PointnetSAModuleBase", "PointnetSAModuleOffset",
                 "ThreeNNInterpolate"):
        assert hasattr(M, name)


def test_state_dict_keys_match_reference_checkpoints():
    sa = M.PointnetSAModuleVotes(npoint=8, radius=0.2, nsample=4, mlp=[1, 8, 16], use_xyz=True)
    keys = list(sa.state_dict().keys())
    assert keys[:6] == ["mlp_module.layer0.conv.weight", "mlp_module.layer0.bn.bn.weight",
                        "mlp_module.layer0.bn.bn.bias", "mlp_module.layer0.bn.bn.running_mean",
                        "mlp_module.layer0.bn.bn.running_var",
                        "mlp_module.layer0.bn.bn.num_batches_tracked"]
    assert sa.state_dict()["mlp_module.layer0.conv.weight"].shape == (8, 4, 1, 1)  # +3 xyz
    fp = M.PointnetFPModule(mlp=[6, 4])
    assert "mlp.layer0.conv.weight" in fp.state_dict()
    # bias only without batch norm (pytorch_utils.py:87)
    assert "layer0.conv.bias" in P.SharedMLP([3, 4], bn=False).state_dict()
    assert "layer0.conv.bias" not in P.SharedMLP([3, 4], bn=True).state_dict()
    blk = P.Conv1d(3, 4, bn=True, preact=True)
    assert [n for n, _ in blk.named_children()] == ["bn", "activation", "conv"]
    assert blk.bn.bn.num_features == 3
    fc = P.FC(3, 4, bn=True)
    assert [n for n, _ in fc.named_children()] == ["fc", "bn", "activation"]


def test_bn_momentum_scheduler():
    net = P.SharedMLP([3, 4, 5], bn=True)
    sched = P.BNMomentumScheduler(net, lambda e: max(0.5 * 0.5 ** (e // 20), 0.001))
    assert net.layer0.bn.bn.momentum == 0.5
    sched.step(40)
    assert net.layer1.bn.bn.momentum == 0.125
    with pytest.raises(RuntimeError):
        P.BNMomentumScheduler(object(), lambda e: 0.1)


def test_product_ext_refuses_cpu_tensors():
    from backtoreality_amd.pointnet2 import _ext
    assert U._ext is _ext
    x = torch.zeros(1, 8, 3)
    for call in (lambda: _ext.furthest_point_sampling(x, 2),
                 lambda: _ext.ball_query(x, x, 0.1, 2),
                 lambda: _ext.three_nn(x, x),
                 lambda: _ext.gather_points(torch.zeros(1, 3, 8), torch.zeros(1, 2, dtype=torch.int32)),
                 lambda: U.furthest_point_sample(x, 2)):
        with pytest.raises(RuntimeError, match="CPU not supported"):
            call()
    with pytest.raises(RuntimeError, match="contiguous"):
        _ext.furthest_point_sampling(torch.zeros(1, 3, 8).transpose(1, 2), 2)


def test_query_and_group_and_group_all(oracle_ext):
    rng = np.random.default_rng(0)
    xyz = torch.from_numpy(rng.uniform(0.3, 2, (2, 64, 3)).astype(np.float32))
    feats = torch.from_numpy(rng.standard_normal((2, 5, 64)).astype(np.float32))
    inds = U.furthest_point_sample(xyz, 8)
    new_xyz = U.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    g = U.QueryAndGroup(0.7, 6, use_xyz=True, ret_grouped_xyz=True, normalize_xyz=True)
    nf, gx = g(xyz, new_xyz, feats)
    assert nf.shape == (2, 8, 8, 6) and gx.shape == (2, 3, 8, 6)
    idx = U.ball_query(0.7, 6, xyz, new_xyz).long()
    want = (torch.gather(xyz, 1, idx.view(2, -1, 1).expand(-1, -1, 3)).view(2, 8, 6, 3)
            - new_xyz.unsqueeze(2)) / 0.7
    torch.testing.assert_close(gx.permute(0, 2, 3, 1), want)
    assert U.QueryAndGroup(0.7, 6, use_xyz=False)(xyz, new_xyz, feats).shape == (2, 5, 8, 6)
    with pytest.raises(AssertionError):
        U.QueryAndGroup(0.7, 6, use_xyz=False)(xyz, new_xyz, None)
    ga = U.GroupAll(use_xyz=True, ret_grouped_xyz=True)
    nf, gx = ga(xyz, None, feats)
    assert nf.shape == (2, 8, 1, 64) and gx.shape == (2, 3, 1, 64)
    # sample_uniformly keeps every row a multiset of its unique neighbours (:336-345)
    torch.manual_seed(0)
    gu = U.QueryAndGroup(0.7, 6, use_xyz=True, sample_uniformly=True, ret_unique_cnt=True)
    nf, cnt = gu(xyz, new_xyz, feats)
    assert nf.shape == (2, 8, 8, 6) and cnt.shape == (2, 8) and cnt.min() >= 1


def test_sample_uniformly_equals_the_reference_loop(oracle_ext, monkeypatch):
    """QueryAndGroup(sample_uniformly=True): the one-pass tensor form against the reference's
    host double loop (pointnet2_utils.py:336-345) with the SAME uniform draws injected into
    both: identical rows and unique counts."""
    rng = np.random.default_rng(7)
    xyz = torch.from_numpy(rng.uniform(0, 2, (2, 300, 3)).astype(np.float32))
    new_xyz = torch.cat([xyz[:, :30], xyz[:, :2] + 50.0], 1).contiguous()   # two empty balls
    S = 12
    idx0 = U.ball_query(0.35, S, xyz, new_xyz)
    u = torch.from_numpy(rng.uniform(0, 1, idx0.shape).astype(np.float32))
    monkeypatch.setattr(U.QueryAndGroup, "_uniform_draws", staticmethod(lambda shape, device: u))
    grouper = U.QueryAndGroup(0.35, S, use_xyz=True, sample_uniformly=True, ret_unique_cnt=True)
    got = idx0.clone()
    cnt = grouper._resample_uniformly(got)
    want, want_cnt = idx0.clone(), torch.zeros(idx0.shape[:2])
    for b in range(idx0.shape[0]):                       # the reference's loop
        for r in range(idx0.shape[1]):
            uniq = torch.unique(idx0[b, r, :])
            k = uniq.shape[0]
            want_cnt[b, r] = k
            pick = torch.clamp((u[b, r, k:] * k).long(), max=k - 1)   # its randint(0, k, S - k)
            want[b, r, :] = torch.cat((uniq, uniq[pick]))
    assert torch.equal(got, want) and torch.equal(cnt, want_cnt)
    assert cnt.min() == 1 and cnt.max() > 3


def test_sa_variants_run(oracle_ext):
    rng = np.random.default_rng(1)
    xyz = torch.from_numpy(rng.uniform(0.3, 2, (2, 96, 3)).astype(np.float32))
    feats = torch.from_numpy(rng.standard_normal((2, 4, 96)).astype(np.float32)).requires_grad_()
    torch.manual_seed(0)
    sa = M.PointnetSAModuleVotes(npoint=16, radius=0.6, nsample=8, mlp=[4, 8, 8], use_xyz=True,
                                 normalize_xyz=True)
    new_xyz, nf, inds = sa(xyz, feats)
    assert new_xyz.shape == (2, 16, 3) and nf.shape == (2, 8, 16) and inds.dtype == torch.int32
    nf.sum().backward()
    assert feats.grad.abs().sum() > 0
    # given indices are used as they are (:234-237)
    given = torch.arange(16, dtype=torch.int32).repeat(2, 1)
    nx2, _, i2 = sa(xyz, feats, given)
    assert torch.equal(i2, given) and torch.equal(nx2, xyz[:, :16])
    for pooling in ("avg", "rbf"):
        m = M.PointnetSAModuleVotes(npoint=16, radius=0.6, nsample=8, mlp=[4, 8], pooling=pooling)
        assert m(xyz, feats)[1].shape == (2, 8, 16)
    centers = M.PointnetSAModuleCenters(npoint=5, radius=0.8, nsample=16, mlp=[4, 8])
    assert centers(xyz, feats, xyz[:, :5].contiguous()).shape == (2, 8, 5)
    # GroupFree3D's variants: same computation as ...Centers, and the bare interpolation
    offset = M.PointnetSAModuleOffset(npoint=5, radius=0.8, nsample=16, mlp=[4, 8])
    offset.load_state_dict(centers.state_dict())
    assert torch.equal(offset(xyz, feats, xyz[:, :5].contiguous()),
                       centers(xyz, feats, xyz[:, :5].contiguous()))
    fp = M.PointnetFPModule(mlp=[4, 4])
    up = M.ThreeNNInterpolate(feats[:, :, :16].contiguous(), xyz[:, :16].contiguous(), xyz)
    assert up.shape == (2, 4, xyz.shape[1])
    assert torch.allclose(up[:, :, :16], feats[:, :, :16], atol=1e-4)  # a known point: itself
    del fp
    msg = M.PointnetSAModuleMSG(npoint=16, radii=[0.4, 0.8], nsamples=[4, 8],
                                mlps=[[4, 8], [4, 16]])
    nx, nf = msg(xyz, feats)
    assert nf.shape == (2, 24, 16)
    msgv = M.PointnetSAModuleMSGVotes(npoint=16, radii=[0.4], nsamples=[4], mlps=[[4, 8]])
    assert msgv(xyz, feats)[2].shape == (2, 16)
    lfp = M.PointnetLFPModuleMSG(mlps=[[4, 8]], radii=[0.6], nsamples=[8], post_mlp=[8 + 3, 6])
    out = lfp(xyz[:, :16].contiguous(), xyz, torch.zeros(2, 3, 16), feats)
    assert out.shape == (2, 6, 16)


def test_drop_in_top_level_imports_like_reference_scripts():
    """train_Votenet_FSB.py:35-38 does sys.path.append(ROOT/pointnet2) and imports
    pointnet2_modules / pointnet2_utils / pytorch_utils as top-level modules."""
    code = (
        "import sys; sys.path.append(%r)\n"
        "import pointnet2_modules, pointnet2_utils, pytorch_utils\n"
        "from pointnet2_modules import PointnetSAModuleVotes, PointnetFPModule\n"
        "import pointnet2._ext as e\n"
        "assert pointnet2_utils._ext is e and callable(e.ball_query)\n"
        "print('ok')\n" % os.path.join(ROOT, "backtoreality_amd", "pointnet2"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_checkpoint_round_trip_in_the_reference_format(tmp_path, oracle_ext):
    """checkpoint.tar keys as written by train_Votenet_FSB.py:310-318; a resumed model takes
    the same next step as the original."""
    import torch
    from backtoreality_amd.votenet import config, synthetic, train
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(0, 1, 1024, cfg)
    net = train.build_model(cfg, torch.device("cpu"), seed=0)
    opt = train.make_optimizer(net)
    train.train_step(net, opt, batch, cfg)
    path = str(tmp_path / "checkpoint.tar")
    train.save_checkpoint(path, net, opt, epoch=4, loss=1.5)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {'epoch', 'optimizer_state_dict', 'loss', 'model_state_dict'}
    assert raw['epoch'] == 5 and raw['loss'] == 1.5
    assert list(raw['model_state_dict']) == list(net.state_dict())

    net2 = train.build_model(cfg, torch.device("cpu"), seed=123)   # different init
    opt2 = train.make_optimizer(net2)
    assert train.load_checkpoint(path, net2, opt2) == 5
    l1, _ = train.train_step(net, opt, batch, cfg)
    l2, _ = train.train_step(net2, opt2, batch, cfg)
    assert float(l1) == float(l2)
    for (n, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), n


def test_conv_autotune_switch():
    """train.enable_conv_autotune = the reference scripts' cudnn.benchmark line
    (train_GF_FSB.py:454-455); nothing is shipped beside it any more (round 6)."""
    import torch
    from backtoreality_amd.votenet import train
    old = torch.backends.cudnn.benchmark
    try:
        torch.backends.cudnn.benchmark = False
        train.enable_conv_autotune()
        assert torch.backends.cudnn.benchmark
    finally:
        torch.backends.cudnn.benchmark = old


def test_epoch_schedules_of_the_training_scripts():
    from backtoreality_amd.groupfree import train as gf_train
    from backtoreality_amd.votenet import train
    assert train.get_current_lr(0) == 0.001 and train.get_current_lr(79) == 0.001
    assert abs(train.get_current_lr(80) - 1e-4) < 1e-12
    assert abs(train.get_current_lr(160) - 1e-6) < 1e-15
    lin = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    opt = torch.optim.Adam(lin.parameters(), lr=1.0)
    assert abs(train.adjust_learning_rate(opt, 120) - 1e-5) < 1e-15
    assert abs(opt.param_groups[0]['lr'] - 1e-5) < 1e-15
    sch = train.make_bn_momentum_scheduler(lin)
    assert lin[1].momentum == 0.5
    for _ in range(20):      # the scripts call step() at the START of epochs 0..19
        sch.step()
    assert lin[1].momentum == 0.5
    sch.step()               # epoch 20
    assert lin[1].momentum == 0.25
    sch.step(400)
    assert lin[1].momentum == 0.001
    opt = torch.optim.AdamW([{"params": [lin[0].weight]}, {"params": [lin[0].bias], "lr": 0.0004}],
                            lr=0.004)
    s = gf_train.get_scheduler(opt, n_iter_per_epoch=2, lr_decay_epochs=(3, 5), warmup_epoch=-1)
    lrs = []
    for _ in range(13):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        s.step()
    # milestones at (3 + 1) * 2 = 8 and (5 + 1) * 2 = 12 iterations
    assert abs(lrs[7] - 0.004) < 1e-12 and abs(lrs[8] - 0.0004) < 1e-12
    assert abs(lrs[12] - 0.00004) < 1e-12
    assert abs(opt.param_groups[1]['lr'] - 0.000004) < 1e-12


def test_decoded_end_points_are_lazy_but_look_complete():
    """proposal_module.DecodedEndPoints: decode_scores runs on first use, and every way of
    looking at the dict sees the decoded keys."""
    from backtoreality_amd.votenet.proposal_module import DECODED_KEYS, DecodedEndPoints
    calls = []

    def decode(ep):
        calls.append(1)
        for k in DECODED_KEYS:
            ep[k] = k.upper()
    ep = DecodedEndPoints({'seed_xyz': 1}, decode)
    ep['loss'] = 2.0                       # plain reads / writes do not decode
    assert ep['seed_xyz'] == 1 and 'loss' in ep and not calls
    assert 'center' in ep and 'nonsense' not in ep and not calls
    assert ep['center'] == 'CENTER' and calls == [1]
    assert ep.get('pred_size') == 'PRED_SIZE' and ep.get('nonsense', 7) == 7
    for make in (lambda e: list(e.keys()), lambda e: dict(e), lambda e: len(e),
                 lambda e: list(e.items()), lambda e: e.copy(), lambda e: [k for k in e]):
        e2 = DecodedEndPoints({'a': 1}, decode)
        got = make(e2)
        assert dict.__contains__(e2, 'center'), make
        if isinstance(got, (list, dict)):
            assert len(got) == 1 + len(DECODED_KEYS)
    rebuilt = type(ep)((k, v) for k, v in ep.items())     # what DataParallel's gather does
    assert rebuilt['center'] == 'CENTER' and len(rebuilt) == len(ep)
    with __import__('pytest').raises(KeyError):
        DecodedEndPoints({}, decode)['nonsense']
    assert calls.count(1) == 7


@pytest.mark.gpu
def test_fast_adam_equals_torch_adam(cuda, monkeypatch):
    """train.FastAdam = torch.optim.Adam(fused=True) minus the per-step Python bookkeeping:
    identical parameters / state after several steps (one parameter without a gradient, a
    learning-rate change, a state_dict round trip).  On torch's own update kernels
    (BTR_ADAM_KERNEL=0) the equality is bit-exact; the library's one-launch kernel is compared
    in tests/test_optimizer_gpu.py (a few ulps)."""
    import copy
    from backtoreality_amd.votenet import train
    monkeypatch.setenv("BTR_ADAM_KERNEL", "0")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4),
                              torch.nn.Linear(4, 4)).to(cuda)      # [3] never used: no gradient
    ref = copy.deepcopy(net)
    fast = train.make_optimizer(net, lr=1e-3)
    assert isinstance(fast, train.FastAdam)
    slow = torch.optim.Adam(ref.parameters(), lr=1e-3, fused=True)
    x = torch.randn(8, 16, device=cuda)
    for i in range(6):
        if i == 3:
            train.adjust_learning_rate(fast, 90)
            train.adjust_learning_rate(slow, 90)
            sd = copy.deepcopy(fast.state_dict())
            assert all('_btr_fast' not in g for g in sd['param_groups'])
            fast.load_state_dict(sd)
        for m, o in ((net, fast), (ref, slow)):
            o.zero_grad(set_to_none=True)
            m[2](m[1](m[0](x))).square().sum().backward()
            o.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.equal(a, b)
    for a, b in zip(net.parameters(), ref.parameters()):
        if a in fast.state and fast.state[a]:
            assert torch.equal(fast.state[a]['exp_avg_sq'], slow.state[b]['exp_avg_sq'])
            assert float(fast.state[a]['step']) == float(slow.state[b]['step']) == 6


def test_fast_adamw_with_folded_clipping_equals_clip_then_adamw():
    """FastAdamW.step(clip_norm=c) == clip_grad_norm_(c) followed by torch.optim.AdamW.step()
    (train_GF_FSB.py:316-319), on the stock fallback here (CPU: not fused)."""
    import copy
    import torch
    from backtoreality_amd.votenet.train import FastAdamW
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    net_b = copy.deepcopy(net_a)
    opt_a = FastAdamW(net_a.parameters(), lr=0.01, weight_decay=0.05)
    opt_b = torch.optim.AdamW(net_b.parameters(), lr=0.01, weight_decay=0.05)
    x = torch.randn(16, 5)
    for _ in range(3):
        for net, opt in ((net_a, opt_a), (net_b, opt_b)):
            opt.zero_grad()
            (net(x) ** 2).sum().backward()
        total = opt_a.step(clip_norm=0.1)
        ref = torch.nn.utils.clip_grad_norm_(net_b.parameters(), 0.1)
        opt_b.step()
        assert torch.allclose(total, ref)
    for a, b in zip(net_a.parameters(), net_b.parameters()):
        assert torch.equal(a, b)


def test_decoder_stack_host_logic():
    """The host side of the GroupFree3D decoder stack's slots (groupfree/fused_stack.py), no GPU:
    the output block's layout, the lease of a slot, the state signature that guards the cached
    coverage checks, the eval constants' cache key."""
    from backtoreality_amd import groupfree
    from backtoreality_amd.groupfree import fused_stack
    from backtoreality_amd.votenet import config
    dims = (6, 4, 256, 288, 97, 100, 1, 18, True)
    n = fused_stack._OutBlock.floats(*dims)
    flat = torch.arange(n, dtype=torch.float32)
    ob = fused_stack._OutBlock(flat, *dims)
    L, B, Pq, E, C, Cp, nh, ns, _ = dims
    assert ob.cls.shape == (L, B * Pq, Cp) and ob.last.shape == (B, E, Pq)
    assert len(ob.outs) == L and ob.outs[3][0].shape == (B, C, Pq)
    pieces = [ob.cls, ob.qpos, ob.qpos_t, ob.last, ob.last_cl] + [t for o in ob.outs for t in o[:5]]
    spans = sorted((int(t.reshape(-1)[0]), int(t.reshape(-1)[0]) + t.numel()) for t in pieces)
    assert all(a % 64 == 0 for a, _ in spans)                      # 256-byte bounds
    assert all(b <= c for (_, b), (c, _) in zip(spans, spans[1:]))   # disjoint
    assert spans[-1][1] <= n
    assert ob.outs[2][5].data_ptr() == ob.qpos[2].data_ptr()         # query_pos are views of qpos

    class S(object):
        busy = False
    s = S()
    lease = fused_stack._Lease(s)
    assert s.busy
    lease.release()
    lease.release()
    assert not s.busy and lease.slot is None
    lease = fused_stack._Lease(s)
    del lease                                                          # freed with its holder
    assert not s.busy

    cfg = config.scannet_md40()
    det = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                      cfg.mean_size_arr, input_feature_dim=0, num_proposal=64,
                                      dropout=0.1)
    a = fused_stack._state_signature(det)
    assert a == fused_stack._state_signature(det)
    det.decoder[1].dropout2.p = 0.2
    b = fused_stack._state_signature(det)
    assert b != a
    det.decoder[1].dropout2.p = 0.1
    det.prediction_heads[0].bn2.eval()
    assert fused_stack._state_signature(det) != a
    det.prediction_heads[0].bn2.train()
    assert fused_stack._state_signature(det) == a
    conv = det.prediction_heads[3].sem_cls_scores_head
    conv.weight = torch.nn.Parameter(conv.weight.detach().clone())
    assert fused_stack._state_signature(det) != a
    subs, rows = fused_stack._head_subs(det)
    assert len(subs) == 14 * det.num_decoder_layers and subs[-2] is \
        det.prediction_heads[-1].sem_cls_scores_head.weight
    assert sum(rows[0]) == det.prediction_heads[0].cat_standin(torch.device("cpu")).out_channels
