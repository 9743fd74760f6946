"""GroupFree3D (backtoreality_amd/groupfree: detector, decoder, loss -- SURVEY 8(f) #2) against
tests/golden/groupfree_step.npz, generated from the REFERENCE's GroupFreeDetector + get_loss
(detection/GroupFree3D/models/) by tests/golden/make_golden.py --groupfree.  The CPU run goes
through the oracle `_ext`; the GPU twin through the HIP kernels."""
import os

import numpy as np
import pytest
import torch

from backtoreality_amd import groupfree
from backtoreality_amd.votenet import config, synthetic

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "groupfree_step.npz")
LOSS_ARGS = dict(num_decoder_layers=6, query_points_generator_loss_coef=0.8, obj_loss_coef=0.1,
                 box_loss_coef=1, sem_cls_loss_coef=0.1, query_points_obj_topk=4)
PREFIXES = ['proposal_', 'last_'] + ['%dhead_' % i for i in range(5)]


def run(device):
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(0, 2, 4096, cfg, use_height=False, device=device)
    torch.manual_seed(0)
    net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                      cfg.mean_size_arr, input_feature_dim=0, num_proposal=256,
                                      dropout=0.0).to(device)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    names = sorted(sd)
    sig = (names, np.array([float(sd[k].double().sum()) for k in names]),
           [str(tuple(sd[k].shape)) for k in names])
    end_points = net({'point_clouds': batch['point_clouds']})
    end_points.update(batch)
    loss, end_points = groupfree.get_loss(end_points, cfg, **LOSS_ARGS)
    loss.backward()
    return net, sig, end_points


def _sub(t, sample, sums, rtol, stride):
    a = t.detach().cpu().numpy().astype(np.float32).ravel()
    scale = np.abs(sample).max() + 1e-12
    np.testing.assert_allclose(a[::stride], sample, rtol=rtol, atol=rtol * scale)
    np.testing.assert_allclose(np.abs(a.astype(np.float64)).sum(), sums[1], rtol=rtol)


def check(res, rtol, grad_rtol):
    g = np.load(GOLD)
    net, (names, sums, shapes), ep = res
    # reference checkpoints load: identical state-dict keys (the decoder layers' aliases of
    # the position embeddings included), shapes and seeded initial values
    assert names == list(g['state_names']) and shapes == list(g['state_shapes'])
    np.testing.assert_allclose(sums, g['state_sums'], rtol=1e-6, atol=1e-6)
    for k in ('sa1_inds', 'seed_inds'):
        np.testing.assert_array_equal(ep[k].cpu().numpy(), g[k])
    _sub(ep['fp2_features'], g['fp2_features_sample'], g['fp2_features_sums'], rtol, 37)
    lg = g['seeds_obj_cls_logits']
    np.testing.assert_allclose(ep['seeds_obj_cls_logits'].detach().cpu().numpy(), lg, rtol=rtol,
                               atol=rtol * np.abs(lg).max())
    # k-closest-point sampling: the same top-k seeds in the same order (a swap is only
    # possible between scores closer than rounding)
    inds = ep['query_points_sample_inds'].cpu().numpy()
    assert inds.dtype == np.int32
    assert np.array_equal(np.sort(inds, 1), np.sort(g['query_points_sample_inds'], 1))
    same_order = np.array_equal(inds, g['query_points_sample_inds'])
    for p in PREFIXES:
        np.testing.assert_array_equal(ep[p + 'objectness_label'].cpu().numpy()[
            np.arange(2)[:, None], np.argsort(inds, 1)],
            g[p + 'objectness_label'][np.arange(2)[:, None],
                                      np.argsort(g['query_points_sample_inds'], 1)])
        if same_order:
            for k in ('object_assignment', 'objectness_label'):
                np.testing.assert_array_equal(ep[p + k].cpu().numpy(), g[p + k])
            np.testing.assert_allclose(ep[p + 'objectness_mask'].cpu().numpy(),
                                       g[p + 'objectness_mask'], rtol=1e-6)
            for k in ('center', 'objectness_scores', 'sem_cls_scores', 'pred_size',
                      'heading_residuals', 'size_scores'):
                want = g[p + k]
                np.testing.assert_allclose(ep[p + k].detach().cpu().numpy(), want,
                                           rtol=10 * rtol, atol=10 * rtol * np.abs(want).max())
            _sub(ep[p + 'size_residuals'], g[p + 'size_residuals_sample'],
                 g[p + 'size_residuals_sums'], 10 * rtol, 37)
        for k in ('objectness_loss', 'center_loss', 'heading_cls_loss', 'heading_reg_loss',
                  'size_cls_loss', 'size_reg_loss', 'box_loss', 'sem_cls_loss', 'pos_ratio',
                  'neg_ratio'):
            a, b = float(ep[p + k]), float(g[p + k])
            assert abs(a - b) <= 10 * rtol * max(1.0, abs(b)), (p + k, a, b)
    for k in ('loss', 'query_points_generation_loss', 'sum_heads_objectness_loss',
              'sum_heads_box_loss', 'sum_heads_sem_cls_loss', 'points_hard_topk4_pos_ratio',
              'points_hard_topk4_neg_ratio'):
        a, b = float(ep[k]), float(g[k])
        assert abs(a - b) <= 10 * rtol * max(1.0, abs(b)), (k, a, b)
    grads = {'grad_sa1_w0': net.backbone_net.sa1.mlp_module.layer0.conv.weight.grad,
             'grad_obj_cls_conv3': net.points_obj_cls.conv3.weight.grad,
             'grad_proposal_conv1': net.proposal_head.conv1.weight.grad,
             'grad_dec0_self_in_proj': net.decoder[0].self_attn.in_proj_weight.grad,
             'grad_dec5_linear2': net.decoder[5].linear2.weight.grad,
             'grad_dec3_cross_pos0':
                 net.decoder_cross_posembeds[3].position_embedding_head[0].weight.grad,
             'grad_key_proj': net.decoder_key_proj.weight.grad,
             'grad_head2_center': net.prediction_heads[2].center_residual_head.weight.grad}
    for k, t in grads.items():
        a = t.detach().cpu().numpy().astype(np.float32).ravel()[::53]
        want = g[k + '_sample']
        rel = np.linalg.norm(a - want) / (np.linalg.norm(want) + 1e-30)
        assert rel <= grad_rtol, (k, rel)


def test_groupfree_step_matches_reference_cpu(oracle_ext, monkeypatch):
    monkeypatch.setenv("BTR_FUSED_SA", "0")
    check(run(torch.device("cpu")), rtol=1e-4, grad_rtol=1e-3)


@pytest.mark.gpu
def test_groupfree_step_matches_reference_gpu(cuda):
    check(run(cuda), rtol=1e-4, grad_rtol=2e-2)


def test_groupfree_variants_build_and_name_their_heads():
    cfg = config.scannet_md40()
    net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                      cfg.mean_size_arr, num_proposal=64, sampling='fps',
                                      num_decoder_layers=2, self_position_embedding='loc_learned')
    assert not hasattr(net, 'points_obj_cls') and len(net.decoder) == 2
    assert net.decoder[0].self_posembed is net.decoder_self_posembeds[0]
    assert net.decoder_self_posembeds[0].position_embedding_head[0].in_channels == 6
    bare = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                       cfg.mean_size_arr, num_decoder_layers=0)
    assert not hasattr(bare, 'decoder')
    from backtoreality_amd.groupfree.loss_helper import head_prefixes
    assert head_prefixes(0) == ['proposal_']
    assert head_prefixes(3) == ['proposal_', 'last_', '0head_', '1head_']
    with pytest.raises(NotImplementedError):
        groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                    cfg.mean_size_arr, sampling='random')


def test_groupfree_backbone_width_and_depth():
    """`width` / `depth` of GroupFree3D's backbone (models/backbone_module.py:33-75; train_GF_FSB.py
    --width): mlp = [in] + [h * width] * depth + [out * width] per level, FP modules
    [512 w, 256 w, 256 w] and [512 w, 256 w, 288]; the detector hands `width` on
    (detector.py:61)."""
    from backtoreality_amd.votenet.backbone_module import Pointnet2Backbone
    bb = Pointnet2Backbone(input_feature_dim=0, fp2_out=288, width=2, depth=3)
    shapes = {n: tuple(p.shape) for n, p in bb.state_dict().items() if n.endswith("conv.weight")}
    want = {"sa1": [3, 128, 128, 128, 256], "sa2": [256 + 3, 256, 256, 256, 512],
            "sa3": [512 + 3, 256, 256, 256, 512], "sa4": [512 + 3, 256, 256, 256, 512]}
    for name, chain in want.items():
        for i in range(len(chain) - 1):
            assert shapes["%s.mlp_module.layer%d.conv.weight" % (name, i)] == (
                chain[i + 1], chain[i], 1, 1), (name, i)
    assert shapes["fp1.mlp.layer0.conv.weight"] == (512, 1024, 1, 1)
    assert shapes["fp1.mlp.layer1.conv.weight"] == (512, 512, 1, 1)
    assert shapes["fp2.mlp.layer0.conv.weight"] == (512, 1024, 1, 1)
    assert shapes["fp2.mlp.layer1.conv.weight"] == (288, 512, 1, 1)
    cfg = config.scannet_md40()
    net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                      cfg.mean_size_arr, width=2, num_decoder_layers=1)
    assert net.backbone_net.width == 2 and net.backbone_net.depth == 2
    assert net.backbone_net.sa2.mlp_module.layer2.conv.weight.shape == (512, 256, 1, 1)
    # width 1 / depth 2 is the layout every golden fixture was generated with
    base = Pointnet2Backbone(input_feature_dim=0, fp2_out=288)
    assert base.sa1.mlp_module.layer2.conv.weight.shape == (128, 64, 1, 1)


@pytest.mark.gpu
def test_groupfree_wide_backbone_fused_equals_unfused(cuda, monkeypatch):
    """width = 2 (train_GF_FSB.py --width 2): the 256 / 512-column layers leave the one-pass
    backward's range (<= 256 columns) and the first-layer recompute sits behind a 128-column
    layer: fused path against the nine-op + torch composition, features 1e-4, gradients 1e-2."""
    from backtoreality_amd.votenet.backbone_module import Pointnet2Backbone
    B = 2
    pc = torch.from_numpy(np.stack([synthetic.make_scene(90 + i, 12000, use_height=False)[
        'point_clouds'] for i in range(B)], 0)).to(cuda)

    def run(fused):
        monkeypatch.setenv("BTR_FUSED_SA", "1" if fused else "0")
        torch.manual_seed(0)
        net = Pointnet2Backbone(input_feature_dim=0, fp2_out=288, width=2).to(cuda)
        end = net(pc)
        end['fp2_features'].square().mean().backward()
        return end, {n: p.grad.detach().clone() for n, p in net.named_parameters()}

    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    end_f, g_f = run(True)
    end_u, g_u = run(False)
    assert end_f['fp2_features'].shape == (B, 288, 1024)
    assert torch.equal(end_f['sa1_inds'], end_u['sa1_inds'])
    for k in ('sa1_features', 'sa2_features', 'sa4_features', 'fp2_features'):
        assert rel(end_f[k], end_u[k]) < 1e-4, k
    for n in g_u:
        d = float((g_f[n] - g_u[n]).norm() / (g_u[n].norm() + 1e-20))
        assert d < 1e-2, (n, d)


@pytest.mark.gpu
def test_groupfree_train_steps(cuda):
    """Training-mode steps with the script defaults (dropout 0.1, clip 0.1, AdamW groups) on
    the fused SA path: finite losses, every parameter receives a gradient, the decoder group
    runs at its own learning rate."""
    from backtoreality_amd.groupfree import train as gf_train
    cfg = config.scannet_md40()
    net = gf_train.build_model(cfg, cuda)
    opt = gf_train.make_optimizer(net)
    assert [g['lr'] for g in opt.param_groups] == [0.004, 0.0004]
    n_dec = sum(p.numel() for n, p in net.named_parameters() if "decoder" in n)
    assert sum(p.numel() for p in opt.param_groups[1]['params']) == n_dec > 10e6
    batch = synthetic.make_batch(0, 2, 8192, cfg, use_height=False, device=cuda)
    losses = []
    for _ in range(3):
        loss, end = gf_train.train_step(net, opt, batch, cfg)
        losses.append(float(loss))
    assert all(np.isfinite(losses)), losses
    missing = [n for n, p in net.named_parameters() if p.grad is None]
    assert not missing, missing
    assert end['last_center'].shape == (2, 256, 3) and end['seeds_obj_cls_logits'].shape == (2, 1, 1024)


@pytest.mark.gpu
def test_groupfree_eval_through_ap_helper(cuda):
    """The last decoder head's boxes through the evaluation path
    (detection/GroupFree3D/models/ap_helper.py: `prefix`, one sigmoid objectness logit)."""
    from backtoreality_amd.groupfree import train as gf_train
    from backtoreality_amd.votenet import ap_helper, train
    cfg = config.scannet_md40()
    net = gf_train.build_model(cfg, cuda).eval()
    batch = synthetic.make_batch(0, 2, 8192, cfg, use_height=False, device=cuda)
    with torch.no_grad():
        end = net({'point_clouds': batch['point_clouds']})
    end.update(batch)
    cd = dict(train.EVAL_CONFIG_DICT, dataset_config=cfg, conf_thresh=0.0)
    pred = ap_helper.parse_predictions(end, cd, prefix='last_')
    # the same boxes through the un-prefixed two-logit form VoteNet's head emits
    for k in ('center', 'heading_scores', 'heading_residuals', 'size_scores', 'size_residuals',
              'sem_cls_scores'):
        end[k] = end['last_' + k]
    obj = end['last_objectness_scores']
    end['objectness_scores'] = torch.cat([torch.zeros_like(obj), obj], -1)
    pred2 = ap_helper.parse_predictions(end, cd)
    assert [len(p) for p in pred] == [len(p) for p in pred2]
    assert np.allclose([s for p in pred for _, _, s in p], [s for p in pred2 for _, _, s in p],
                       rtol=1e-5, atol=1e-7)
    gt = ap_helper.parse_groundtruths(end, cd)
    calc = ap_helper.APCalculator(0.25)
    calc.step(pred, gt)
    assert 'mAP' in calc.compute_metrics()


@pytest.mark.gpu
def test_groupfree_graphed_step_trains_like_the_eager_step(cuda):
    """The captured HIP graph replays the same step: with dropout off, the parameters after
    one eager warm-up step and two replays equal the parameters after three eager steps (up to atomics order)."""
    from backtoreality_amd.groupfree import train as gf_train
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(0, 2, 8192, cfg, use_height=False, device=cuda)
    nets = []
    for graphed in (False, True):
        net = gf_train.build_model(cfg, cuda, dropout=0.0)
        opt = gf_train.make_optimizer(net, capturable=graphed)
        if graphed:   # one eager step inside the constructor (warm-up), then two replays
            step = gf_train.GraphedTrainStep(net, opt, batch, cfg, warmup=1)
            for _ in range(2):
                loss, _ = step(batch)
        else:
            for _ in range(3):
                loss, _ = gf_train.train_step(net, opt, batch, cfg)
        nets.append((net, float(loss)))
    (ne, le), (ng, lg) = nets
    # (float atomics order differs between runs and top-k query sampling amplifies it: two
    # eager runs differ by as much)
    assert abs(le - lg) <= 3e-2 * abs(le), (le, lg)
    # Adam moves a parameter by ~lr per step whatever the gradient's size, so parameters whose
    # gradient is rounding noise differ by up to steps * lr; compare in aggregate
    num = sum(float((a - b).double().pow(2).sum())
              for a, b in zip(ne.parameters(), ng.parameters()))
    den = sum(float(a.double().pow(2).sum()) for a in ne.parameters())
    # (two eager runs of one build: 0.0024; eager against replay 0.0098-0.0105; the same eager
    # step on two GEMM kernels whose BatchNorm sums are added in another order 0.013 --
    # tools/diag_gf_eager_vs_graph.py; a replay that did nothing: 0.15)
    assert (num / den) ** 0.5 <= 2e-2, (num / den) ** 0.5
    for a, b in zip(ne.parameters(), ng.parameters()):
        # (each run moves a parameter by at most ~lr per step; opposite signs on a noise-level
        # gradient put two runs 2 * lr apart per step)
        assert float((a - b).abs().max()) <= 2 * 3 * 0.004 + 1e-6


@pytest.mark.gpu
def test_groupfree_pipelined_steps_equal_the_sequential_ones(cuda):
    """Software pipelining of the GroupFree3D step (next batch's sampling pyramid under this
    step's backward): the eager pipelined loop consumes exactly the indices the sequential
    loop computes (first loss bit-identical), and the one-graph form
    (train.GraphedPipelinedStep, what bench.py --workload gf replays) follows it."""
    from backtoreality_amd.groupfree import train as gf_train
    cfg = config.scannet_md40()
    batches = [synthetic.make_batch(10 * i, 2, 8192, cfg, use_height=False, device=cuda)
               for i in range(2)]

    def run(mode):
        net = gf_train.build_model(cfg, cuda, dropout=0.0)
        opt = gf_train.make_optimizer(net, capturable=True)
        losses = []
        if mode == "sequential":
            for i in range(3):
                losses.append(float(gf_train.train_step(net, opt, batches[i % 2], cfg)[0]))
        elif mode == "pipelined":
            sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
            for i in range(3):
                loss, end = gf_train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                                                next_batch=batches[(i + 1) % 2])
                sampling = end['next_sampling']
                losses.append(float(loss))
        else:
            state = {k: v.clone() for k, v in net.state_dict().items()}
            gs = gf_train.GraphedPipelinedStep(net, opt, batches[0], batches[1], cfg, warmup=1)
            net.load_state_dict(state)   # capture + warm-up stepped the model: start over
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            gs.prime(batches[0])
            for i in range(3):
                losses.append(float(gs(batches[i % 2], batches[(i + 1) % 2])[0]))
        return losses

    seq, pipe, graph = run("sequential"), run("pipelined"), run("graphed")
    assert seq[0] == pipe[0], (seq, pipe)
    # (the capture takes the layer-by-layer backbone and leaves small chains to stock ops: same
    # function, other kernels)
    np.testing.assert_allclose(graph[0], seq[0], rtol=1e-5)
    # later steps: float atomics order + top-k query sampling (two eager runs differ as much)
    np.testing.assert_allclose(pipe[1:], seq[1:], rtol=3e-2)
    # (the captured step runs other kernels for the small chains: its third loss was seen 3.3 %
    # from the sequential one in 1 of 8 runs -- the chaotic proposal picks again, not an error
    # that grows with the step count: tests/test_configs_gpu.py compares parameters)
    np.testing.assert_allclose(graph[1:2], seq[1:2], rtol=3e-2)
    np.testing.assert_allclose(graph[2:], seq[2:], rtol=8e-2)


# ------------------------------------------------------------ Back-to-Reality step (8f #2)
GOLD_BR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                       "groupfree_br_step.npz")


def run_br(device):
    cfg = config.scannet_md40()
    batch_S = synthetic.make_batch(0, 2, 4096, cfg, use_height=False, device=device)
    batch_T = synthetic.make_batch(100, 2, 4096, cfg, use_height=False, device=device)
    torch.manual_seed(0)
    net = groupfree.GroupFreeDetector_DA(cfg.num_class, cfg.num_heading_bin,
                                         cfg.num_size_cluster, cfg.mean_size_arr,
                                         input_feature_dim=0, num_proposal=256,
                                         dropout=0.0).to(device)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    names = sorted(sd)
    sig = (names, np.array([float(sd[k].double().sum()) for k in names]),
           [str(tuple(sd[k].shape)) for k in names])
    eS = net({'point_clouds': batch_S['point_clouds']})
    eT = net({'point_clouds': batch_T['point_clouds']})
    eS.update(batch_S)
    eT.update(batch_T)
    loss, eS, eT = groupfree.get_loss_DA(eS, eT, cfg, **LOSS_ARGS)
    loss.backward()
    return net, sig, loss, eS, eT


def check_br(res, rtol, grad_rtol):
    g = np.load(GOLD_BR)
    net, (names, sums, shapes), loss, eS, eT = res
    assert names == list(g['state_names']) and shapes == list(g['state_shapes'])
    np.testing.assert_allclose(sums, g['state_sums'], rtol=1e-6, atol=1e-6)
    close = lambda a, b, k: abs(a - b) <= 10 * rtol * max(1.0, abs(b)) or \
        pytest.fail("%s: %r vs %r" % (k, a, b))                            # noqa: E731
    close(float(loss), float(g['loss']), 'loss')
    close(float(eS['loss']), float(g['loss_S']), 'loss_S')
    close(float(eT['loss']), float(g['loss_T']), 'loss_T')
    for tag, e in (("S_", eS), ("T_", eT)):
        inds = e['query_points_sample_inds'].cpu().numpy()
        same = np.array_equal(inds, g[tag + 'query_points_sample_inds'])
        assert np.array_equal(np.sort(inds, 1), np.sort(g[tag + 'query_points_sample_inds'], 1))
        np.testing.assert_allclose(e['global_d_pred'].detach().cpu().numpy(),
                                   g[tag + 'global_d_pred'], rtol=10 * rtol, atol=10 * rtol)
        if same:
            np.testing.assert_allclose(e['last_local_d_pred'].detach().cpu().numpy(),
                                       g[tag + 'last_local_d_pred'], rtol=10 * rtol,
                                       atol=10 * rtol)
            for k in ('last_objectness_label', 'last_object_assignment'):
                np.testing.assert_array_equal(e[k].cpu().numpy(), g[tag + k])
        for k in ('query_points_generation_loss', 'sum_heads_objectness_loss',
                  'sum_heads_box_loss', 'sum_heads_sem_cls_loss'):
            close(float(e[k]), float(g[tag + k]), tag + k)
        for p in ('proposal_', 'last_', '2head_'):
            for k in ('objectness_loss', 'center_loss', 'size_cls_loss', 'box_loss',
                      'sem_cls_loss'):
                close(float(e[p + k]), float(g[tag + p + k]), tag + p + k)
    grads = {'grad_global_netD2': net.global_netD2.weight.grad,
             'grad_decoder_netD6': net.decoder_netD[6].weight.grad,
             'grad_global_netD1_0': net.global_netD1[0].weight.grad,
             'grad_sa1_w0': net.backbone_net.sa1.mlp_module.layer0.conv.weight.grad,
             'grad_dec5_linear2': net.decoder[5].linear2.weight.grad,
             'grad_proposal_conv1': net.proposal_head.conv1.weight.grad}
    for k, t in grads.items():
        a = t.detach().cpu().numpy().astype(np.float32).ravel()[::53]
        want = g[k + '_sample']
        rel = np.linalg.norm(a - want) / (np.linalg.norm(want) + 1e-30)
        assert rel <= grad_rtol, (k, rel)


def test_groupfree_br_step_matches_reference_cpu(oracle_ext, monkeypatch):
    monkeypatch.setenv("BTR_FUSED_SA", "0")
    check_br(run_br(torch.device("cpu")), rtol=1e-4, grad_rtol=1e-3)


@pytest.mark.gpu
def test_groupfree_br_step_matches_reference_gpu(cuda):
    check_br(run_br(cuda), rtol=1e-4, grad_rtol=2e-2)


@pytest.mark.gpu
def test_groupfree_br_train_steps(cuda):
    from backtoreality_amd.groupfree import train as gf_train
    cfg = config.scannet_md40()
    net = gf_train.build_model(cfg, cuda, domain_adaptation=True)
    opt = gf_train.make_optimizer(net)
    bS = synthetic.make_batch(0, 2, 8192, cfg, use_height=False, device=cuda)
    bT = synthetic.make_batch(50, 2, 8192, cfg, use_height=False, device=cuda)
    for _ in range(2):
        loss, eS, eT = gf_train.train_step_br(net, opt, bS, bT, cfg)
    assert np.isfinite(float(loss))
    assert not [n for n, p in net.named_parameters() if p.grad is None]
    assert eT['last_local_d_pred'].shape == (2, 1, 256) and eS['global_d_pred'].shape == (2, 2)


# ------------------------------------------------------------------ CenterRefine step (8f #2/#3)
GOLD_CR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                       "groupfree_cr_step.npz")


def run_cr(device):
    cfg = config.scannet_md40()
    bS = synthetic.make_batch(0, 2, 4096, cfg, use_height=False, center_jitter=0.1, device=device)
    bT = synthetic.make_batch(100, 2, 4096, cfg, use_height=False, center_jitter=0.1,
                              device=device)
    torch.manual_seed(0)
    net = groupfree.GroupFreeDetector_DA_jitter(cfg.num_class, cfg.num_heading_bin,
                                                cfg.num_size_cluster, cfg.mean_size_arr,
                                                input_feature_dim=0, num_proposal=256,
                                                dropout=0.0).to(device)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    names = sorted(sd)
    sig = (names, np.array([float(sd[k].double().sum()) for k in names]),
           [str(tuple(sd[k].shape)) for k in names])
    eS = net({'point_clouds': bS['point_clouds']}, bS['center_label'], bS['sem_cls_label'])
    eT = net({'point_clouds': bT['point_clouds']}, bT['center_label'], bT['sem_cls_label'])
    eS.update(bS)
    eT.update(bT)
    before = bS['center_label'].clone()
    loss, eS, eT = groupfree.get_loss_DA_jitter(eS, eT, 30, cfg, **LOSS_ARGS)
    assert torch.equal(bS['center_label'], before)      # the caller's batch is not edited
    loss.backward()
    return net, sig, loss, eS, eT


def check_cr(res, rtol, grad_rtol):
    g = np.load(GOLD_CR)
    net, (names, sums, shapes), loss, eS, eT = res
    assert names == list(g['state_names']) and shapes == list(g['state_shapes'])
    np.testing.assert_allclose(sums, g['state_sums'], rtol=1e-6, atol=1e-6)
    for a, k in ((loss, 'loss'), (eS['loss'], 'loss_S'), (eT['loss'], 'loss_T'),
                 (eS['jitter_loss'], 'jitter_loss')):
        assert abs(float(a) - float(g[k])) <= 10 * rtol * max(1.0, abs(float(g[k]))), k
    for tag, e in (("S_", eS), ("T_", eT)):
        for k in ('jitter_pred', 'center_label'):
            want = g[tag + k]
            np.testing.assert_allclose(e[k].detach().cpu().numpy(), want, rtol=10 * rtol,
                                       atol=10 * rtol * np.abs(want).max())
    _sub(eS['center_features'], g['S_center_features_sample'], g['S_center_features_sums'],
         10 * rtol, 37)
    grads = {'grad_jitter_net3': net.jitter_net[3].weight.grad,
             'grad_ctjt_w0': net.backbone_net.ctjt_head.mlp_module.layer0.conv.weight.grad,
             'grad_sa1_w0': net.backbone_net.sa1.mlp_module.layer0.conv.weight.grad}
    for k, t in grads.items():
        a = t.detach().cpu().numpy().astype(np.float32).ravel()[::11]
        want = g[k + '_sample']
        rel = np.linalg.norm(a - want) / (np.linalg.norm(want) + 1e-30)
        assert rel <= grad_rtol, (k, rel)


def test_groupfree_cr_step_matches_reference_cpu(oracle_ext, monkeypatch):
    monkeypatch.setenv("BTR_FUSED_SA", "0")
    check_cr(run_cr(torch.device("cpu")), rtol=1e-4, grad_rtol=1e-3)


@pytest.mark.gpu
def test_groupfree_cr_step_matches_reference_gpu(cuda):
    check_cr(run_cr(cuda), rtol=1e-4, grad_rtol=2e-2)
    from backtoreality_amd.groupfree import train as gf_train
    cfg = config.scannet_md40()
    net = gf_train.build_model(cfg, cuda, center_refine=True)
    opt = gf_train.make_optimizer(net)
    bS = synthetic.make_batch(0, 2, 8192, cfg, use_height=False, center_jitter=0.1, device=cuda)
    bT = synthetic.make_batch(9, 2, 8192, cfg, use_height=False, center_jitter=0.1, device=cuda)
    loss, eS, eT = gf_train.train_step_br_jitter(net, opt, bS, bT, cfg, epoch=60)
    assert np.isfinite(float(loss)) and eT['jitter_pred'].shape == (2, 3, 64)


# ------------------------------------------------------------ weakly supervised baseline (WSB)
def run_wsb(device):
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(0, 2, 4096, cfg, use_height=False, device=device)
    torch.manual_seed(0)
    net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                      cfg.mean_size_arr, input_feature_dim=0, num_proposal=256,
                                      dropout=0.0).to(device)
    e = net({'point_clouds': batch['point_clouds']})
    e.update(batch)
    loss, e = groupfree.get_loss_weak(e, cfg, **LOSS_ARGS)
    loss.backward()
    return net, loss, e


def check_wsb(res, rtol, grad_rtol):
    g = np.load(os.path.join(os.path.dirname(GOLD), "groupfree_wsb_step.npz"))
    net, loss, e = res
    assert abs(float(loss) - float(g['loss'])) <= 10 * rtol * abs(float(g['loss']))
    for k in ('query_points_generation_loss', 'sum_heads_objectness_loss', 'sum_heads_box_loss',
              'sum_heads_sem_cls_loss'):
        assert abs(float(e[k]) - float(g[k])) <= 10 * rtol * max(1.0, abs(float(g[k]))), k
    for p in ('proposal_', 'last_', '3head_'):
        for k in ('objectness_loss', 'center_loss', 'size_cls_loss', 'box_loss', 'sem_cls_loss'):
            a, b = float(e[p + k]), float(g[p + k])
            assert abs(a - b) <= 10 * rtol * max(1.0, abs(b)), (p + k, a, b)
    for k, t, st in (('grad_sa1_w0', net.backbone_net.sa1.mlp_module.layer0.conv.weight.grad, 11),
                     ('grad_dec5_linear2', net.decoder[5].linear2.weight.grad, 53)):
        a = t.detach().cpu().numpy().astype(np.float32).ravel()[::st]
        want = g[k + '_sample']
        rel = np.linalg.norm(a - want) / (np.linalg.norm(want) + 1e-30)
        assert rel <= grad_rtol, (k, rel)


def test_groupfree_wsb_step_matches_reference_cpu(oracle_ext, monkeypatch):
    """train_GF_WSB.py:217: GroupFreeDetector + get_loss_weak."""
    monkeypatch.setenv("BTR_FUSED_SA", "0")
    check_wsb(run_wsb(torch.device("cpu")), rtol=1e-4, grad_rtol=1e-3)


@pytest.mark.gpu
def test_groupfree_wsb_step_matches_reference_gpu(cuda):
    check_wsb(run_wsb(cuda), rtol=1e-4, grad_rtol=2e-2)
    from backtoreality_amd.groupfree import train as gf_train
    cfg = config.scannet_md40()
    net = gf_train.build_model(cfg, cuda)
    opt = gf_train.make_optimizer(net)
    batch = synthetic.make_batch(0, 2, 8192, cfg, use_height=False, device=cuda)
    loss, _ = gf_train.train_step(net, opt, batch, cfg, criterion=groupfree.get_loss_weak)
    assert np.isfinite(float(loss))


@pytest.mark.gpu
def test_groupfree_evaluate_one_epoch(cuda):
    from backtoreality_amd.groupfree import train as gf_train
    cfg = config.scannet_md40()
    net = gf_train.build_model(cfg, cuda, num_decoder_layers=2)
    batches = [synthetic.make_batch(3 * i, 2, 4096, cfg, use_height=False, device=cuda)
               for i in range(2)]
    stats, metrics = gf_train.evaluate_one_epoch(net, batches, cfg,
                                                 loss_args={'num_decoder_layers': 2})
    assert net.training and np.isfinite(stats['loss'])
    assert sorted(metrics) == [0.25, 0.5]
    assert sorted(metrics[0.25]) == ['0head_', 'last_', 'proposal_']
    assert 'mAP' in metrics[0.5]['last_']
    # both thresholds come from one pass over the boxes: the same as a calculator per threshold
    from backtoreality_amd.votenet import ap_helper, train
    cd = dict(train.EVAL_CONFIG_DICT, conf_thresh=0.0, dataset_config=cfg)
    calc = {thr: ap_helper.APCalculator(ap_iou_thresh=thr) for thr in (0.25, 0.5)}
    net.eval()
    with torch.no_grad():
        for batch in batches:
            end = net({'point_clouds': batch['point_clouds']})
            end.update(batch)
            pred = ap_helper.parse_predictions(end, cd, 'last_')
            gt = ap_helper.parse_groundtruths(end, cd)
            for c in calc.values():
                c.step(pred, gt)
    for thr in (0.25, 0.5):
        want = calc[thr].compute_metrics()
        got = metrics[thr]['last_']
        assert sorted(want) == sorted(got)
        for k in want:
            assert np.allclose(float(want[k]), float(got[k]), rtol=0, atol=1e-12, equal_nan=True)


@pytest.mark.gpu
def test_fast_adamw_folded_clipping_on_the_fused_kernels(cuda):
    """FastAdamW.step(clip_norm=c) on the GPU (the clip factor rides into torch's fused AdamW
    kernel as its grad_scale operand) against clip_grad_norm_ + torch.optim.AdamW."""
    import copy
    from backtoreality_amd.votenet.train import FastAdamW
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(50, 70), torch.nn.ReLU(),
                                torch.nn.Linear(70, 30)).to(cuda)
    net_b = copy.deepcopy(net_a)
    groups = lambda n: [{"params": list(n[0].parameters())},
                        {"params": list(n[2].parameters()), "lr": 0.001}]
    opt_a = FastAdamW(groups(net_a), lr=0.01, weight_decay=0.05, fused=True)
    opt_b = torch.optim.AdamW(groups(net_b), lr=0.01, weight_decay=0.05, fused=True)
    x = torch.randn(64, 50, device=cuda)
    for i in range(4):
        for net, opt in ((net_a, opt_a), (net_b, opt_b)):
            opt.zero_grad(set_to_none=True)
            (net(x) ** 2).sum().backward()
        total = opt_a.step(clip_norm=0.1)
        ref = torch.nn.utils.clip_grad_norm_(net_b.parameters(), 0.1)
        opt_b.step()
        assert torch.allclose(total, ref, rtol=1e-6)
        if i:
            assert opt_a.param_groups[0].get('_btr_fast') is not None   # the lean path ran
    for a, b in zip(net_a.parameters(), net_b.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_decoder_stack_keeps_the_module_loop_on_the_cpu(oracle_ext, monkeypatch):
    """groupfree/fused_stack.py (the decoder loop as one library call per direction) covers CUDA
    tensors only: on the CPU the detector runs the module loop and says why."""
    from backtoreality_amd.groupfree import fused_stack
    monkeypatch.setenv("BTR_FUSED_SA", "0")
    cfg = config.scannet_md40()
    torch.manual_seed(0)
    net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                      cfg.mean_size_arr, input_feature_dim=0, num_proposal=32,
                                      num_decoder_layers=2, self_position_embedding='loc_learned')
    batch = synthetic.make_batch(0, 2, 4096, cfg, use_height=False, device=torch.device("cpu"))
    calls, refused = fused_stack.CALLS[0], sum(fused_stack.REFUSED.values())
    ep = net({'point_clouds': batch['point_clouds']})
    assert fused_stack.CALLS[0] == calls and sum(fused_stack.REFUSED.values()) == refused + 1
    assert ep['last_center'].shape == (2, 32, 3) and '0head_center' in ep
