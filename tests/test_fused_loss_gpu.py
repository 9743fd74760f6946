"""The fused VoteNet loss (csrc/votenet_loss.hip, votenet/fused_loss.py) against the op-by-op
torch composition of loss_helper.py -- the reference's own arithmetic
(detection/Votenet/models/loss_helper.py:336-400), which tests/test_golden_cpu.py pins to the
reference's outputs.  Scalars and gradients within 1e-5 relative (f32 sums in a different
order), labels / assignments bit-exact."""
import numpy as np
import pytest
import torch

from backtoreality_amd.votenet import config, fused_loss, loss_helper, proposal_module, synthetic

pytestmark = pytest.mark.gpu


def _case(cfg, dev, B, K, S1, N, seed, near_gt=True):
    """Random head outputs over a synthetic labelled batch; proposals scattered around the GT
    centres so that positives, negatives and the grey zone all occur."""
    g = torch.Generator().manual_seed(seed)
    batch = synthetic.make_batch(seed, B, N, cfg, device=dev)
    cout = 5 + 2 * cfg.num_heading_bin + 4 * cfg.num_size_cluster + cfg.num_class
    net = torch.randn(B, cout, K, generator=g).mul(1.5).to(dev)
    gt = batch['center_label']
    K2 = gt.shape[1]
    pick = torch.randint(0, K2, (B, K), generator=g).to(dev)
    agg = torch.gather(gt, 1, pick.unsqueeze(-1).expand(-1, -1, 3))
    agg = agg + torch.randn(B, K, 3, generator=g).to(dev) * (0.35 if near_gt else 3.0)
    seed_inds = torch.randint(0, N, (B, S1), generator=g).int().to(dev)
    seed_xyz = torch.gather(batch['point_clouds'][..., :3], 1,
                            seed_inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    vote_xyz = seed_xyz + torch.randn(B, S1, 3, generator=g).to(dev) * 0.4
    return batch, net, agg.contiguous(), seed_inds, seed_xyz, vote_xyz.contiguous()


def _run(cfg, case, fused, monkeypatch, gscale=1.0):
    batch, net, agg, seed_inds, seed_xyz, vote_xyz = case
    monkeypatch.setenv("BTR_FUSED_LOSS", "1" if fused else "0")
    net = net.clone().requires_grad_(True)
    agg = agg.clone().requires_grad_(True)
    vote_xyz = vote_xyz.clone().requires_grad_(True)
    end = {'aggregated_vote_xyz': agg, 'seed_xyz': seed_xyz, 'seed_inds': seed_inds,
           'vote_xyz': vote_xyz, fused_loss.HEAD_KEY: net}
    proposal_module.decode_scores(net, end, cfg.num_class, cfg.num_heading_bin,
                                  cfg.num_size_cluster, cfg.mean_size_arr)
    end.update(batch)
    assert fused_loss.can_fuse(end, cfg) == fused
    loss, end = loss_helper.get_loss(end, cfg)
    (loss * gscale).backward()
    return end, net.grad, agg.grad, vote_xyz.grad


def _close(a, b, tol=1e-5):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) <= tol * (float(b.abs().max()) + 1e-12)


@pytest.mark.parametrize("name,B,K,S1,N,near", [
    ("scannet", 8, 256, 1024, 20000, True),
    ("scannet", 3, 256, 1024, 5000, False),     # no positives at all: sums of empty masks
    ("matterport", 2, 256, 1024, 8000, True),   # 12 heading bins, 256 GT slots
    ("scannet", 2, 300, 700, 3000, True),       # K and S1 not multiples of the block size
])
def test_fused_loss_matches_torch_composition(cuda, monkeypatch, name, B, K, S1, N, near):
    cfg = config.scannet_md40() if name == "scannet" else config.matterport_md40()
    case = _case(cfg, cuda, B, K, S1, N, seed=11, near_gt=near)
    e_t, gn_t, ga_t, gv_t = _run(cfg, case, False, monkeypatch)
    e_f, gn_f, ga_f, gv_f = _run(cfg, case, True, monkeypatch)
    assert torch.equal(e_f['objectness_label'], e_t['objectness_label'])
    assert torch.equal(e_f['objectness_mask'], e_t['objectness_mask'])
    # ties between identical (zero-padded) GT slots may resolve to different slots; their
    # labels are identical, so compare what the assignment is used for
    for k in ('heading_class_label', 'size_class_label', 'sem_cls_label'):
        assert torch.equal(torch.gather(e_f[k], 1, e_f['object_assignment']),
                           torch.gather(e_t[k], 1, e_t['object_assignment'])), k
    if near:
        assert int(e_t['objectness_label'].sum()) > 0
    for k in fused_loss.STAT_KEYS:
        a, b = float(e_f[k]), float(e_t[k])
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (k, a, b)
    assert e_f['objectness_label'].dtype == torch.int64
    assert e_f['object_assignment'].dtype == torch.int64
    assert _close(gn_f, gn_t), float((gn_f - gn_t).abs().max())
    assert _close(ga_f, ga_t)
    assert _close(gv_f, gv_t)


def test_fused_loss_scales_with_upstream_gradient(cuda, monkeypatch):
    cfg = config.scannet_md40()
    case = _case(cfg, cuda, 2, 256, 1024, 4000, seed=5)
    _, gn1, ga1, gv1 = _run(cfg, case, True, monkeypatch)
    _, gn3, ga3, gv3 = _run(cfg, case, True, monkeypatch, gscale=-2.5)
    assert _close(gn3, gn1 * -2.5, 1e-6) and _close(ga3, ga1 * -2.5, 1e-6)
    assert _close(gv3, gv1 * -2.5, 1e-6)


def test_in_place_edit_of_the_loss_leaves_the_reported_terms_alone(cuda, monkeypatch):
    """The returned loss is a 0-dim tensor on the statistics vector's storage (no copy launch),
    which autograd does not know about: it lives on a word of its own (the kernel writes the total
    twice), so `loss *= w` must not change any reported term, and the backward must scale with
    the edit."""
    cfg = config.scannet_md40()
    case = _case(cfg, cuda, 2, 256, 1024, 4000, seed=7)
    e1, gn1, _, _ = _run(cfg, case, True, monkeypatch)
    before = {k: float(e1[k]) for k in fused_loss.STAT_KEYS}
    batch, net, agg, seed_inds, seed_xyz, vote_xyz = case
    net = net.clone().requires_grad_(True)
    end = {'aggregated_vote_xyz': agg.clone().requires_grad_(True), 'seed_xyz': seed_xyz,
           'seed_inds': seed_inds, 'vote_xyz': vote_xyz.clone().requires_grad_(True),
           fused_loss.HEAD_KEY: net}
    proposal_module.decode_scores(net, end, cfg.num_class, cfg.num_heading_bin,
                                  cfg.num_size_cluster, cfg.mean_size_arr)
    end.update(batch)
    loss, end = loss_helper.get_loss(end, cfg)
    reported = float(loss)
    loss *= 2.0
    assert float(loss) == 2.0 * reported
    for k in fused_loss.STAT_KEYS:
        if k != 'loss':   # (end_points['loss'] IS the returned tensor, as in the reference)
            assert float(end[k]) == before[k], k
    loss.backward()
    assert _close(net.grad, gn1 * 2.0, 1e-6)


def test_fused_loss_is_the_path_a_training_step_takes(cuda, monkeypatch):
    """End to end: VoteNet forward -> get_loss -> backward with and without the fused loss."""
    from backtoreality_amd.votenet import train
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(0, 2, 8192, cfg, device=cuda)
    out = {}
    for fused in (True, False):
        monkeypatch.setenv("BTR_FUSED_LOSS", "1" if fused else "0")
        net = train.build_model(cfg, cuda, seed=0)
        end = net({'point_clouds': batch['point_clouds']})
        end.update(batch)
        assert fused_loss.can_fuse(end, cfg) == fused
        loss, end = loss_helper.get_loss(end, cfg)
        loss.backward()
        out[fused] = (float(loss), end['aggregated_vote_inds'].clone(),
                      {n: p.grad.clone() for n, p in net.named_parameters()})
    (lf, inds_f, gf), (lt, inds_t, gt) = out[True], out[False]
    assert torch.equal(inds_f, inds_t)
    assert abs(lf - lt) <= 1e-5 * abs(lt)
    gmax = max(float(g.abs().max()) for g in gt.values())
    for n in gt:
        if float(gt[n].abs().max()) > 1e-4 * gmax:
            rel = float((gf[n] - gt[n]).norm() / gt[n].norm())
            assert rel < 2e-3, (n, rel)


def test_fused_loss_falls_back_only_when_it_must(cuda, monkeypatch):
    cfg = config.scannet_md40()
    case = _case(cfg, cuda, 2, 256, 1024, 4000, seed=5)
    batch, net, agg, seed_inds, seed_xyz, vote_xyz = case
    end = {'aggregated_vote_xyz': agg, 'seed_xyz': seed_xyz, 'seed_inds': seed_inds,
           'vote_xyz': vote_xyz, fused_loss.HEAD_KEY: net}
    end.update(batch)
    monkeypatch.setenv("BTR_FUSED_LOSS", "1")
    assert fused_loss.can_fuse(end, cfg)
    two = dict(end, vote_xyz=torch.cat([vote_xyz, vote_xyz], 1))  # vote_factor 2
    assert not fused_loss.can_fuse(two, cfg)
    assert not fused_loss.can_fuse({k: v for k, v in end.items() if k != fused_loss.HEAD_KEY},
                                   cfg)
    monkeypatch.setenv("BTR_FUSED_LOSS", "0")
    assert not fused_loss.can_fuse(end, cfg)


def test_fused_da_loss_matches_torch_composition(cuda, monkeypatch):
    """get_loss_DA (Back-to-Reality): both branches through the fused kernels (weak vote loss,
    source / target term weights) against the torch composition."""
    cfg = config.scannet_md40()
    case_S = _case(cfg, cuda, 4, 256, 1024, 6000, seed=21)
    case_T = _case(cfg, cuda, 4, 256, 1024, 6000, seed=22)
    g = torch.Generator().manual_seed(3)
    gd = [torch.randn(4, 2, generator=g).to(cuda) for _ in range(2)]
    ld = [torch.rand(4, 1, 256, generator=g).to(cuda) for _ in range(2)]

    def run(fused):
        monkeypatch.setenv("BTR_FUSED_LOSS", "1" if fused else "0")
        ends, leaves = [], []
        for case, gdp, ldp in zip((case_S, case_T), gd, ld):
            batch, net, agg, seed_inds, seed_xyz, vote_xyz = case
            net = net.clone().requires_grad_(True)
            agg = agg.clone().requires_grad_(True)
            vote = vote_xyz.clone().requires_grad_(True)
            gdp = gdp.clone().requires_grad_(True)
            ldp = ldp.clone().requires_grad_(True)
            end = {'aggregated_vote_xyz': agg, 'seed_xyz': seed_xyz, 'seed_inds': seed_inds,
                   'vote_xyz': vote, fused_loss.HEAD_KEY: net, 'global_d_pred': gdp,
                   'local_d_pred': ldp}
            proposal_module.decode_scores(net, end, cfg.num_class, cfg.num_heading_bin,
                                          cfg.num_size_cluster, cfg.mean_size_arr)
            end.update(batch)
            ends.append(end)
            leaves.append((net, agg, vote, gdp, ldp))
        loss, eS, eT = loss_helper.get_loss_DA(ends[0], ends[1], cfg)
        loss.backward()
        return loss.detach(), eS, eT, [[t.grad for t in lv] for lv in leaves]

    l_t, eS_t, eT_t, g_t = run(False)
    l_f, eS_f, eT_f, g_f = run(True)
    assert abs(float(l_f) - float(l_t)) <= 1e-5 * abs(float(l_t))
    for k in ('vote_loss', 'objectness_loss', 'center_loss', 'heading_cls_loss',
              'heading_reg_loss', 'size_cls_loss', 'size_reg_loss', 'sem_cls_loss', 'box_loss',
              'pos_ratio', 'neg_ratio', 'obj_acc', 'DA_loss'):
        a, b = float(eS_f[k]), float(eS_t[k])
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), ("S", k, a, b)
    for k in ('vote_loss', 'objectness_loss', 'center_loss', 'size_cls_loss', 'sem_cls_loss'):
        a, b = float(eT_f[k]), float(eT_t[k])
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), ("T", k, a, b)
    for e_f, e_t in ((eS_f, eS_t), (eT_f, eT_t)):
        assert torch.equal(e_f['objectness_label'], e_t['objectness_label'])
        assert torch.equal(e_f['objectness_mask'], e_t['objectness_mask'])
    for branch in range(2):
        for a, b in zip(g_f[branch], g_t[branch]):
            assert _close(a, b), (branch, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("B,K", [(8, 256), (3, 100), (1, 17)])
def test_fused_domain_loss_matches_torch_composition(cuda, monkeypatch, B, K):
    """loss_helper._domain_loss (the domain-adaptation term of get_loss_DA, reference
    loss_helper.py:618-650): one launch each way against the torch composition -- value 1e-6,
    every gradient 1e-5 of its largest entry, with an upstream gradient other than 1."""
    g = torch.Generator().manual_seed(B * 1000 + K)
    mk = lambda *s: torch.randn(*s, generator=g).to(cuda)
    base = {t: {'global_d_pred': mk(B, 2) * 2, 'local_d_pred': torch.sigmoid(mk(B, 1, K)),
                'objectness_label': (torch.rand(B, K, generator=g) < 0.3).long().to(cuda)}
            for t in "ST"}
    res = {}
    for fused in (False, True):
        monkeypatch.setenv("BTR_FUSED_LOSS", "1" if fused else "0")
        ends = {t: {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 else v)
                    for k, v in base[t].items()} for t in "ST"}
        loss = loss_helper._domain_loss(ends["S"], ends["T"])
        (loss * -3.5).backward()
        res[fused] = (loss.detach(), [ends[t][k].grad for t in "ST"
                                      for k in ('global_d_pred', 'local_d_pred')])
    (lt, gt), (lf, gf) = res[False], res[True]
    assert abs(float(lf) - float(lt)) <= 1e-6 * max(1.0, abs(float(lt)))
    for a, b in zip(gf, gt):
        assert a.shape == b.shape
        assert _close(a, b), float((a - b).abs().max())
    # GroupFree3D's form (loss_helper.py:673-712): the same terms unweighted, keys of its last head
    gfe = {t: {'global_d_pred': base[t]['global_d_pred'].clone().requires_grad_(True),
               'last_local_d_pred': base[t]['local_d_pred'].clone().requires_grad_(True),
               'last_objectness_label': base[t]['objectness_label']} for t in "ST"}
    assert fused_loss.domain_loss_fusable(gfe["S"], gfe["T"], 'last_')
    l2 = fused_loss.domain_loss(gfe["S"], gfe["T"], 3.0, 1.0, 'last_')
    (l2 * -3.5).backward()
    assert abs(float(l2) - 2.0 * float(lt)) <= 2e-6 * max(1.0, abs(float(lt)))
    g2 = [gfe[t][k].grad for t in "ST" for k in ('global_d_pred', 'last_local_d_pred')]
    for a, b in zip(g2, gt):
        assert _close(a, 2.0 * b), float((a - 2.0 * b).abs().max())
