"""csrc/gf_loss.hip (per-head GroupFree3D loss + gradient in one launch) against the op-by-op
torch composition of groupfree/loss_helper.py, which restates the reference's
compute_objectness_loss_based_on_query_points / compute_box_and_sem_cls_loss
(detection/GroupFree3D/models/loss_helper.py:81-275) and is pinned by the reference golden."""
import numpy as np
import pytest
import torch

from backtoreality_amd.groupfree import fused_loss, loss_helper
from backtoreality_amd.groupfree.modules import PredictHead
from backtoreality_amd.votenet import config, synthetic

pytestmark = pytest.mark.gpu


def _end_points(dev, B, P, layers, seed):
    cfg = config.scannet_md40()
    torch.manual_seed(seed)
    batch = synthetic.make_batch(seed, B, 4096, cfg, use_height=False, device=dev)
    end = dict(batch)
    end['seed_inds'] = torch.randint(0, 4096, (B, 1024), device=dev, dtype=torch.int32)
    end['query_points_sample_inds'] = torch.randint(0, 1024, (B, P), device=dev,
                                                    dtype=torch.int32)
    base = torch.rand(B, P, 3, device=dev) * 4 - 2
    heads = []
    for prefix in loss_helper.head_prefixes(layers):
        head = PredictHead(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                           cfg.mean_size_arr, P, 288).to(dev)
        feats = torch.randn(B, 288, P, device=dev)
        head(feats, base_xyz=base, end_points=end, prefix=prefix)
        raw = end[prefix + '_head_output']
        raw.retain_grad()
        heads.append(raw)
    return cfg, end, heads


def _loss(cfg, end, layers):
    return loss_helper.get_loss(end, cfg, num_decoder_layers=layers,
                                query_points_generator_loss_coef=0.8, obj_loss_coef=0.1,
                                box_loss_coef=1, sem_cls_loss_coef=0.1, query_points_obj_topk=4)


@pytest.mark.parametrize("B,P,layers", [(4, 256, 6), (2, 100, 2), (3, 64, 0)])
def test_heads_loss_and_gradient_match_the_torch_composition(B, P, layers, monkeypatch):
    dev = torch.device("cuda:0")
    out = {}
    for fused in (True, False):
        monkeypatch.setenv("BTR_FUSED_GF_LOSS", "1" if fused else "0")
        cfg, end, heads = _end_points(dev, B, P, layers, seed=3)
        prefixes = loss_helper.head_prefixes(layers)
        assert fused_loss.can_fuse(end, cfg, prefixes, ('smoothl1',) * 3) == fused
        loss, end = _loss(cfg, end, layers)
        loss.backward()
        out[fused] = (float(loss), end, [h.grad.clone() for h in heads])
    (lf, ef, gf), (lt, et, gt) = out[True], out[False]
    assert abs(lf - lt) <= 1e-5 * abs(lt), (lf, lt)
    for prefix in loss_helper.head_prefixes(layers):
        assert torch.equal(ef[prefix + 'objectness_label'], et[prefix + 'objectness_label'])
        assert torch.equal(ef[prefix + 'object_assignment'], et[prefix + 'object_assignment'])
        assert torch.allclose(ef[prefix + 'objectness_mask'], et[prefix + 'objectness_mask'])
        for key in fused_loss.TERMS + ('pos_ratio', 'neg_ratio'):
            a, b = float(ef[prefix + key]), float(et[prefix + key])
            assert abs(a - b) <= 2e-5 * abs(b) + 1e-7, (prefix, key, a, b)
    for key in ('sum_heads_objectness_loss', 'sum_heads_box_loss', 'sum_heads_sem_cls_loss'):
        np.testing.assert_allclose(float(ef[key]), float(et[key]), rtol=2e-5)
    for a, b in zip(gf, gt):
        scale = float(b.abs().max())
        assert scale > 0
        assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale


def test_the_other_loss_forms_stay_on_the_torch_composition():
    dev = torch.device("cuda:0")
    cfg, end, _ = _end_points(dev, 2, 64, 1, seed=1)
    prefixes = loss_helper.head_prefixes(1)
    assert fused_loss.can_fuse(end, cfg, prefixes, ('smoothl1',) * 3)
    assert not fused_loss.can_fuse(end, cfg, prefixes, ('l1', 'smoothl1', 'smoothl1'))
    del end['last_' + fused_loss.HEAD_KEY]
    assert not fused_loss.can_fuse(end, cfg, prefixes, ('smoothl1',) * 3)


def test_fused_loss_is_refused_once_the_published_entries_stop_being_its_inputs():
    """Advisor finding (round 3): the kernel reads `_head_output` and ONE base_xyz.  A head with
    its own centre base, an in-place edit of a published entry, or a replaced entry must send the
    loss through the op-by-op composition (which reads the entries themselves)."""
    dev = torch.device("cuda:0")
    kinds = ('smoothl1',) * 3
    prefixes = loss_helper.head_prefixes(1)
    cfg, end, _ = _end_points(dev, 2, 64, 1, seed=1)
    assert fused_loss.can_fuse(end, cfg, prefixes, kinds)
    end2 = dict(end)
    end2['last_base_xyz'] = end['last_base_xyz'].clone()      # a refined base for one head
    assert not fused_loss.can_fuse(end2, cfg, prefixes, kinds)
    end3 = dict(end)
    end3['last_sem_cls_scores'] = end['last_sem_cls_scores'] * 2     # replaced entry
    assert not fused_loss.can_fuse(end3, cfg, prefixes, kinds)
    with torch.no_grad():
        end['proposal_objectness_scores'].add_(1.0)                  # in-place edit of a view
    assert not fused_loss.can_fuse(end, cfg, prefixes, kinds)


@pytest.mark.parametrize("B,K", [(4, 1024), (2, 1000), (1, 37)])
def test_focal_sum_matches_the_torch_composition(B, K):
    """The seed points' objectness term (loss_helper.py:17-78 after the labels are made) as one
    launch each way against sigmoid_focal_loss + the weights / sum / division around it: value
    1e-6, gradient 1e-5 (relative to its largest entry), including saturated logits."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * K)
    logits = (torch.randn(B, 1, K, generator=g) * 4).to(dev)
    logits[0, 0, :4] = torch.tensor([40.0, -40.0, 0.0, 1e-4], device=dev)
    label = (torch.rand(B, K, generator=g) < 0.2).long().to(dev)
    a = logits.clone().requires_grad_(True)
    weights = (label >= 0).float()
    weights = weights / torch.clamp(weights.sum(dim=1, keepdim=True), min=1.0)
    ref = loss_helper.sigmoid_focal_loss(a.view(B, K, 1), label.unsqueeze(-1).float(),
                                         weights).sum() / B
    (ref * 1.7).backward()
    b = logits.clone().requires_grad_(True)
    assert fused_loss.focal_sum_fusable(b, label)
    out = fused_loss.focal_sum(b.reshape(1, B * K), label, 1.0 / K, 1.0 / B)[0]
    (out * 1.7).backward()
    assert abs(float(out) - float(ref)) <= 1e-6 * abs(float(ref)) + 1e-9
    err = (a.grad - b.grad).abs().max().item() / a.grad.abs().max().item()
    assert err < 1e-5, err
    assert torch.isfinite(b.grad).all()


def test_focal_sum_per_head_matches_the_torch_composition():
    """The weakly supervised branch's objectness of the query points (loss_helper.py:416-476): one
    label per query for all seven heads, one sum per head."""
    dev = torch.device("cuda:0")
    H, B, K = 7, 4, 256
    g = torch.Generator().manual_seed(11)
    scores = (torch.randn(H, B, K, 1, generator=g) * 3).to(dev)
    label = (torch.rand(B, K, generator=g) < 0.3).long().to(dev)
    coef = torch.linspace(0.5, 2.0, H, device=dev)
    a = scores.clone().requires_grad_(True)
    mask = torch.ones((B, K), device=dev)
    weights = mask / torch.clamp(mask.sum(dim=1, keepdim=True), min=1.0)
    ref = loss_helper.sigmoid_focal_loss(a.reshape(-1, K, 1),
                                         label.unsqueeze(-1).float().repeat(H, 1, 1),
                                         weights.repeat(H, 1)).view(H, -1).sum(1) / B
    (ref * coef).sum().backward()
    b = scores.clone().requires_grad_(True)
    assert fused_loss.focal_sum_fusable(b, label)
    out = fused_loss.focal_sum(b.reshape(H, B * K), label, 1.0 / K, 1.0 / B)
    (out * coef).sum().backward()
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-9)
    err = (a.grad - b.grad).abs().max().item() / a.grad.abs().max().item()
    assert err < 1e-5, err


@pytest.mark.parametrize("B,P,layers", [(4, 256, 6), (2, 100, 2)])
def test_weak_heads_loss_and_gradient_match_the_torch_composition(B, P, layers, monkeypatch):
    """get_loss_weak (centre labels only, loss_helper.py:416-606) through btr_gf_loss_weak_fwd
    against the torch composition: every published term, the labels, the gradient of every
    head's raw output (dead-zone centre term, no heading / size-residual gradient)."""
    dev = torch.device("cuda:0")
    out = {}
    for fused in (True, False):
        monkeypatch.setenv("BTR_FUSED_GF_LOSS", "1" if fused else "0")
        cfg, end, heads = _end_points(dev, B, P, layers, seed=5)
        # query points near the labelled centres, so that a fair share is positive
        centre = end['center_label'][:, :, 0:3]
        pick = torch.randint(0, centre.shape[1], (B, P), device=dev)
        end['query_points_xyz'] = torch.gather(centre, 1, pick.unsqueeze(2).expand(-1, -1, 3)) + \
            torch.randn(B, P, 3, device=dev) * 0.2
        end['seed_xyz'] = torch.rand(B, 1024, 3, device=dev) * 4 - 2
        end['seeds_obj_cls_logits'] = torch.randn(B, 1, 1024, device=dev)
        prefixes = loss_helper.head_prefixes(layers)
        assert fused_loss.can_fuse_weak(end, cfg, prefixes, 'smoothl1') == fused
        loss, end = loss_helper.get_loss_weak(
            end, cfg, num_decoder_layers=layers, query_points_generator_loss_coef=0.8,
            obj_loss_coef=0.1, box_loss_coef=1, sem_cls_loss_coef=0.1, query_points_obj_topk=4)
        loss.backward()
        out[fused] = (float(loss), end, [h.grad.clone() for h in heads])
    (lf, ef, gf), (lt, et, gt) = out[True], out[False]
    assert abs(lf - lt) <= 1e-5 * abs(lt), (lf, lt)
    assert float(et['last_objectness_label'].float().mean()) > 0.05
    for prefix in loss_helper.head_prefixes(layers):
        assert torch.equal(ef[prefix + 'objectness_label'], et[prefix + 'objectness_label'])
        assert torch.equal(ef[prefix + 'object_assignment'], et[prefix + 'object_assignment'])
        assert torch.allclose(ef[prefix + 'objectness_mask'], et[prefix + 'objectness_mask'])
        for key in ('objectness_loss', 'center_loss', 'size_cls_loss', 'box_loss', 'sem_cls_loss'):
            a, b = float(ef[prefix + key]), float(et[prefix + key])
            assert abs(a - b) <= 2e-5 * abs(b) + 1e-7, (prefix, key, a, b)
    for key in ('sum_heads_objectness_loss', 'sum_heads_box_loss', 'sum_heads_sem_cls_loss'):
        np.testing.assert_allclose(float(ef[key]), float(et[key]), rtol=2e-5)
    for a, b in zip(gf, gt):
        scale = float(b.abs().max())
        assert scale > 0
        assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale
