"""The decoder loop of GroupFreeDetector as one autograd node (groupfree/fused_stack.py ->
csrc/gf_stack.hip; reference: detection/GroupFree3D/models/detector.py:161-219) against the
per-module path it replaces: the same kernels in the same order, so everything is compared
bit for bit -- predictions of all heads, losses, and the gradient of every parameter."""
import ctypes
import itertools

import numpy as np
import pytest
import torch

from backtoreality_amd import groupfree
from backtoreality_amd.groupfree import fused_attention, fused_stack
from backtoreality_amd.pointnet2 import _ext
from backtoreality_amd.votenet import config, synthetic

LOSS_ARGS = dict(num_decoder_layers=6, query_points_generator_loss_coef=0.8, obj_loss_coef=0.1,
                 box_loss_coef=1, sem_cls_loss_coef=0.1, query_points_obj_topk=4)


def test_struct_mirrors_match_the_library():
    assert _ext._lib.btr_gf_stack_sizeof(0) == ctypes.sizeof(_ext.GfStack)
    assert _ext._lib.btr_gf_stack_sizeof(1) == ctypes.sizeof(_ext.GfStackPlan)


def _step(cuda, monkeypatch, stack, dropout, cls=None, self_pos='loc_learned',
          cross_pos='xyz_learned', extra_loss=False):
    monkeypatch.setenv("BTR_FUSED_GF_STACK", "1" if stack else "0")
    monkeypatch.setattr(fused_attention, "_calls", itertools.count())
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(3, 2, 8192, cfg, use_height=False, device=cuda)
    torch.manual_seed(0)
    cls = cls or groupfree.GroupFreeDetector
    net = cls(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster, cfg.mean_size_arr,
              input_feature_dim=0, num_proposal=256, dropout=dropout,
              self_position_embedding=self_pos, cross_position_embedding=cross_pos).to(cuda)
    calls = fused_stack.CALLS[0]
    end_points = net({'point_clouds': batch['point_clouds']})
    took = fused_stack.CALLS[0] - calls
    end_points.update(batch)
    loss, end_points = groupfree.get_loss(end_points, cfg, **LOSS_ARGS)
    if extra_loss:   # gradients through the decoded tensors and the last layer's output
        loss = loss + end_points['2head_center'].square().mean() + \
            end_points['last_pred_size'].sum() * 1e-3 + \
            end_points['0head_heading_residuals'].mean() + \
            end_points['4head_size_residuals'].square().mean()
        if 'last_local_d_pred' in end_points:
            loss = loss + end_points['last_local_d_pred'].mean()
    loss.backward()
    grads = {n: (None if p.grad is None else p.grad.detach().clone())
             for n, p in net.named_parameters()}
    bufs = {n: b.detach().clone() for n, b in net.named_buffers()}
    return took, float(loss), end_points, grads, bufs


def _compare(a, b):
    took_a, loss_a, ep_a, g_a, b_a = a
    took_b, loss_b, ep_b, g_b, b_b = b
    assert took_a == 1 and took_b == 0
    assert loss_a == loss_b
    for k, v in ep_b.items():
        if torch.is_tensor(v) and v.is_floating_point() and k in ep_a:
            assert torch.equal(ep_a[k], v), k
    keys = [k for k in ep_b if k.endswith('center') or k.endswith('_head_output')]
    assert len(keys) >= 14
    assert set(g_a) == set(g_b)
    exact = 0
    for n in g_b:
        assert (g_a[n] is None) == (g_b[n] is None), n
        if g_b[n] is None:
            continue
        if (n.startswith('backbone_net.') and not n.startswith('backbone_net.fp2')) or \
                n.startswith(('decoder_netD', 'global_netD')):
            # below the last feature propagation the module loop does not reproduce ITSELF bit
            # for bit (float atomics of the interpolation backward: 42 tensors differ by
            # 3e-8 .. 5e-6 between two runs, tools/diag_gf_stack.py); the discriminators are
            # stock convolutions (MIOpen's weight-gradient kernels)
            d = float((g_a[n] - g_b[n]).norm() / (g_b[n].norm() + 1e-30))
            assert d < 2e-5, (n, d)
        else:
            assert torch.equal(g_a[n], g_b[n]), n
            exact += 1
    assert exact >= 250   # the decoder, its embeddings, heads, projections, fp2
    for n in b_b:   # BatchNorm running statistics of the embeddings and heads
        assert torch.equal(b_a[n], b_b[n]), n


@pytest.mark.gpu
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_stack_equals_the_module_loop(cuda, monkeypatch, dropout):
    _compare(_step(cuda, monkeypatch, True, dropout), _step(cuda, monkeypatch, False, dropout))


@pytest.mark.gpu
def test_stack_with_gradients_through_the_decoded_tensors(cuda, monkeypatch):
    _compare(_step(cuda, monkeypatch, True, 0.0, extra_loss=True),
             _step(cuda, monkeypatch, False, 0.0, extra_loss=True))


@pytest.mark.gpu
def test_stack_domain_adaptation_variant(cuda, monkeypatch):
    """GroupFreeDetector_DA: the local discriminator reads the last layer's output."""
    cls = groupfree.GroupFreeDetector_DA
    _compare(_step(cuda, monkeypatch, True, 0.1, cls=cls, extra_loss=True),
             _step(cuda, monkeypatch, False, 0.1, cls=cls, extra_loss=True))


@pytest.mark.gpu
def test_stack_without_position_embeddings(cuda, monkeypatch):
    _compare(_step(cuda, monkeypatch, True, 0.0, self_pos='none', cross_pos='none'),
             _step(cuda, monkeypatch, False, 0.0, self_pos='none', cross_pos='none'))


@pytest.mark.gpu
def test_xyz_query_position_keeps_the_module_loop(cuda, monkeypatch):
    """'xyz_learned' embeds the centre alone: not what the decode kernel hands on."""
    took, loss, _ep, _g, _b = _step(cuda, monkeypatch, True, 0.0, self_pos='xyz_learned')
    assert took == 0 and np.isfinite(loss)


@pytest.mark.gpu
def test_eval_mode_keeps_the_module_loop(cuda, monkeypatch):
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(3, 2, 8192, cfg, use_height=False, device=cuda)
    net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                      cfg.mean_size_arr, input_feature_dim=0,
                                      num_proposal=256).to(cuda).eval()
    calls = fused_stack.CALLS[0]
    with torch.no_grad():
        ep = net({'point_clouds': batch['point_clouds']})
    assert fused_stack.CALLS[0] == calls and 'last_center' in ep


def _several_steps(cuda, monkeypatch, graphs, n=5, cls=None):
    """n forward + backward passes of one detector on two alternating batches with new dropout
    masks every time; the decoder stack's calls are replayed HIP graphs on three lanes from the
    third pass on (graphs) or single launches on one stream (not graphs).  (No parameter update:
    the backbone's backward is not bit-reproducible -- float atomics -- and the query points are
    a top-k of its output, so two runs that update would part ways for reasons outside the
    stack.)"""
    monkeypatch.setenv("BTR_FUSED_GF_STACK", "1")
    monkeypatch.setenv("BTR_GF_SLOTS", "1")
    monkeypatch.setenv("BTR_GRAPHS", "1" if graphs else "0")
    monkeypatch.setattr(fused_attention, "_calls", itertools.count())
    cfg = config.scannet_md40()
    batches = [synthetic.make_batch(s, 2, 8192, cfg, use_height=False, device=cuda)
               for s in (3, 4)]
    torch.manual_seed(0)
    cls = cls or groupfree.GroupFreeDetector
    net = cls(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster, cfg.mean_size_arr,
              input_feature_dim=0, num_proposal=256, dropout=0.1,
              self_position_embedding='loc_learned',
              cross_position_embedding='xyz_learned').to(cuda)
    before, calls = _ext.graph_stats(), fused_stack.CALLS[0]
    out = []
    for it in range(n):
        fused_attention.bump_step(cuda)           # new dropout masks every step
        batch = batches[it % 2]
        for p in net.parameters():
            p.grad = None
        end_points = net({'point_clouds': batch['point_clouds']})
        end_points.update(batch)
        loss, end_points = groupfree.get_loss(end_points, cfg, **LOSS_ARGS)
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()
                 if p.grad is not None and k.startswith(('decoder.', 'prediction_heads.'))}
        out.append((float(loss), end_points['last_center'].detach().clone(), grads))
    after = _ext.graph_stats()
    assert fused_stack.CALLS[0] - calls == n
    return out, {k: after[k] - before[k] for k in after}


@pytest.mark.gpu
def test_replayed_graphs_on_three_lanes_equal_the_single_launches(cuda, monkeypatch):
    """csrc/graph_cache.hip + the lanes of csrc/gf_stack.hip: the same launches with the same
    operands, so every step's loss, predictions and decoder / head gradients are the single-stream
    sequence's bit for bit -- including the steps that only replay."""
    step0 = int(fused_attention.step_counter(cuda).item())
    a, used = _several_steps(cuda, monkeypatch, True)
    fused_attention.step_counter(cuda).fill_(step0)      # the same masks for the second run
    b, unused = _several_steps(cuda, monkeypatch, False)
    assert used["captures"] >= 20 and used["replays"] >= 2 * used["captures"], used
    assert unused["captures"] == 0 and unused["replays"] == 0, unused
    for it, ((la, ca, ga), (lb, cb, gb)) in enumerate(zip(a, b)):
        assert torch.equal(ca, cb), it
        assert len(ga) >= 150 and set(ga) == set(gb)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), (it, k)
        assert abs(la - lb) <= 1e-6 * abs(lb), (it, la, lb)   # (the backbone's float atomics)


@pytest.mark.gpu
def test_cached_coverage_checks_follow_the_modules(cuda, monkeypatch):
    """run() keeps the outcome of its coverage checks with the detector; a module whose state
    changes between two calls (a dropout rate, a BatchNorm put into eval mode, a replaced
    parameter) must be seen by the next call."""
    monkeypatch.setenv("BTR_FUSED_GF_STACK", "1")
    cfg = config.scannet_md40()
    batch = synthetic.make_batch(3, 2, 8192, cfg, use_height=False, device=cuda)
    torch.manual_seed(0)
    net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                      cfg.mean_size_arr, input_feature_dim=0, num_proposal=256,
                                      dropout=0.1, self_position_embedding='loc_learned',
                                      cross_position_embedding='xyz_learned').to(cuda)

    def took():
        calls = fused_stack.CALLS[0]
        ep = net({'point_clouds': batch['point_clouds']})
        return fused_stack.CALLS[0] - calls, ep

    assert took()[0] == 1 and took()[0] == 1
    net.decoder[2].dropout1.p = 0.3                     # rates differ inside a layer
    assert took()[0] == 0
    net.decoder[2].dropout1.p = 0.1
    assert took()[0] == 1
    net.prediction_heads[1].bn1.eval()                  # a chain that is no longer covered
    assert took()[0] == 0
    net.prediction_heads[1].bn1.train()
    assert took()[0] == 1
    # a replaced parameter object: the next call must read the new one
    head = net.prediction_heads[5].center_residual_head
    _, before = took()
    with torch.no_grad():
        head.weight = torch.nn.Parameter(head.weight.detach().clone() * 0.0)
        head.bias = torch.nn.Parameter(head.bias.detach().clone() * 0.0 + 0.25)
    n, after = took()
    assert n == 1
    assert not torch.equal(before['last_center'], after['last_center'])
    assert torch.allclose(after['last_center'] - after['last_base_xyz'],
                          torch.full_like(after['last_center'], 0.25), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", ["2", "3"])
def test_lanes_reproduce_every_pass(cuda, monkeypatch, lanes):
    """A race between the decoder stack's lanes shows as a pass that differs from an earlier pass
    over the same batch (no dropout, no updates: tools/gf_lanes_soak.py, shortened).  It caught
    the one this code had: two head chains' backwards on two streams sharing one scratch."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BTR_GF_LANES=lanes, BTR_GRAPHS="1", BTR_GF_SLOTS="1",
               BTR_FUSED_GF_STACK="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gf_lanes_soak.py"), "40"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "40 passes, 0 with a difference" in out.stdout
