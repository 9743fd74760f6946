"""train.FastAdam / FastAdamW on the library's one-launch update (csrc/optimizer.hip) against
torch.optim.Adam / AdamW (fused): parameters and both moments after several steps, with two
parameter groups, ragged tensor sizes (scalar-path tensors beside 16-byte-path ones), the folded
gradient clipping, a learning-rate change EVERY step (no device-table rebuild), steps taken by
other kernels in between (HIP-graph replays, BTR_ADAM_KERNEL toggled), parameters whose step
counts differ, and the state_dict round trip."""
import copy

import pytest
import torch

from backtoreality_amd.votenet.train import FastAdam, FastAdamW

pytestmark = pytest.mark.gpu


SHAPES = [(290, 37), (290,), (3, 290), (3,), (1,), (700, 512), (700,)]


def _params(dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in SHAPES]


@pytest.mark.parametrize("decoupled", [False, True])
@pytest.mark.parametrize("clip", [None, 0.5])
def test_library_step_equals_torch_fused(cuda, decoupled, clip, monkeypatch):
    pa, pb = _params(cuda, 0), _params(cuda, 0)
    groups = lambda ps: [{"params": ps[:2] + ps[4:5]}, {"params": ps[2:4] + ps[5:], "lr": 3e-4}]
    fast_cls, ref_cls = (FastAdamW, torch.optim.AdamW) if decoupled else (FastAdam, torch.optim.Adam)
    opt_a = fast_cls(groups(pa), lr=2e-3, weight_decay=0.01, fused=True)
    opt_b = ref_cls(groups(pb), lr=2e-3, weight_decay=0.01, fused=True)
    used = []
    real = opt_a._library_step
    monkeypatch.setattr(opt_a, "_library_step", lambda *a: used.append(real(*a)) or used[-1])
    g = torch.Generator(device="cpu").manual_seed(7)

    def give_grads():
        # the SAME gradients to both (the comparison is of the update arithmetic, not of two
        # trajectories that amplify its rounding)
        for a, b in zip(pa, pb):
            gr = (torch.randn(*a.shape, generator=g) * 3).to(cuda)
            a.grad, b.grad = gr.clone(), gr.clone()

    tables = set()
    for i in range(6):
        # a per-iteration scheduler (train_GF_FSB.py:322): lr rides in the kernel arguments, the
        # device table is built once
        for opt in (opt_a, opt_b):
            opt.param_groups[0]['lr'] = 2e-3 * (0.9 ** i) if i < 4 else 1e-3
            opt.param_groups[1]['weight_decay'] = 0.01 + 0.001 * i
        give_grads()
        if clip is None:
            opt_a.step()
        else:
            total = opt_a.step(clip_norm=clip)
            ref = torch.nn.utils.clip_grad_norm_(pb, clip)
            assert torch.allclose(total, ref, rtol=1e-6)
        opt_b.step()
        if getattr(opt_a, '_btr_lib', None) is not None:
            tables.add(id(opt_a._btr_lib['items']))
    # steps 2.. ran on btr_adam_multi (step 1 builds the state)
    assert used and all(u is not None for u in used), used
    assert len(tables) == 1, "the device table was rebuilt by a learning-rate change"
    for n, (a, b) in enumerate(zip(pa, pb)):
        # (a few ulps of the parameter: the two kernels round p - update in their own order)
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-6), (n, float((a - b).abs().max()))
        sa, sb = opt_a.state[a], opt_b.state[b]
        assert torch.allclose(sa['exp_avg'], sb['exp_avg'], rtol=2e-6, atol=1e-6), n
        assert torch.allclose(sa['exp_avg_sq'], sb['exp_avg_sq'], rtol=2e-6, atol=1e-6), n
        assert float(sa['step']) == float(sb['step']) == 6.0
    # state_dict round trip: the reloaded optimizer continues identically
    sd = copy.deepcopy(opt_a.state_dict())
    opt_c = fast_cls(groups(pa), lr=2e-3, weight_decay=0.01, fused=True)
    opt_c.load_state_dict(sd)
    opt_c.param_groups[0]['lr'] = 1e-3
    give_grads()
    opt_c.step()
    opt_b.step()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=3e-6, atol=1.5e-6)


def test_misaligned_gradients_fall_back_for_that_step_only(cuda, monkeypatch):
    """A gradient the 16-byte loads cannot take (a view at an odd offset) sends ONE step through
    torch's kernels; the device table and the step count survive, the next step is the
    library's again and the trajectory stays that of torch.optim.Adam."""
    pa, pb = _params(cuda, 1), _params(cuda, 1)
    opt_a = FastAdam(pa, lr=1e-3, fused=True)
    opt_b = torch.optim.Adam(pb, lr=1e-3, fused=True)
    used = []
    real = opt_a._library_step
    monkeypatch.setattr(opt_a, "_library_step", lambda *a: used.append(real(*a)) or used[-1])
    g = torch.Generator(device="cpu").manual_seed(3)
    for i in range(5):
        for a, b in zip(pa, pb):
            gr = torch.randn(*a.shape, generator=g).to(cuda)
            if i == 2:    # an odd-offset view for every tensor
                buf = torch.empty(gr.numel() + 1, device=cuda)
                buf[1:] = gr.flatten()
                a.grad = buf[1:].view_as(gr)
            else:
                a.grad = gr.clone()
            b.grad = gr.clone()
        opt_a.step()
        opt_b.step()
    # (step 1 never reaches the library path; it answers (total norm | None,) or None)
    assert [u is not None for u in used] == [True, False, True, True], used
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-6)
        assert float(opt_a.state[a]['step']) == 5.0


def test_steps_taken_elsewhere_do_not_stale_the_bias_correction(cuda, monkeypatch):
    """Advisor finding (round 3): the library step took Adam's bias-correction count from a host
    mirror that went stale whenever a step bypassed it.  Sequence: eager library steps, then
    steps through torch's kernels (BTR_ADAM_KERNEL=0: what a HIP-graph replay of the capturable
    optimizer or a fallback does to the DEVICE step tensors), then library steps again -- the
    trajectory must stay torch.optim.Adam's."""
    pa, pb = _params(cuda, 2), _params(cuda, 2)
    opt_a = FastAdam(pa, lr=1e-2, fused=True)
    opt_b = torch.optim.Adam(pb, lr=1e-2, fused=True)
    used = []
    real = opt_a._library_step
    monkeypatch.setattr(opt_a, "_library_step", lambda *a: used.append(real(*a)) or used[-1])
    g = torch.Generator(device="cpu").manual_seed(5)
    for i in range(9):
        monkeypatch.setenv("BTR_ADAM_KERNEL", "0" if 3 <= i < 6 else "1")
        for a, b in zip(pa, pb):
            gr = torch.randn(*a.shape, generator=g).to(cuda)
            a.grad, b.grad = gr.clone(), gr.clone()
        opt_a.step()
        opt_b.step()
    assert [u is not None for u in used] == [True, True, False, False, False, True, True, True]
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-6), float((a - b).abs().max())
        assert float(opt_a.state[a]['step']) == 9.0


def test_graph_replays_between_eager_steps(cuda):
    """capturable FastAdam (votenet.make_optimizer(capturable=True), GraphedPipelinedStep):
    eager step(s), HIP-graph replays of step(), eager again, against stock Adam stepping the
    same gradients eagerly.  The replays advance the device step tensors only."""
    pa, pb = _params(cuda, 3), _params(cuda, 3)
    opt_a = FastAdam(pa, lr=1e-2, fused=True, capturable=True)
    opt_b = torch.optim.Adam(pb, lr=1e-2, fused=True, capturable=True)
    g = torch.Generator(device="cpu").manual_seed(11)
    static = [torch.zeros_like(p) for p in pa]
    for a, s in zip(pa, static):
        a.grad = s            # static gradient buffers, as a captured step has

    def feed():
        for s, b in zip(static, pb):
            gr = torch.randn(*s.shape, generator=g).to(cuda)
            s.copy_(gr)
            b.grad = gr.clone()

    for _ in range(2):
        feed(); opt_a.step(); opt_b.step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        feed(); opt_a.step(); opt_b.step()      # warm-up on the capture stream
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        feed()
        with torch.cuda.graph(graph, stream=side):
            opt_a.step()
        # (the capture does not execute; the first replay applies this feed)
        graph.replay(); opt_b.step()
        for _ in range(3):
            feed(); graph.replay(); opt_b.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for _ in range(3):
        feed(); opt_a.step(); opt_b.step()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=3e-6, atol=2e-6), float((a - b).abs().max())
        assert float(opt_a.state[a]['step']) == float(opt_b.state[b]['step']) == 10.0


def test_parameters_with_different_step_counts(cuda):
    """torch keeps a step per parameter: one that had grad=None on earlier steps is behind.  The
    library kernel reads each tensor's own counter."""
    pa, pb = _params(cuda, 4), _params(cuda, 4)
    opt_a = FastAdam(pa, lr=1e-2, fused=True)
    opt_b = torch.optim.Adam(pb, lr=1e-2, fused=True)
    g = torch.Generator(device="cpu").manual_seed(13)
    for i in range(7):
        for n, (a, b) in enumerate(zip(pa, pb)):
            if n in (1, 5) and i < 3:     # joins at step 4
                a.grad = b.grad = None
                continue
            gr = torch.randn(*a.shape, generator=g).to(cuda)
            a.grad, b.grad = gr.clone(), gr.clone()
        opt_a.step()
        opt_b.step()
    for n, (a, b) in enumerate(zip(pa, pb)):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-6), (n, float((a - b).abs().max()))
        assert float(opt_a.state[a]['step']) == float(opt_b.state[b]['step']) == \
            (4.0 if n in (1, 5) else 7.0)
