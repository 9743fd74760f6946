"""FPS over an FPS-ordered prefix is the identity: levels 2-4 of the backbone's sampling pyramid
must return 0..m-1 in EVERY training step, on changing batches, with the other streams of the
step running beside the sampling kernels.  (This is the check that caught the register-resident
FPS kernel returning wrong sequences in 1-3 % of its launches when it was compiled with
SLP-vectorised packed-f32 code and other kernels shared the chip -- never when it ran alone, so
no op-level test saw it.  build.py compiles with -fno-slp-vectorize since; tools/
diag_pipeline_inds.py is the long-running form of this test.)

Round 4: levels 2-4 normally go through the parallel check of exactly this identity
(csrc/sampling.hip fps_prefix_*_kernel) and the serial kernel only starts when it fails.  The
test therefore runs in both modes -- `prefix`: the check kernels beside the step's other
streams (a false alarm would only cost time, a false confirmation would be a wrong index: the
serial kernel is run afterwards on the same points as the judge) and `serial`
(BTR_FPS_PREFIX=0): the register-resident kernel itself, as before -- and under the loops whose
co-runners differ: the Back-to-Reality step (two forwards, the target pyramid under the source
forward) and the GroupFree3D step."""
import pytest
import torch

from backtoreality_amd.pointnet2 import fused_backbone
from backtoreality_amd.votenet import config, synthetic, train

pytestmark = pytest.mark.gpu


def _spy_on_handles(monkeypatch, seen):
    orig = fused_backbone.FusedBackboneFn.apply

    def spy(cloud, handle, entry, *params):
        seen.append((list(handle.inds), [x for x in handle.xyz]))
        return orig(cloud, handle, entry, *params)

    monkeypatch.setattr(fused_backbone.FusedBackboneFn, "apply", staticmethod(spy))


def _count_wrong(seen, cuda, judge_serial, monkeypatch):
    """Levels 2-4 of every recorded pyramid: not the identity -> wrong.  `judge_serial`: the
    serial kernel, alone on the chip, must agree on the same input points."""
    from backtoreality_amd.pointnet2 import _ext
    wrong = total = 0
    for inds_l, xyz_l in seen:
        for level in (1, 2, 3):
            inds = inds_l[level]
            want = torch.arange(inds.shape[1], device=cuda, dtype=inds.dtype).expand_as(inds)
            total += 1
            wrong += int(not torch.equal(inds, want))
    if judge_serial and seen:
        monkeypatch.setenv("BTR_FPS_PREFIX", "0")
        inds_l, xyz_l = seen[-1]
        for level in (1, 2, 3):
            again = _ext.furthest_point_sampling(xyz_l[level - 1].contiguous(),
                                                 inds_l[level].shape[1])
            assert torch.equal(again, inds_l[level])
    return wrong, total


@pytest.mark.parametrize("mode", ["prefix", "serial"])
def test_pyramid_levels_are_the_identity_in_every_back_to_reality_step(cuda, mode, monkeypatch):
    """train_Votenet_BR.py:267-289: two forwards per step; the target pyramid runs on the side
    stream under the source forward, the next step's two pyramids under this step's backward."""
    monkeypatch.setenv("BTR_FPS_PREFIX", "1" if mode == "prefix" else "0")
    cfg = config.scannet_md40()
    net = train.build_model(cfg, cuda, seed=0, domain_adaptation=True)
    opt = train.make_optimizer(net)
    bs = [synthetic.make_batch(11 * s, 4, 20000, cfg, device=cuda) for s in range(3)]
    bt = [synthetic.make_batch(100000 + 13 * s, 4, 20000, cfg, device=cuda) for s in range(3)]
    seen = []
    _spy_on_handles(monkeypatch, seen)
    samp_s = net.backbone_net.prefetch_sampling(bs[0]['point_clouds'])
    samp_t = None
    steps = 80
    for it in range(steps):
        out = train.train_step_br(net, opt, bs[it % 3], bt[it % 3], cfg, sampling_S=samp_s,
                                  sampling_T=samp_t, next_batch_S=bs[(it + 1) % 3],
                                  next_batch_T=bt[(it + 1) % 3])
        samp_s, samp_t = out[1]['next_sampling'], out[2]['next_sampling']
        torch.cuda.synchronize()
    assert len(seen) == 2 * steps
    wrong, total = _count_wrong(seen, cuda, mode == "prefix", monkeypatch)
    assert wrong == 0, "%d of %d level evaluations were not the identity" % (wrong, total)


@pytest.mark.parametrize("mode", ["prefix", "serial"])
def test_pyramid_levels_are_the_identity_in_every_groupfree_step(cuda, mode, monkeypatch):
    """train_GF_FSB.py:287-322: the decoder's attention / chain kernels are the co-runners."""
    from backtoreality_amd.groupfree import train as gf_train
    monkeypatch.setenv("BTR_FPS_PREFIX", "1" if mode == "prefix" else "0")
    cfg = config.scannet_md40()
    net = gf_train.build_model(cfg, cuda)
    opt = gf_train.make_optimizer(net)
    batches = [synthetic.make_batch(17 * s, 2, 20000, cfg, device=cuda, use_height=False)
               for s in range(3)]
    seen = []
    _spy_on_handles(monkeypatch, seen)
    sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
    steps = 60
    for it in range(steps):
        loss, end = gf_train.train_step(net, opt, batches[it % 3], cfg, sampling=sampling,
                                        next_batch=batches[(it + 1) % 3])
        sampling = end['next_sampling']
        torch.cuda.synchronize()
    assert len(seen) == steps
    wrong, total = _count_wrong(seen, cuda, mode == "prefix", monkeypatch)
    assert wrong == 0, "%d of %d level evaluations were not the identity" % (wrong, total)


@pytest.mark.parametrize("mode", ["prefix", "serial"])
@pytest.mark.parametrize("pipelined", [False, True])
def test_pyramid_levels_are_the_identity_in_every_step(cuda, pipelined, mode, monkeypatch):
    monkeypatch.setenv("BTR_FPS_PREFIX", "1" if mode == "prefix" else "0")
    cfg = config.scannet_md40()
    net = train.build_model(cfg, cuda, seed=0)
    opt = train.make_optimizer(net)
    batches = [synthetic.make_batch(7 * s, 4, 20000, cfg, device=cuda) for s in range(4)]
    seen = {}
    orig = fused_backbone.FusedBackboneFn.apply

    def spy(cloud, handle, entry, *params):
        seen["inds"] = list(handle.inds)
        return orig(cloud, handle, entry, *params)

    monkeypatch.setattr(fused_backbone.FusedBackboneFn, "apply", staticmethod(spy))
    sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds']) if pipelined \
        else None
    wrong = 0
    steps = 200
    for it in range(steps):
        b = batches[it % 4]
        if pipelined:
            loss, end = train.train_step(net, opt, b, cfg, sampling=sampling,
                                         next_batch=batches[(it + 1) % 4])
            sampling = end['next_sampling']
        else:
            loss, end = train.train_step(net, opt, b, cfg)
        torch.cuda.synchronize()
        for level in (1, 2, 3):
            inds = seen["inds"][level]
            want = torch.arange(inds.shape[1], device=cuda, dtype=inds.dtype).expand_as(inds)
            wrong += int(not torch.equal(inds, want))
    assert wrong == 0, "%d of %d level evaluations were not the identity" % (wrong, 3 * steps)
