"""FPS over an FPS-ordered prefix is the identity: levels 2-4 of the backbone's sampling pyramid
must return 0..m-1 in EVERY training step, on changing batches, with the other streams of the
step running beside the sampling kernels.  (This is the check that caught the register-resident
FPS kernel returning wrong sequences in 1-3 % of its launches when it was compiled with
SLP-vectorised packed-f32 code and other kernels shared the chip -- never when it ran alone, so
no op-level test saw it.  build.py compiles with -fno-slp-vectorize since; tools/
diag_pipeline_inds.py is the long-running form of this test.)"""
import pytest
import torch

from backtoreality_amd.pointnet2 import fused_backbone
from backtoreality_amd.votenet import config, synthetic, train

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pipelined", [False, True])
def test_pyramid_levels_are_the_identity_in_every_step(cuda, pipelined, monkeypatch):
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
