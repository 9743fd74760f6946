"""A float64 evaluation of the VoteNet / Back-to-Reality step ON THE GPU, as the yardstick for
float32 gradient comparisons at full size (tests/test_configs_gpu.py, tools/diag_c3_f64.py).

Not an oracle of the reference's arithmetic (that is oracle/ and the golden fixtures) and not a
product path: it runs this package's reference-shaped Python layers (nine-op path, every fused
path off) with float64 parameters and inputs; the nine ops are replaced by `ExtF64Gpu` -- indices
from the library's own float32 index ops on the float32 casts of the coordinates (bit-exact
against the oracle, tests/test_ops_gpu.py), copies / blends / scatter-adds as float64 torch ops.
tools/f64_truth.py is the CPU twin that produced tests/golden/f64_truth.npz."""
import os

import torch

from backtoreality_amd.pointnet2 import _ext, pointnet2_utils
from backtoreality_amd.votenet import loss_helper, train

STAGES = ('backbone_net.sa1.', 'backbone_net.sa2.', 'backbone_net.sa3.', 'backbone_net.sa4.',
          'backbone_net.fp1.', 'backbone_net.fp2.', 'vgen.', 'pnet.vote_aggregation.')


class ExtF64Gpu(object):
    """The nine `_ext` callables for float64 CUDA tensors."""

    def furthest_point_sampling(self, points, n):
        return _ext.furthest_point_sampling(points.detach().float().contiguous(), int(n))

    def gather_points(self, points, idx):
        return torch.gather(points, 2, idx.long().unsqueeze(1).expand(-1, points.size(1), -1))

    def gather_points_grad(self, grad_out, idx, n):
        out = grad_out.new_zeros(grad_out.size(0), grad_out.size(1), n)
        return out.scatter_add_(2, idx.long().unsqueeze(1).expand(-1, grad_out.size(1), -1),
                                grad_out)

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return _ext.ball_query(new_xyz.detach().float().contiguous(),
                               xyz.detach().float().contiguous(), float(radius), int(nsample))

    def group_points(self, points, idx):
        B, C, N = points.shape
        _, M, S = idx.shape
        flat = idx.long().reshape(B, 1, M * S).expand(-1, C, -1)
        return torch.gather(points, 2, flat).reshape(B, C, M, S).clone()

    def group_points_grad(self, grad_out, idx, n):
        B, C, M, S = grad_out.shape
        out = grad_out.new_zeros(B, C, n)
        flat = idx.long().reshape(B, 1, M * S).expand(-1, C, -1)
        return out.scatter_add_(2, flat, grad_out.reshape(B, C, M * S))

    def three_nn(self, unknown, known):
        _, idx = _ext.three_nn(unknown.detach().float().contiguous(),
                               known.detach().float().contiguous())
        B, n, _ = unknown.shape
        nb = torch.gather(known.unsqueeze(1).expand(-1, n, -1, -1), 2,
                          idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
        return [((unknown.unsqueeze(2) - nb) ** 2).sum(-1), idx]

    def three_interpolate(self, points, idx, weight):
        B, C, m = points.shape
        n = idx.size(1)
        flat = idx.long().reshape(B, 1, n * 3).expand(-1, C, -1)
        return (torch.gather(points, 2, flat).reshape(B, C, n, 3) * weight.unsqueeze(1)).sum(-1)

    def three_interpolate_grad(self, grad_out, idx, weight, m):
        B, C, n = grad_out.shape
        out = grad_out.new_zeros(B, C, m)
        flat = idx.long().reshape(B, 1, n * 3).expand(-1, C, -1)
        return out.scatter_add_(2, flat, (grad_out.unsqueeze(-1) *
                                          weight.unsqueeze(1)).reshape(B, C, n * 3))


_FUSED_SWITCHES = ("BTR_FUSED_SA", "BTR_FUSED_MLP", "BTR_FUSED_LOSS", "BTR_FUSED_VOTES")


def br_step(cfg, batch_S, batch_T, dev, fused, vote_inds=None, vote_idx=None, float64=False):
    """One Back-to-Reality forward pair + get_loss_DA + backward (train_Votenet_BR.py:267-289).
    fused: the HIP path (float32) or the nine-op + torch composition; float64: the latter in
    float64 over ExtF64Gpu.  vote_inds / vote_idx: (source, target) proposals / neighbour lists of
    the two vote-aggregation calls (both ops sit downstream of computed floats).  Returns (loss,
    end_points S, end_points T, gradients by parameter name, the neighbour lists the run's own
    ball queries produced)."""
    assert not (fused and float64)
    env = {k: os.environ.get(k) for k in _FUSED_SWITCHES}
    saved_ext = pointnet2_utils._ext
    try:
        os.environ["BTR_FUSED_SA"] = "1" if fused else "0"
        if float64:
            for k in _FUSED_SWITCHES:
                os.environ[k] = "0"
            pointnet2_utils._ext = ExtF64Gpu()
        net = train.build_model(cfg, dev, seed=0, domain_adaptation=True)
        if float64:
            net = net.double()
            batch_S = {k: (v.double() if v.dtype == torch.float32 else v)
                       for k, v in batch_S.items()}
            batch_T = {k: (v.double() if v.dtype == torch.float32 else v)
                       for k, v in batch_T.items()}
        sa = net.pnet.vote_aggregation
        own = sa.forward
        queue = list(vote_inds) if vote_inds is not None else None
        idx_queue = list(vote_idx) if vote_idx is not None else None
        made = []

        def forward(xyz, features=None, inds=None):
            real_bq = pointnet2_utils.ball_query

            def bq(radius, nsample, xyz_, new_xyz_):
                made.append(real_bq(radius, nsample, xyz_, new_xyz_))
                return idx_queue.pop(0) if idx_queue is not None else made[-1]
            pointnet2_utils.ball_query = bq
            try:
                return own(xyz, features, queue.pop(0) if queue is not None else inds)
            finally:
                pointnet2_utils.ball_query = real_bq
        sa.forward = forward
        eS = net({'point_clouds': batch_S['point_clouds']})
        eT = net({'point_clouds': batch_T['point_clouds']})
        eS.update(batch_S)
        eT.update(batch_T)
        loss, eS, eT = loss_helper.get_loss_DA(eS, eT, cfg)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()
                 if p.grad is not None}
        return loss.detach(), eS, eT, grads, made
    finally:
        pointnet2_utils._ext = saved_ext
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def stage_of(name):
    for i, prefix in enumerate(STAGES):
        if name.startswith(prefix):
            return i
    return len(STAGES)


def errors_vs_f64(g_f, g_u, g64):
    """Per module, in BACKWARD order (heads first): worst relative L2 over its live parameters of
    (fused - f64, nine-op - f64, fused - nine-op) and the parameter of the first."""
    gmax = max(float(g.abs().max()) for g in g64.values())
    live = [n for n in g64 if float(g64[n].abs().max()) > 1e-4 * gmax]

    def l2(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))
    by = {}
    for n in live:
        st = STAGES[stage_of(n)] if stage_of(n) < len(STAGES) else "heads"
        eh, er, efu = l2(g_f[n], g64[n]), l2(g_u[n], g64[n]), l2(g_f[n], g_u[n])
        cur = by.get(st)
        if cur is None:
            by[st] = [eh, er, efu, n]
        else:
            if eh > cur[0]:
                cur[0], cur[3] = eh, n
            cur[1] = max(cur[1], er)
            cur[2] = max(cur[2], efu)
    order = ["heads"] + list(reversed(STAGES))
    return {k: tuple(by[k]) for k in order if k in by}
