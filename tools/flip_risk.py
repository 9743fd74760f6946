"""How close does a BR / FSB step sit to a discontinuity of its own gradient?

The gradient of this network jumps when a ReLU input crosses 0 (or two samples of a max-pool
group swap order).  Two correct float32 implementations differ by ~1e-5 in their activations,
so an element whose pre-activation is closer than that to 0 AND that carries a large upstream
gradient can flip and move whole gradient tensors by percents (measured on MI355X: BR step,
target scenes 100/101, fp2.mlp.layer0 channel 43, point (1, 424): z = +1.6e-5 vs -1.4e-5,
upstream gradient = 48 % of the layer's largest -> grad_sa1_w0 moves by 1.4e-2).  A parity
fixture should not sit on such a point.  This tool evaluates the step in float64 on the CPU
and lists the elements with |z| < 5e-5 * std(channel) whose upstream gradient is more than 1 % of
the 2-norm of their layer's whole upstream gradient; tests/golden/make_golden.py uses seeds for which the list
is empty.     python tools/flip_risk.py br 0 100   |   python tools/flip_risk.py fsb 0
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from f64_truth import ExtF64, to64  # noqa: E402
from backtoreality_amd.pointnet2 import pointnet2_utils  # noqa: E402
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

REL_Z, EXPOSURE = 5e-5, 1e-2


def risks(kind, seeds, center_jitter=0.0):
    cfg = config.scannet_md40()
    pointnet2_utils._ext = ExtF64()
    os.environ["BTR_FUSED_LOSS"] = "0"
    dev = torch.device("cpu")
    mk = dict(center_jitter=center_jitter) if center_jitter else {}
    batches = [to64(synthetic.make_batch(s, 2, 4096, cfg, **mk)) for s in seeds]
    sampling = kind.split(":")[1] if ":" in kind else "vote_fps"
    kind = kind.split(":")[0]
    net = train.build_model(cfg, dev, seed=0, domain_adaptation=(kind == "br"),
                            center_refine=(kind == "cr"), sampling=sampling).double()
    if sampling == "random":    # the fixture's draws
        import numpy as _np
        gs = _np.load(os.path.join(ROOT, "tests", "golden", "votenet_sampling.npz"))
        real = torch.randint
        torch.randint = lambda *a, **k: torch.as_tensor(gs['random_aggregated_vote_inds'],
                                                        dtype=k.get('dtype', torch.int64))
    cap = []

    def pre(name):
        def h(mod, inp):
            z = inp[0]
            cap.append([name, z.detach().clone(), None])
            return None
        return h

    def post(name):
        def h(mod, inp, out):
            out.retain_grad()
            cap[-1][2] = out
        return h
    pools = []

    def pool_hook(name):
        def h(mod, inp, out):
            out.retain_grad()
            pools.append((name, out))
        return h
    for n, m in net.named_modules():
        if isinstance(m, nn.ReLU):
            m.register_forward_pre_hook(pre(n))
            m.register_forward_hook(post(n))
        if n.endswith("mlp_module"):      # (B, C, npoint, nsample) right before the max-pool
            m.register_forward_hook(pool_hook(n))
    ends = []
    for b in batches:
        if kind == "cr":
            e = net({'point_clouds': b['point_clouds']}, b['center_label'], b['sem_cls_label'])
        else:
            e = net({'point_clouds': b['point_clouds']})
        e.update(b)
        ends.append(e)
    if kind == "br":
        loss = loss_helper.get_loss_DA(ends[0], ends[1], cfg)[0]
    elif kind == "cr":
        loss = loss_helper.get_loss_DA_jitter(ends[0], ends[1], 30, cfg)[0]
    else:
        loss = loss_helper.get_loss(ends[0], cfg)[0]
    loss.backward()
    found = []
    for name, z, out in cap:
        if out is None or out.grad is None:
            continue
        g = out.grad
        dims = tuple(d for d in range(z.dim()) if d != 1)
        std = z.std(dim=dims, keepdim=True) + 1e-30
        near = (z.abs() < REL_Z * std)
        # exposure: what flipping this element's mask removes from / adds to the layer's
        # gradient, relative to the layer gradient's 2-norm
        expo = g.abs() / (g.norm() + 1e-300)
        hit = near & (expo > EXPOSURE)
        for idx in hit.nonzero().tolist()[:5]:
            found.append((name, tuple(idx), float(z[tuple(idx)]), float(expo[tuple(idx)])))
    # max-pool: two samples of a group within float32 noise of each other at the top, the
    # pooled gradient then goes to the one or the other point
    for name, z in pools:
        if z.grad is None or z.dim() != 4 or z.size(3) < 2:
            continue
        top = torch.topk(z.detach(), 2, dim=3).values
        gap = top[..., 0] - top[..., 1]
        std = z.detach().std(dim=(0, 2, 3), keepdim=True).squeeze(-1) + 1e-30
        g = z.grad.sum(dim=3)
        expo = g.abs() / (g.norm() + 1e-300)
        hit = (gap < REL_Z * std) & (gap > 0) & (expo > EXPOSURE)
        for idx in hit.nonzero().tolist()[:5]:
            found.append((name + " [max-pool top-2 gap]", tuple(idx), float(gap[tuple(idx)]),
                          float(expo[tuple(idx)])))
    return found


if __name__ == "__main__":
    kind = sys.argv[1]
    seeds = [int(a) for a in sys.argv[2:]]
    f = risks(kind, seeds, 0.1 if kind == "cr" else 0.0)
    print("%s seeds %s: %d risky elements" % (kind, seeds, len(f)))
    for r in f:
        print("   %-50s at %s  z = %+.2e  |upstream grad| = %.1f %% of the layer's gradient norm" % (
            r[0], r[1], r[2], 100 * r[3]))
