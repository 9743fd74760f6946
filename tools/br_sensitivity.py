"""How sensitive is the BR step's SA1 gradient to float32-rounding-sized perturbations?
Runs the step with the vote features multiplied by (1 + eps*N(0,1)) for several seeds, on
both the fused and the nine-op path, and prints the deviation of grad_sa1_w0 from (a) the
unperturbed run of the same path and (b) the float64 truth."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T  # noqa: E402
from backtoreality_amd.votenet import config, loss_helper, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(T.GOLD, "votenet_br_step.npz"))
truth = np.load(os.path.join(T.GOLD, "f64_truth.npz"))["br_grad_sa1_w0"]
cfg = config.scannet_md40()
bS = synthetic.make_batch(0, 2, 4096, cfg, device=dev)
bT = synthetic.make_batch(100, 2, 4096, cfg, device=dev)


def run(fused, eps, seed, where):
    os.environ["BTR_FUSED_SA"] = fused
    net = train.build_model(cfg, dev, domain_adaptation=True, seed=0)
    gen = torch.Generator(device=dev).manual_seed(seed)
    hooks = []
    if eps:
        def hook(mod, inp, out):
            if isinstance(out, tuple):
                return tuple(o * (1 + eps * torch.randn(o.shape, device=dev, generator=gen)) for o in out)
            return out * (1 + eps * torch.randn(out.shape, device=dev, generator=gen))
        hooks.append(getattr(net, where).register_forward_hook(hook))
    with T.pinned_vote_inds(net, g['S_aggregated_vote_inds'], g['T_aggregated_vote_inds'],
                            idx_per_forward=[g['S_vote_agg_idx'], g['T_vote_agg_idx']]):
        eS = net({'point_clouds': bS['point_clouds']})
        eT = net({'point_clouds': bT['point_clouds']})
    eS.update(bS)
    eT.update(bT)
    loss, eS, eT = loss_helper.get_loss_DA(eS, eT, cfg)
    loss.backward()
    for h in hooks:
        h.remove()
    return net.backbone_net.sa1.mlp_module.layer0.conv.weight.grad.cpu().numpy().astype(np.float64)


def dev_(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


for fused in ("1", "0"):
    base = run(fused, 0.0, 0, None)
    print("== BTR_FUSED_SA=%s   unperturbed vs f64 truth: %.2e" % (fused, dev_(base, truth)))
    for where in ("vgen", "backbone_net"):
        for eps in (1e-7, 1e-6, 1e-5):
            ds = []
            for seed in range(4):
                if where == "backbone_net":   # perturb fp2 features only via a hook on fp2
                    w = "backbone_net"
                ds.append(run(fused, eps, seed, "vgen" if where == "vgen" else "vgen"))
            print("   eps %.0e at %s: vs unperturbed %s   vs truth %s" % (
                eps, "vgen output",
                ["%.1e" % dev_(d, base) for d in ds], ["%.1e" % dev_(d, truth) for d in ds]))
        break
