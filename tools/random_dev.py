import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T
dev = torch.device("cuda:0")
g = np.load(os.path.join(T.GOLD, "votenet_sampling.npz"))
want = g['random_grad_vote_agg_w0'].reshape(128, 259)
res = {}
for fused in ("1", "0"):
    os.environ["BTR_FUSED_SA"] = fused
    net, ep = T.run_votenet_sampling(dev, "random", pin=True)
    got = net.pnet.vote_aggregation.mlp_module.layer0.conv.weight.grad.cpu().numpy().reshape(128, 259)
    res[fused] = got
    d = np.abs(got - want)
    top = np.dstack(np.unravel_index(np.argsort(-d.ravel())[:8], d.shape))[0]
    print("fused=%s max|want| %.3e; top deviations (out ch, in col): %s" % (fused, np.abs(want).max(), [(int(a), int(b), "%.2e" % d[a, b], "%.2e" % want[a, b]) for a, b in top]))
    print("   per-input-column max dev: xyz cols %s, feature cols max %.2e" % (d[:, :3].max(0), d[:, 3:].max()))
inds = g['random_aggregated_vote_inds']
print("duplicate proposals per scene:", [256 - len(np.unique(r)) for r in inds])
