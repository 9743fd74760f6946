"""Inference-mode forward: fused path and nine-op path against the fixture (GPU)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(T.GOLD, "votenet_eval.npz"))
res = {}
for fused in ("1", "0"):
    os.environ["BTR_FUSED_SA"] = fused
    net, ep = T.run_votenet_eval(dev, pin=True)
    res[fused] = ep
    print("== BTR_FUSED_SA=%s" % fused)
    for k in ('sa1_features', 'sa2_features', 'sa3_features', 'sa4_features', 'fp2_features',
              'aggregated_vote_features'):
        a = ep[k].detach().cpu().numpy().astype(np.float32).ravel()[::37]
        w = g[k + '_sample']
        print("   %-26s max abs diff %.3e (max |want| %.3e)" % (k, np.abs(a - w).max(), np.abs(w).max()))
    for n, m in net.named_modules():
        if n.endswith("sa1.mlp_module.layer0.bn.bn") or n.endswith("sa1.mlp_module.layer1.bn.bn"):
            print("   ", n, "rm[:3]", m.running_mean[:3].tolist(), "rv[:3]", m.running_var[:3].tolist(),
                  "momentum", m.momentum, "training", m.training)
for k in ('sa1_features', 'sa2_features'):
    a, b = res["1"][k], res["0"][k]
    print("fused vs nine-op", k, float((a - b).abs().max()), float(b.abs().max()))
