"""Which fused-SA option moves the BR step's SA1 first-layer gradient away from the fixture?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(T.GOLD, "votenet_br_step.npz"))
want = g['grad_sa1_w0']
base = {"BTR_FUSED_SA": "1", "BTR_SA_RECOMPUTE": "1", "BTR_POOLGRAD": "1", "BTR_POOL_EPILOGUE": "1"}
grads = {}
for name, env in (("fused", {}), ("fused again", {}), ("no recompute", {"BTR_SA_RECOMPUTE": "0"}),
                  ("no poolgrad", {"BTR_POOLGRAD": "0"}), ("no pool epilogue", {"BTR_POOL_EPILOGUE": "0"}),
                  ("all off", {"BTR_SA_RECOMPUTE": "0", "BTR_POOLGRAD": "0", "BTR_POOL_EPILOGUE": "0"}),
                  ("unfused", {"BTR_FUSED_SA": "0"}), ("unfused again", {"BTR_FUSED_SA": "0"})):
    os.environ.update(base)
    os.environ.update(env)
    net, sig, loss, eS, eT = T.run_votenet_br(dev, pin=True)
    got = net.backbone_net.sa1.mlp_module.layer0.conv.weight.grad.cpu().numpy()
    grads[name] = got
    err = np.abs(got - want).max() / np.abs(want).max()
    l2 = np.linalg.norm(got - want) / np.linalg.norm(want)
    print("%-18s loss %.6f (fixture %.6f)  grad_sa1_w0 max-norm %.3e  rel L2 %.3e" % (
        name, float(loss), float(g['loss']), err, l2))
    for other in ('grad_global_netD2_w', 'grad_local_netD_last_w'):
        pass
print("want[:4] =", want.reshape(64, -1)[:4])
print("fused[:4] =", grads["fused"].reshape(64, -1)[:4])
print("unfused[:4] =", grads["unfused"].reshape(64, -1)[:4])
d = np.abs(grads["fused"] - want).reshape(64, -1)
print("worst rows:", np.argsort(-d.max(1))[:5], d.max(1)[np.argsort(-d.max(1))[:5]], "max|want| =", np.abs(want).max())
