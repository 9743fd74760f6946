#!/usr/bin/env python3
"""tests/test_backbone_gpu.py's comparison (layer-by-layer run, then whole-backbone run, same
cloud) repeated: hunts the intermittent sa2_inds corruption seen once in the full suite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_backbone_gpu as T
from backtoreality_amd.votenet import backbone_module
dev = torch.device("cuda:0")
bad = 0
iters = int(os.environ.get("ITERS", "40"))
for it in range(iters):
    torch.manual_seed(3)
    net = backbone_module.Pointnet2Backbone(input_feature_dim=1).to(dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    pc = torch.rand((3, 20000, 4), generator=g)
    pc[..., :3] = pc[..., :3] * torch.tensor([6.0, 5.0, 2.5])
    pc = pc.to(dev)
    ref = T._run(False, net, pc, False, extra_grads=True)
    got = T._run(True, net, pc, False, extra_grads=True)
    for name, res in (("layerwise", ref), ("native", got)):
        for key in ("sa1_inds", "sa2_inds"):
            if key == "sa1_inds":
                ok = torch.equal(ref[0][key], got[0][key])
            else:
                inds = res[0][key]
                ok = torch.equal(inds, torch.arange(inds.shape[1], device=dev,
                                                    dtype=inds.dtype).expand_as(inds))
            if not ok:
                bad += 1
                print("iter", it, name, key, "WRONG")
                if key == "sa2_inds":
                    want = torch.arange(inds.shape[1], device=dev, dtype=inds.dtype).expand_as(inds)
                    d = (inds != want).nonzero()
                    print("  mismatches", d.shape[0], "first", d[:3].tolist(), "last", d[-2:].tolist())
                    b0, c0 = int(d[0, 0]), int(d[0, 1])
                    print("  values from first mismatch:", inds[b0, c0:c0 + 24].tolist())
                    sx = res[0]["sa2_xyz"]
                    print("  sa2_xyz equal to reference run:", bool(torch.equal(sx, ref[0]["sa2_xyz"])),
                          " sa1_xyz equal:", bool(torch.equal(res[0]["sa1_xyz"], ref[0]["sa1_xyz"])))
                    for kk in ("sa3_inds", "sa2_features", "sa3_xyz", "fp2_features"):
                        if kk in res[0]:
                            print("  ", kk, "equal:", bool(torch.equal(res[0][kk], ref[0][kk])))
print("bad", bad, "of", iters)
