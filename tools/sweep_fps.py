import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from backtoreality_amd.pointnet2 import _ext
from backtoreality_amd.votenet import synthetic
from tools.bench_ops import timeit, scenes
xyz = scenes(8, 40000)
ref = None
for cfg in ["16,1", "16,2", "8,1", "8,2", "8,4", "4,2", "4,4"]:
    os.environ["BTR_FPS_CFG"] = cfg
    out = _ext.furthest_point_sampling(xyz, 2048)
    if ref is None: ref = out
    ok = bool(torch.equal(out, ref))
    med, mn = timeit(lambda: _ext.furthest_point_sampling(xyz, 2048), iters=5)
    print("cfg %-5s  %7.3f ms  %.3f us/iter  same=%s" % (cfg, mn, 1e3 * mn / 2047, ok))
