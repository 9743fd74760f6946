import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.pointnet2 import _ext
from tools.bench_ops import scenes, timeit
xyz = scenes(8, 40000)
os.environ["BTR_FPS_IMPL"] = "single"
ref = _ext.furthest_point_sampling(xyz, 2048)
os.environ.pop("BTR_FPS_IMPL")
for cfg in ["1,2", "2,2", "1,4", "2,4", "4,4", "1,8", "2,8", "4,8", "2,16", "4,16"]:
    os.environ["BTR_FPS_MULTI"] = cfg
    out = _ext.furthest_point_sampling(xyz, 2048)
    med, mn = timeit(lambda: _ext.furthest_point_sampling(xyz, 2048), iters=5)
    print("UB,KMAX=%-5s %.3f ms same=%s" % (cfg, mn, bool(torch.equal(out, ref))))
