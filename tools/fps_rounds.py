import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.pointnet2 import _ext
from tools.bench_ops import scenes, timeit
xyz = scenes(8, 40000)
ref = None
for impl in ("single", "pm"):
    if impl == "single": os.environ["BTR_FPS_IMPL"] = "single"
    else: os.environ.pop("BTR_FPS_IMPL", None)
    out = _ext.furthest_point_sampling(xyz, 2048)
    if ref is None: ref = out
    med, mn = timeit(lambda: _ext.furthest_point_sampling(xyz, 2048), iters=5)
    print(impl, "%.3f ms" % mn, "same as single:", bool(torch.equal(out, ref)))
os.environ["BTR_FPS_ROUNDS"] = "1"
_ext.furthest_point_sampling(xyz, 2048); torch.cuda.synchronize()
