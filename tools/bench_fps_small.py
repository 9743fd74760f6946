import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from backtoreality_amd.pointnet2 import _ext
from tools.bench_ops import timeit
for (n, m) in [(64, 64), (256, 256), (512, 512), (1024, 1024), (2048, 2048), (4096, 4096), (5000, 4096), (20000, 2048), (40000, 4096)]:
    x = (torch.rand(8, n, 3, device="cuda") * 4 + 0.2).contiguous()
    med, mn = timeit(lambda: _ext.furthest_point_sampling(x, m), iters=5)
    print("n=%6d m=%5d  %8.3f ms  %.3f us/iter" % (n, m, mn, 1e3 * mn / (m - 1)))
