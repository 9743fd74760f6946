"""s_memtime phase counters of the large-scene FPS kernel (BTR_FPS_PROF=1: the PROF build of the
same source; the instrumentation costs ~11 % of the wave cycles), with the running min-dists in
LDS (the default where the scene fits) and in global memory (the form of rounds 2 - 5)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import scenes  # noqa: E402

xyz = scenes(8, 40000)
_ext.furthest_point_sampling(xyz, 2048)
torch.cuda.synchronize()
os.environ["BTR_FPS_PROF"] = "1"
for kb, name in ((-1, "min-dists in LDS"), (0, "min-dists in global memory")):
    _ext.set_fps_lds_kb(kb)
    sys.stderr.write("== %s\n" % name)
    sys.stderr.flush()
    _ext.furthest_point_sampling(xyz, 2048)
    torch.cuda.synchronize()
_ext.set_fps_lds_kb(-1)
