#!/usr/bin/env python3
"""Large-scene FPS alone on the chip with its running min-dists in LDS (default), partly in LDS
and in global memory (round 5's form): milliseconds per call and cycles per dependent step.
Usage: fps_lds_ab.py [points [batch [samples]]]; the results are identical by construction
(tests/test_ops_gpu.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from tools.bench_ops import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
xyz = scenes(b, n)
ref = None
for kb in (-1, 128, 64, 0, -1):
    _ext.set_fps_lds_kb(kb)
    for _ in range(3):
        out = _ext.furthest_point_sampling(xyz, m)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = _ext.furthest_point_sampling(xyz, m)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    if ref is None:
        ref = out.clone()
    assert torch.equal(out, ref)
    eff = _ext._idx.btr_fps_lds_kb(n) if kb < 0 else kb
    print("lds %4d KB (%6d of %d min-dists in LDS): call (sort + kernel) median %.3f ms, min %.3f ms"
          " = %.0f cycles per dependent step at 2.4 GHz" % (
              eff, min(n, eff * 256 // 64 * 64), n, ts[len(ts) // 2], ts[0],
              ts[0] * 1e-3 / (m - 1) * 2.4e9))
_ext.set_fps_lds_kb(-1)
