"""Small-M NT GEMM (gemm_nt_sm_kernel on pre-split weight planes) against gemm_nt_kernel on the
shapes of the point-wise chains and of GroupFree3D's 1 024-row layers, alone on the chip.
python tools/gemm_sm_ab.py"""
import os
import sys

os.environ.setdefault("BTR_PM_SM_ROWS", "16384")   # the kernel beyond its default row limit too

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from backtoreality_amd.pointnet2 import _ext  # noqa: E402

_lib, _p = _ext._lib, _ext._p


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(rows, n, k, pro, stats):
    dev = torch.device("cuda:0")
    r = lambda *sh: torch.randn(*sh, device=dev)
    A, W = r(rows, k), (r(n, k) * 0.3).contiguous()
    pa, pb = (r(k), r(k)) if pro else (None, None)
    grid = _lib.btr_pm_gemm_grid(rows)
    part = r(grid, 2, n) if stats else None
    C = r(rows, n)
    planes = torch.empty((int(_lib.btr_pm_weight_planes_bytes(n, k)),), dtype=torch.uint8, device=dev)
    st = _ext._stream(0)
    _ext._call(_lib.btr_pm_weight_planes, n, k, _p(W), k, _p(planes), st)

    def new():
        _ext._call(_lib.btr_pm_gemm_nt_sm, rows, n, k, _p(A), k, _p(planes), _p(C), n, _p(pa),
                   _p(pb), _p(part), None, st)

    def old():
        _ext._call(_lib.btr_pm_gemm_nt, rows, n, k, _p(A), k, _p(W), k, _p(C), n, _p(pa), _p(pb),
                   _p(part), None, st)

    tn, to = timed(new), timed(old)
    gf = 2.0 * rows * n * k / 1e9
    print("rows %6d n %3d k %4d pro %d stats %d: small-M %6.1f us (%5.1f TF), 64-row tiles %6.1f us "
          "(%5.1f TF)" % (rows, n, k, pro, stats, tn, gf / tn * 1e3, to, gf / to * 1e3))


if __name__ == "__main__":
    case(8192, 256, 256, 1, 1)
    case(8192, 256, 512, 0, 1)
    case(4096, 256, 512, 0, 1)
    case(4096, 256, 256, 1, 1)
    case(2048, 128, 128, 1, 1)
    case(1024, 288, 288, 1, 1)
    case(1024, 288, 288, 0, 0)
    case(16384, 256, 256, 1, 1)
    case(8192, 256, 256, 0, 0)
