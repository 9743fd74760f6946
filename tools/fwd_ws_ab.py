"""Forward streaming GEMM of a set-abstraction layer: single-role kernel (BTR_FWD_WS=0) against
the producer / consumer form with two chunks (1) or one chunk (2) per workgroup, alone on the
chip.  python tools/fwd_ws_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from backtoreality_amd.pointnet2 import _ext  # noqa: E402

_lib, _p = _ext._lib, _ext._p


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(rows, n, k, ps, store=True):
    dev = torch.device("cuda:0")
    r = lambda *sh: torch.randn(*sh, device=dev)
    x, W, pa, pb, gamma = r(rows, k), r(n, k) * 0.2, r(k), r(k), r(n)
    grid = _lib.btr_sa_gemm_grid(rows)
    part = r(grid, 2, n)
    y = r(rows, n)
    st = _ext._stream(0)
    if ps:
        gext = r(rows // ps, n)
        aext = torch.zeros(rows // ps, n, device=dev, dtype=torch.uint8)

    def fwd():
        if ps:
            _ext._call(_lib.btr_sa_gemm_nt_poolfwd, rows, n, k, _p(x), k, _p(W), k,
                       _p(y) if store else None, n, _p(pa), _p(pb), _p(part), ps, _p(gamma),
                       _p(gext), _p(aext), st)
        else:
            _ext._call(_lib.btr_sa_gemm_nt, rows, n, k, _p(x), k, _p(W), k, _p(y), n, _p(pa),
                       _p(pb), _p(part), st)
    out = []
    for mode in ("0", "1", "2"):
        os.environ["BTR_FWD_WS"] = mode
        out.append(timed(fwd))
    print("rows %7d n %3d k %3d ps %2d store %d: single-role %7.1f us, ws (2 chunks/wg) %7.1f us, "
          "ws (1 chunk/wg) %7.1f us" % (rows, n, k, ps, store, *out))


if __name__ == "__main__":
    case(706560, 128, 64, 8)
    case(706560, 128, 64, 8, store=False)
    case(706560, 64, 64, 0)
    case(114688, 128, 128, 0)
    case(114688, 256, 128, 8)
    case(65536, 256, 128, 16)
    case(32768, 128, 128, 0)
