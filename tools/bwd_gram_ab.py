"""btr_sa_bwd_gram (the pooled layer's backward without Y_l) against btr_sa_bwd_fused on the same
layer, and the pooled layer's forward with and without the Y_l store, alone on the chip at the
benchmark shapes (dense rows; event-pair time over `reps` calls).
python tools/bwd_gram_ab.py"""
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


def case(rows, n, k, s, ps):
    dev = torch.device("cuda:0")
    r = lambda *sh: torch.randn(*sh, device=dev)
    x = r(rows, k)
    pa, pb, mup, isp = r(k), r(k), r(k), r(k).abs() + .5
    W = r(n, k) * 0.2
    Wt = W.t().contiguous()
    Y = torch.relu(pa * x + pb) @ Wt
    groups = rows // s
    arg = torch.randint(0, s, (groups, n), device=dev, dtype=torch.uint8)
    dcl, alpha, beta = r(groups, n), r(n) * .1, r(n) * .1
    chunks = max(_lib.btr_sa_bwd_fused_chunks(rows, n, k), _lib.btr_sa_bwd_gram_chunks(rows, n, k))
    dz, pw, dw = r(rows, k), r(chunks, n, k), r(n, k)
    part = r(max(chunks, 1024), 2, max(n, k))
    gs = r(int(_lib.btr_sa_bwd_gram_scratch_floats(rows, n, k)))
    m1, m2, dg, db = r(k), r(k), r(k), r(k)
    st = _ext._stream(0)

    def fused():
        _ext._call(_lib.btr_sa_bwd_fused, rows, n, k, _p(Y), n, None, None, None, None, None,
                   None, None, s, _p(arg), _p(dcl), _p(alpha), _p(beta), _p(x), k, None, _p(pa),
                   _p(pb), _p(mup), _p(isp), _p(Wt), n, _p(dz), k, _p(pw), _p(dw), _p(part),
                   _p(m1), _p(m2), _p(dg), _p(db), st)

    def gram():
        _ext._call(_lib.btr_sa_bwd_gram, rows, n, k, _p(x), k, _p(pa), _p(pb), _p(mup), _p(isp),
                   _p(W), _p(Wt), n, s, _p(arg), _p(dcl), _p(alpha), _p(beta), _p(dz), k, _p(pw),
                   _p(dw), _p(gs), _p(part), _p(m1), _p(m2), _p(dg), _p(db), st)

    fused()
    dz_f, dw_f = dz.clone(), dw.clone()
    gram()
    torch.cuda.synchronize()
    ez = float((dz - dz_f).abs().max() / dz_f.abs().max())
    ew = float((dw - dw_f).abs().max() / dw_f.abs().max())
    tf, tg = timed(fused), timed(gram)
    # forward: the pooled layer's streaming GEMM with / without the Y_l store
    grid = _lib.btr_sa_gemm_grid(rows)
    fpart = r(grid, 2, n)
    gamma = r(n)
    gext = r(rows // ps, n)
    aext = torch.zeros(rows // ps, n, device=dev, dtype=torch.uint8)
    yout = r(rows, n)

    def fwd(store):
        _ext._call(_lib.btr_sa_gemm_nt_poolfwd, rows, n, k, _p(x), k, _p(W), k,
                   _p(yout) if store else None, n, _p(pa), _p(pb), _p(fpart), ps, _p(gamma),
                   _p(gext), _p(aext), st)

    t1, t0 = timed(lambda: fwd(True)), timed(lambda: fwd(False))
    print("rows %7d n %3d k %3d s %3d: backward Y-reading %7.1f us, Gram form %7.1f us "
          "(max dev dz %.1e dw %.1e); forward with Y store %7.1f us, without %7.1f us" % (
              rows, n, k, s, tf, tg, ez, ew, t1, t0))


if __name__ == "__main__":
    # (the Gram form covers n <= 128, k <= 64 since round 6; profiles/r05_h_bwd_gram_ab.txt keeps
    # the single-role kernel's numbers at the 256 x 128 layers)
    case(706560, 128, 64, 64, 8)      # SA1 pooled layer (compact-row count, dense form)
    case(262144, 128, 64, 32, 8)
    case(65536, 64, 64, 16, 8)
