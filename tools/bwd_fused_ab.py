"""btr_sa_bwd_fused against the calls it replaces, alone on the chip, at the SA1 / SA2 shapes of
the benchmark step (dense rows; event-pair time over `reps` calls).
python tools/bwd_fused_ab.py"""
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


def case(rows, n, k, pooled, rc, s):
    dev = torch.device("cuda:0")
    r = lambda *sh: torch.randn(*sh, device=dev)
    G, Yl = r(rows, n), r(rows, n)
    sc, sh, mu, isd, m1l, m2l = r(n), r(n), r(n), r(n).abs() + .5, r(n) * .05, r(n) * .05
    x = r(rows, 4) if rc else r(rows, k)
    w0 = r(k, 4) if rc else None
    pa, pb, mup, isp = r(k), r(k), r(k), r(k).abs() + .5
    Wt = r(k, n)
    W = Wt.t().contiguous()
    groups = rows // s
    arg = torch.randint(0, s, (groups, n), device=dev, dtype=torch.uint8)
    dcl, alpha, beta = r(groups, n), r(n), r(n)
    chunks = _lib.btr_sa_gemm_tn_chunks(rows, n, k)
    dz, pw, dw = r(rows, k), r(chunks, n, k), r(n, k)
    part = r(1024, 2, max(n, k))
    m1, m2, dg, db = r(k), r(k), r(k), r(k)
    st = _ext._stream(0)
    nb = _lib.btr_sa_rc_wgrad_blocks(rows, k)
    pw0, dw0 = r(nb, k, 4), r(k, 4)

    def fused():
        _ext._call(_lib.btr_sa_bwd_fused, rows, n, k, _p(G), n, None if pooled else _p(Yl),
                   _p(sc), _p(sh), _p(mu), _p(isd), _p(m1l), _p(m2l), s,
                   _p(arg) if pooled else None, _p(dcl) if pooled else None,
                   _p(alpha) if pooled else None, _p(beta) if pooled else None, _p(x),
                   4 if rc else k, _p(w0), _p(pa), _p(pb), _p(mup), _p(isp), _p(Wt), n, _p(dz), k,
                   _p(pw), _p(dw), _p(part), _p(m1), _p(m2), _p(dg), _p(db), st)

    def old():
        # what the layer loop issued before: [apply of layer l's sums] wgrad, dgrad, sums of l-1
        if not pooled:
            _ext._call(_lib.btr_sa_bn_relu_bwd_apply, rows, n, n, _p(G), _p(Yl), _p(sc), _p(sh),
                       _p(mu), _p(isd), _p(m1l), _p(m2l), st)
        if pooled:
            _ext._call(_lib.btr_sa_gemm_tn_pool, rows, n, k, _p(G), n, s, _p(arg), _p(dcl),
                       _p(alpha), _p(beta), _p(x), k, _p(pa), _p(pb), _p(pw), _p(dw), st)
            _ext._call(_lib.btr_sa_gemm_nt_pool, rows, k, n, _p(G), n, _p(Wt), n, _p(dz), k, s,
                       _p(arg), _p(dcl), _p(alpha), _p(beta), st)
        elif rc:
            _ext._call(_lib.btr_sa_gemm_tn_rc, rows, n, k, _p(G), n, _p(x), _p(w0), _p(pa), _p(pb),
                       _p(pw), _p(dw), st)
            _ext._call(_lib.btr_sa_gemm_nt, rows, k, n, _p(G), n, _p(Wt), n, _p(dz), k, None, None,
                       None, st)
        else:
            _ext._call(_lib.btr_sa_gemm_tn, rows, n, k, _p(G), n, _p(x), k, _p(pa), _p(pb),
                       _p(pw), _p(dw), st)
            _ext._call(_lib.btr_sa_gemm_nt, rows, k, n, _p(G), n, _p(Wt), n, _p(dz), k, None, None,
                       None, st)
        if rc:
            _ext._call(_lib.btr_sa_bn_relu_bwd_rc, rows, k, k, _p(dz), _p(x), _p(w0), _p(pa),
                       _p(pb), _p(mup), _p(isp), _p(part), _p(m1), _p(m2), _p(dg), _p(db),
                       _p(pw0), _p(dw0), st)
        else:
            _ext._call(_lib.btr_sa_bn_relu_bwd_sums, rows, k, k, _p(dz), _p(x), _p(pa), _p(pb),
                       _p(mup), _p(isp), _p(part), _p(m1), _p(m2), _p(dg), _p(db), st)

    tf, to = timed(fused), timed(old)
    gb = rows * 4.0 * ((n if pooled else 2 * n) + (4 if rc else k) + k) / 1e9
    print("rows %7d n %3d k %3d %s%s: fused %7.1f us (%.2f TB/s of its 4-pass bytes), "
          "separate calls %7.1f us%s" % (
              rows, n, k, "pooled" if pooled else "bn    ", " rc" if rc else "   ", tf,
              gb / tf * 1e3, to, "  (incl. the rc layer-0 pass the fused flow still runs)"
              if rc else ""))


if __name__ == "__main__":
    case(706560, 128, 64, True, False, 64)     # SA1 pooled layer (compact-row count, dense form)
    case(706560, 64, 64, False, True, 64)      # SA1 hidden layer over the recomputed first layer
    case(114688, 128, 128, False, False, 32)   # SA2 hidden layer
    case(65536, 128, 128, False, False, 16)    # SA3 hidden layer
    case(32768, 128, 128, False, False, 16)    # SA4 / vote aggregation hidden layer
    case(114688, 256, 128, True, False, 32)    # SA2 pooled layer (256-wide variant)
    case(65536, 256, 128, True, False, 16)     # SA3 pooled layer
    case(32768, 256, 128, True, False, 16)     # SA4 pooled layer
    case(8192, 256, 256, False, False, 16)     # feature-propagation chain layer
