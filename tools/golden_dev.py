"""Prints, instead of asserting, every deviation of the GPU path from the reference-generated
fixtures (tests/test_golden_cpu.py checks run with recording asserts): used to set the stated
bounds.  python tools/golden_dev.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_cpu as T  # noqa: E402

rows = []


def rec_allclose(a, b, rtol=1e-7, atol=0, err_msg="", **kw):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    rows.append((err_msg or "tensor%s" % (a.shape,), err, rtol))


def rec_grad(got, want, tol, name):
    got = got.detach().cpu().numpy()
    err = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
    l2 = np.linalg.norm(got - want) / np.linalg.norm(want)
    rows.append((name + " (max-norm; rel L2 %.2e)" % l2, err, tol))


np.testing.assert_allclose = rec_allclose
T.grad_close = rec_grad
dev = torch.device("cuda:0")
for fused in ("1", "0"):
    os.environ["BTR_FUSED_SA"] = fused
    for name, run, chk in (("fsb", T.run_votenet, T.check_votenet),
                           ("br", T.run_votenet_br, T.check_votenet_br),
                           ("cr", T.run_votenet_br_jitter, T.check_votenet_br_jitter),
                           ("wsb", T.run_votenet_wsb, T.check_votenet_wsb),
                           ("seed_fps", lambda d, pin: T.run_votenet_sampling(d, "seed_fps", pin),
                            lambda r, a, b: T.check_votenet_sampling(r, "seed_fps", a, b)),
                           ("random", lambda d, pin: T.run_votenet_sampling(d, "random", pin),
                            lambda r, a, b: T.check_votenet_sampling(r, "random", a, b))):
        print("== %s  BTR_FUSED_SA=%s" % (name, fused))
        rows.clear()
        rep = []
        real = T.pinned_vote_inds
        T.pinned_vote_inds = lambda *a, **k: real(*a, report=rep, **k)
        try:
            res = run(dev, pin=True)
        finally:
            T.pinned_vote_inds = real
        chk(res, 1e-4, 1e-3)
        print("   own discrete choices that differ from the fixture's:", rep)
        for k, e, t in rows:
            if e > 0.2 * t:
                print("   %-60s %.3e  (bound %.0e)%s" % (k, e, t, "  <-- over" if e > t else ""))
