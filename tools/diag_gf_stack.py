#!/usr/bin/env python3
"""Which gradients differ between the decoder stack and the module loop, and by how much; and
between two runs of the module loop (run-to-run determinism of the baseline)."""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pytest
import test_gf_stack_gpu as T


class MP(object):
    def setenv(self, k, v): os.environ[k] = v
    def setattr(self, o, n, v): setattr(o, n, v)


cuda = torch.device("cuda:0")
mp = MP()
drop = float(os.environ.get("DROP", "0"))
a = T._step(cuda, mp, True, drop)
b = T._step(cuda, mp, False, drop)
c = T._step(cuda, mp, False, drop)
for name, x, y in (("stack vs loop", a, b), ("loop vs loop", b, c)):
    bad = []
    for n in y[3]:
        if y[3][n] is None:
            continue
        if not torch.equal(x[3][n], y[3][n]):
            d = float((x[3][n] - y[3][n]).norm() / (y[3][n].norm() + 1e-30))
            bad.append((n, d))
    print(name, "took", x[0], y[0], "loss", x[1], y[1], "differing grads:", len(bad), "of", len(y[3]))
    for n, d in bad[:60]:
        print("   %-70s %.3e" % (n, d))
