#!/usr/bin/env python3
"""Full-size Back-to-Reality step (2 x 8 x 40 000 points): the fused HIP path and the nine-op +
torch float32 composition, each against the SAME step evaluated in float64 on the GPU (indices
from the float32 ops on the float32 casts, proposals and vote-ball neighbour lists pinned to the
nine-op run's; everything differentiable in float64 torch ops).

Why: the two float32 paths differ by 1 - 3 % in relative L2 on many gradient tensors although
every forward quantity agrees to 1e-4.  If that were one flipped max-pool decision, the deviation
would start at one pooled layer; measured (tools/diag_c3_grads.py) it starts at the proposal
head's first layer and grows smoothly towards SA1 -- the signature of sums that cancel (a
detection loss pulls positives and negatives apart), i.e. of conditioning, not of a wrong path.
This tool states the yardstick: how far each float32 path is from float64.
Usage: diag_c3_f64.py [first_seed ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from backtoreality_amd.votenet import config, synthetic  # noqa: E402
import f64_path  # noqa: E402  (tests/f64_path.py: the float64 evaluation)


def main():
    dev = torch.device("cuda:0")
    cfg = config.scannet_md40()
    seeds = [int(a) for a in sys.argv[1:]] or [24, 0]
    for s in seeds:
        bS = synthetic.make_batch(s, 8, 40000, cfg, device=dev)
        bT = synthetic.make_batch(100000 + s, 8, 40000, cfg, device=dev)
        lu, uS, uT, gu, idx_u = f64_path.br_step(cfg, bS, bT, dev, fused=False)
        pins = (uS['aggregated_vote_inds'], uT['aggregated_vote_inds'])
        lf, fS, fT, gf, _ = f64_path.br_step(cfg, bS, bT, dev, fused=True, vote_inds=pins,
                                              vote_idx=idx_u)
        l64, _, _, g64, _ = f64_path.br_step(cfg, bS, bT, dev, fused=False, vote_inds=pins,
                                             vote_idx=idx_u, float64=True)
        print("seed %d: loss f64 %.9f | nine-op f32 rel %.2e | fused f32 rel %.2e" % (
            s, float(l64), abs(float(lu) - float(l64)) / abs(float(l64)),
            abs(float(lf) - float(l64)) / abs(float(l64))))
        rows = f64_path.errors_vs_f64(gf, gu, g64)
        for st, (eh, er, efu, n) in rows.items():
            print("   %-28s fused-vs-f64 %.5f  nine-op-vs-f64 %.5f  fused-vs-nine-op %.5f  (%s)"
                  % (st, eh, er, efu, n))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
