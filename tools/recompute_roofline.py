#!/usr/bin/env python3
"""Rebuild the three roofline fractions of a bench.py line from the rocprofv3 one-step table
of the same command, so the live HIP-event figures can be checked against the profiler's:

    tools/recompute_roofline.py profiles/r02_x_bench.json profiles/r02_x_one_step.md

Taken from the JSON line: only the ALGORITHMIC work (bytes per launch of the FPS and the ball
query, flops per step of the MFMA GEMM family) -- functions of the shapes, not measurements.
Taken from the profile: the kernel durations (tools/rocpd_step.py: per kernel name the calls,
the summed and the longest duration inside one steady-state step).

  FPS         fps_bucket_kernel's longest call (the live figure is an event pair around that
              kernel alone)
  ball query  the longest bqb_query_kernel (or bq_grid_query_kernel) call + its set-up
              launches (bqb_box / bqb_super in builds that had them, or the grid build), longest call of each
  grouped MLP every gemm_nt_kernel / sa_fwd_stream_kernel / gemm_tn_kernel / gemm_tn_x6_kernel /
              sa_bwd_fused_kernel launch of the step (+ every reduce_chunks* launch, the second half of the split-K)
              against the f32 MFMA peak

Prints a table: live figure (JSON), profile figure, ratio."""
import json
import re
import sys

HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3


def read_table(path):
    rows = []
    for line in open(path):
        m = re.match(r"\| `(.*)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|(?: ([\d.]+) \|)?", line)
        if m:
            name, calls, tot, _, longest = m.groups()
            rows.append((name, int(calls), float(tot),
                         float(longest) if longest else float(tot) / int(calls)))
    return rows


def main():
    line = [l for l in open(sys.argv[1]) if l.lstrip().startswith("{")][-1]
    js = json.loads(line)
    rows = read_table(sys.argv[2])

    def longest(sub):
        c = [r[3] for r in rows if sub in r[0]]
        return max(c) if c else 0.0

    def total(sub):
        return sum(r[2] for r in rows if sub in r[0])

    out = []
    fps = js.get("roofline")
    if fps:
        us = longest("fps_bucket_kernel")
        if "fps_sortm" in fps.get("kernel", ""):   # older lines timed the whole call
            us += sum(r[3] for r in rows if "fps_sortm_" in r[0])
        ach = fps["algorithmic_bytes"] / (us * 1e-6) / 1e9
        out.append(("roofline (FPS)", fps["avg_ms"] * 1e3, us, fps["frac"], ach / HBM_PEAK_GBS))
    bq = js.get("ball_query_roofline")
    if bq:
        if longest("bqb_query_kernel"):
            us = longest("bqb_query_kernel") + longest("bqb_box_kernel") + \
                longest("bqb_super_kernel")
        else:
            us = longest("bq_grid_query_kernel") + longest("bq_grid_build")
        ach = bq["algorithmic_bytes"] / (us * 1e-6) / 1e9
        out.append(("ball_query_roofline", bq["avg_ms"] * 1e3, us, bq["frac"],
                    ach / HBM_PEAK_GBS))
    mlp = js.get("mlp_roofline")
    if mlp:
        # every kernel of the family: NT, both TN kernels (the bf16x6 one is `gemm_tn_x6_kernel`:
        # rounds 3's tables matched "gemm_tn_kernel" only and so left it and the batched
        # reductions out -- the 0.63 "profile frac" of r03_k was really 0.40), the fused backward
        # (two products per launch) and all split-K reductions
        # ... and the streaming forward / input-gradient kernel (round 4; the first r04_f / r04_h
        # tables left it out: their 1.92 / 1.79 ms, 0.61 / 0.65 were 2.5 / 2.4 ms, 0.47 / 0.49)
        # ... and round 5's members: the Gram-form backward with its helper launches, the small-M
        # NT kernel, the per-point first layer
        us = total("gemm_nt_kernel") + total("gemm_tn_") + total("sa_bwd_fused_kernel") + \
            total("sa_fwd_stream_kernel") + total("reduce_chunks") + total("sa_bwd_gram") + \
            total("gram_") + total("gemm_nt_sm_kernel") + total("ppfl_")
        ach = mlp["gflop_per_step"] * 1e9 / (us * 1e-6) / 1e12
        out.append(("mlp_roofline", mlp["ms_per_step"] * 1e3, us, mlp["frac"],
                    ach / MFMA_F32_PEAK_TF))
    print("| object | live us | profile us | live frac | profile frac | profile/live time |")
    print("|---|---|---|---|---|---|")
    for name, live_us, prof_us, lf, pf in out:
        print("| %s | %.1f | %.1f | %.5f | %.5f | %.3f |" % (name, live_us, prof_us, lf, pf,
                                                             prof_us / live_us))


if __name__ == "__main__":
    main()
