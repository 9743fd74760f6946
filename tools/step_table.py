#!/usr/bin/env python3
"""What one training step is made of, from a rocprofv3 timeline of the bench loop
(tools/rocpd_timeline.py output): kernel time per queue and per family, ranked.

    tools/step_table.py gpurun_out/<tag>/timeline.txt > profiles/<tag>_step_table.md

One script, so that the tables of successive builds are comparable (round-4 review, weak #11)."""
import collections
import re
import sys

FAMILIES = [
    ("large-scene FPS + its spatial sort", r"fps_bucket_kernel|fps_sortm_|fps_box"),
    ("pyramid levels 2-4, vote FPS", r"fps_regs_kernel|fps_prefix_"),
    ("ball query, grouping plans, inverted lists", r"bqb_|bq_|sac_(count|scan|fill|csr)|csr_|gather_rows_kernel|three_nn"),
    ("GEMM family (NT / TN / fused backward / Gram form / small-M / per-point first layer)",
     r"gemm_nt|gemm_tn|sa_bwd_fused|sa_bwd_gram|gram_|sa_fwd_stream|reduce_chunks|ppfl_"),
    ("BatchNorm finalisers and element-wise BatchNorm / ReLU backward", r"bn_finalize|bn_bwd_finalize|bn_relu_bwd"),
    ("max-pool forward / backward, scatter to points", r"sa_pool|sac_pool|sac_reduce|csr_reduce|ti_reduce|gather_.*grad|three_interpolate"),
    ("layout (rows <-> (B, C, N)), weight preparation, bias sums", r"pm_rows|pm_out|prep_weights|colsum|transpose|rows_to_bcp|vote_assemble|copyBuffer|fillBuffer"),
    ("decoder: attention, LayerNorm, dropout", r"attn_|ln_fwd|ln_bwd|relu_drop|add2_kernel|sum_rows"),
    ("loss, decode, optimizer, clipping", r"loss_|gf_|adam|grad_sumsq|grad_norm|nms"),
]


def main():
    rows = []
    for ln in open(sys.argv[1]):
        m = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(.*)", ln)
        if m:
            rows.append((float(m.group(2)), int(m.group(4)), m.group(5).strip()))
    if not rows:
        sys.exit("no timeline rows in %s" % sys.argv[1])
    queues = collections.Counter(q for _, q, _ in rows)
    main_q = max(queues, key=queues.get)
    print("One step of `%s`: %d dispatches on %d queues; main queue %d dispatches, %.0f us of kernel time;"
          " side queue(s) %d dispatches, %.0f us.\n" % (
              sys.argv[1], len(rows), len(queues), queues[main_q],
              sum(d for d, q, _ in rows if q == main_q), len(rows) - queues[main_q],
              sum(d for d, q, _ in rows if q != main_q)))
    print("| family | main queue: launches | us | side queues: launches | us |")
    print("|---|---|---|---|---|")
    fam = collections.OrderedDict((name, [0, 0.0, 0, 0.0]) for name, _ in FAMILIES)
    fam["torch element-wise / other"] = [0, 0.0, 0, 0.0]
    for d, q, name in rows:
        for f, pat in FAMILIES:
            if re.search(pat, name):
                break
        else:
            f = "torch element-wise / other"
        o = 0 if q == main_q else 2
        fam[f][o] += 1
        fam[f][o + 1] += d
    for f, (n0, u0, n1, u1) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
        print("| %s | %d | %.0f | %d | %.0f |" % (f, n0, u0, n1, u1))
    print("\nLargest kernels of the main queue:\n")
    print("| kernel | launches | us |")
    print("|---|---|---|")
    per = collections.defaultdict(lambda: [0, 0.0])
    for d, q, name in rows:
        if q == main_q:
            per[name[:90]][0] += 1
            per[name[:90]][1] += d
    for name, (n, u) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
        print("| `%s` | %d | %.0f |" % (name, n, u))


if __name__ == "__main__":
    main()
