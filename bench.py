#!/usr/bin/env python3
"""Headline benchmark: scenes/sec of the VoteNet FSB training step (forward + loss + backward
+ Adam) on synthetic 40 000-point scenes, batch 8 per GPU -- BASELINE.json configs[1].

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job scenes/s, a
`roofline` object for the dominant hand-written kernel measured live with HIP events on the
launch stream, and (N=1 only) a `cpu_baseline` object: the same step run over the CPU oracle
on the host cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=8, help="scenes per GPU")
    p.add_argument("--points", type=int, default=40000)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-points", type=int, default=40000)
    return p.parse_args()


def cpu_baseline(cfg, points):
    """The same training step over the CPU oracle (kind "port": the reference has no CPU path
    for the nine ops).  Bounded sample: ONE 40k-point scene, one untimed + one timed step."""
    import oracle
    from backtoreality_amd.pointnet2 import pointnet2_utils
    from backtoreality_amd.votenet import synthetic, train

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    saved = pointnet2_utils._ext
    pointnet2_utils._ext = oracle.ext_cpu
    try:
        net = train.build_model(cfg, torch.device("cpu"))
        opt = train.make_optimizer(net)
        batch = synthetic.make_batch(0, 1, points, cfg)
        t0 = time.time()
        train.train_step(net, opt, batch, cfg)
        first = time.time() - t0
        steps = 1 if first > 12 else 2
        t0 = time.time()
        for _ in range(steps):
            train.train_step(net, opt, batch, cfg)
        dt = (time.time() - t0) / steps
    finally:
        pointnet2_utils._ext = saved
    return {"value": 1.0 / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": "VoteNet FSB step (fwd+loss+bwd+Adam), batch 1 x %d points, %d timed "
                      "step(s) after 1 warm-up, C oracle kernels (OpenMP) + torch CPU "
                      "conv/BN" % (points, steps)}


def main():
    args = parse()
    from backtoreality_amd.pointnet2 import _ext
    from backtoreality_amd.votenet import config, synthetic, train

    rank, world, local_rank = train.init_distributed()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus>1"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = config.scannet_md40()
    net = train.build_model(cfg, dev)
    ddp = train.wrap_ddp(net, dev)
    opt = train.make_optimizer(net)
    B = args.batch
    batch = synthetic.make_batch(rank * B, B, args.points, cfg, device=dev)  # resident in HBM

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train.train_step(ddp, opt, batch, cfg)
    barrier()
    _ext.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        train.train_step(ddp, opt, batch, cfg)
    barrier()
    elapsed = time.perf_counter() - t0
    kernels = _ext.timing_end()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        scenes = world * B * args.steps
        out = {
            "metric": "scenes/sec (40k-pt VoteNet fwd+bwd)",
            "value": scenes / elapsed,
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "VoteNet FSB train step (fwd+loss+bwd+Adam), %d points, "
                                   "batch %d per GPU, scannet-md40 heads" % (args.points, B),
                       "points": args.points, "batch_per_gpu": B, "parallelism": "dp%d" % world},
        }
        out.update(roofline_objects(kernels, B, args.points))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_points)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def roofline_objects(kernels, B, N):
    """kernels: {(op, shape-key): [ms, ...]} from the HIP-event timer in `_ext`.
    Algorithmic bytes per launch: ball_query B*(12N + 12M + 4MS);
    FPS B*(12N + 4M)."""
    per_op = {}
    for (op, key), times in kernels.items():
        ms = sum(times) / len(times)
        per_op["%s%s" % (op, list(key))] = {"avg_ms": ms, "launches": len(times)}
    res = {"kernels": per_op}

    def pick(op):
        cands = [(sum(t), k, t) for (o, k), t in kernels.items() if o == op]
        if not cands:
            return None
        _, key, times = max(cands)
        return key, sum(times) / len(times)

    bq = pick("ball_query")
    if bq:
        (b, n, m, s), ms = bq
        nbytes = b * (12 * n + 12 * m + 4 * m * s)
        ach = nbytes / (ms * 1e-3) / 1e9
        res["ball_query_roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                      "traffic": None, "shape": [b, n, m, s], "avg_ms": ms,
                                      "distance_tests": b * n * m}
    fps = pick("furthest_point_sampling")
    if fps:
        (b, n, m), ms = fps
        nbytes = b * (12 * n + 4 * m)
        ach = nbytes / (ms * 1e-3) / 1e9
        res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "kernel": "fps_kernel", "shape": [b, n, m], "avg_ms": ms,
                           "streaming_GBs": b * (m - 1) * n * 20 / (ms * 1e-3) / 1e9,
                           "iterations_per_s": b * (m - 1) / (ms * 1e-3)}
    return res


if __name__ == "__main__":
    main()
