#!/usr/bin/env python3
"""Loss curve of a short GroupFree3D training run (two alternating synthetic batches, AdamW,
software-pipelined sampling) with the decoder stack replayed as HIP graphs on three lanes vs
issued launch by launch on one stream vs the per-module loop: all must go down alike (the runs
part ways numerically after the first update -- the backbone's float atomics -- so this is a
curve-level check; the bit-level one is tests/test_gf_stack_gpu.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT)
    import torch
    from backtoreality_amd.groupfree import train as gf_train
    from backtoreality_amd.pointnet2 import _ext
    from backtoreality_amd.votenet import config, synthetic
    dev = torch.device("cuda:0")
    cfg = config.scannet_md40()
    torch.manual_seed(0)
    net = gf_train.build_model(cfg, dev)
    opt = gf_train.make_optimizer(net)
    batches = [synthetic.make_batch(1000 * i, 4, 20000, cfg, use_height=False, device=dev)
               for i in range(2)]
    sampling, losses = None, []
    for i in range(80):
        loss, end = gf_train.train_step(net, opt, batches[i % 2], cfg, sampling=sampling,
                                        next_batch=batches[(i + 1) % 2])
        sampling = end.get('next_sampling')
        losses.append(float(loss))
    print(" ".join("%.3f" % losses[i] for i in (0, 1, 5, 10, 20, 30, 40, 60, 79)),
          "| graphs", _ext.graph_stats())
else:
    for name, env in (("graphs, three lanes", {}), ("single launches", {"BTR_GRAPHS": "0"}),
                      ("per-module loop", {"BTR_FUSED_GF_STACK": "0"})):
        out = subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, **env),
                             capture_output=True, text=True)
        print("%-22s %s" % (name, out.stdout.strip().splitlines()[-1] if out.stdout.strip()
                            else out.stderr[-400:]))
