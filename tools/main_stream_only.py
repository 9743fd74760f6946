#!/usr/bin/env python3
"""The training step WITHOUT any sampling pyramid beside it: the same batch every step, its
pyramid computed once before the loop (train_step(sampling=handle), no next_batch).  What the
main stream's kernels take when nothing shares the chip -- the floor of the software-pipelined
step, and by difference what the pyramid still costs it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
b = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
b2 = synthetic.make_batch(1, 8, 40000, cfg, device=dev)
h = net.backbone_net.prefetch_sampling(b['point_clouds'])
torch.cuda.synchronize()


occ = None
if "--occupant" in sys.argv:   # a synthetic stand-in for the FPS beside the pyramid-less loop
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "liboccupant.so"))
    i = sys.argv.index("--occupant")
    o_mode, o_usec, o_lds = int(sys.argv[i + 1]), int(sys.argv[i + 2]), int(sys.argv[i + 3])
    o_wgs = int(sys.argv[i + 4]) if len(sys.argv) > i + 4 else 8
    perm = torch.randperm(160 * 1024, device=dev, dtype=torch.int32)
    sink = torch.zeros(4, device=dev)
    side = torch.cuda.Stream()

    def occ():
        side.wait_stream(torch.cuda.current_stream())
        if o_mode < 0:   # one sleeping WAVE per workgroup
            lib.occupant_launch_small(o_wgs, o_usec, ctypes.c_void_p(side.cuda_stream))
            return
        lib.occupant_launch_lds(o_wgs, o_mode, o_usec, ctypes.c_void_p(perm.data_ptr()), perm.numel(),
                                ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream),
                                o_lds)


def loop(n, pipelined):
    global h
    t0 = time.perf_counter()
    if pipelined:
        bs = [b, b2]
        s = net.backbone_net.prefetch_sampling(bs[0]['point_clouds'])
        for i in range(n):
            out = train.train_step(net, opt, bs[i % 2], cfg, sampling=s, next_batch=bs[(i + 1) % 2])
            s = out[1]['next_sampling']
    else:
        for _ in range(n):
            if occ is not None:
                occ()
            train.train_step(net, opt, b, cfg, sampling=h)
            if occ is not None:
                torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for mode in (False, True):
    loop(5, mode)
    train.freeze_gc()
    print("%s: %.3f ms per step" % ("software-pipelined loop (pyramid of the next batch beside the step)"
                                    if mode else "no pyramid at all (same batch, ready handle)      ",
                                    min(loop(20, mode) for _ in range(3))))
