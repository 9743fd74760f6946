#!/usr/bin/env python3
"""What the next batch's sampling pyramid (side stream) costs the backbone forward of this
batch (main stream), measured directly: the forward with a ready sampling handle alone, and
with prefetch_sampling of another batch issued right before it.  Environment switches apply
(BTR_FWD_STREAM, BTR_FPS_PRIO, ...)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
bb = net.backbone_net
b0 = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
b1 = synthetic.make_batch(1, 8, 40000, cfg, device=dev)
pc0, pc1 = b0['point_clouds'], b1['point_clouds']


def fwd(handle):
    with torch.no_grad():
        return bb(pc0, sampling=handle)


def run(concurrent, reps=20):
    ts = []
    for _ in range(reps):
        h = bb.prefetch_sampling(pc0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if concurrent and occ is not None:
            occ()
        elif concurrent:
            other = bb.prefetch_sampling(pc1)   # side stream: FPS of the other batch starts now
        e0.record()
        fwd(h)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


from backtoreality_amd.pointnet2 import _ext  # noqa: E402
occ = None
if "--occupant" in sys.argv:   # a synthetic co-runner instead of the pyramid (tools/probe/occupant.hip)
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "liboccupant.so"))
    n = 160 * 1024   # 640 KB of ints: one scene's worth of L2-resident data
    perm = torch.randperm(n, device=dev, dtype=torch.int32)
    sink = torch.zeros(4, device=dev)
    side = torch.cuda.Stream()
    i = sys.argv.index("--occupant")
    wgs, mode, usec = int(sys.argv[i + 1]), int(sys.argv[i + 2]), int(sys.argv[i + 3])

    def occ():
        side.wait_stream(torch.cuda.current_stream())
        lib.occupant_launch(wgs, mode, usec, ctypes.c_void_p(perm.data_ptr()), n,
                            ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))

for _ in range(3):
    fwd(bb.prefetch_sampling(pc0))
torch.cuda.synchronize()
a = run(False)
b = run(True)
print("backbone forward alone:            median %.1f us  (min %.1f)" % a)
print("... beside the next pyramid:       median %.1f us  (min %.1f)   +%.1f us" %
      (b[0], b[1], b[0] - a[0]))
