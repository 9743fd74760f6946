"""Pyramid levels 2-4 (FPS of an FPS-ordered prefix): the serial register-resident kernel against
btr_furthest_point_sampling_ordered (parallel check of "the answer is 0..m-1" + the serial kernel
behind it), per level: event-pair time over `reps` calls alone on the chip, the verdict slots the
check wrote, identical indices.  python tools/fps_prefix_ab.py [points] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from backtoreality_amd.pointnet2 import _ext  # noqa: E402
from backtoreality_amd.votenet import config, synthetic  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    cfg = config.scannet_md40()
    for seed in (0, 8):
        batch = synthetic.make_batch(seed, b, n, cfg, device=dev)
        cur = batch['point_clouds'][..., :3].contiguous()
        for m in (2048, 1024, 512, 256):
            if cur.shape[1] <= 4096:
                plain = _ext.furthest_point_sampling(cur, m)
                t_plain = timed(lambda: _ext.furthest_point_sampling(cur, m))
                _ext.mark_fps_ordered(cur)
                need = _ext._idx.btr_fps_ordered_scratch_bytes(b, cur.shape[1], m)
                scratch = torch.zeros((need // 4,), dtype=torch.int32, device=dev)
                out = torch.empty((b, m), dtype=torch.int32, device=dev)
                _ext._call(_ext._idx.btr_furthest_point_sampling_ordered, b, cur.shape[1], m,
                           _ext._p(cur), None, _ext._p(out), 0, _ext._p(scratch), need,
                           _ext._stream(0))
                torch.cuda.synchronize()
                t_ord = timed(lambda: _ext.furthest_point_sampling(cur, m))
                ok = torch.equal(out, plain)
                slots = scratch[4 * b * m:].view(b, -1)   # behind q[b][m] float4
                print("seed %d  %5d -> %4d: serial %7.1f us, ordered %7.1f us, same=%s, "
                      "scenes confirmed %d / %d, identity=%s" % (
                          seed, cur.shape[1], m, t_plain, t_ord, ok,
                          int((slots == 0x600D0001).all(1).sum()), b,
                          bool((plain == torch.arange(m, device=dev, dtype=torch.int32)).all())))
                del cur._btr_fps_ordered
                inds = plain
            else:
                inds = _ext.furthest_point_sampling(cur, m)
            cur = _ext.gather_rows(cur, inds)


if __name__ == "__main__":
    main()
