#!/usr/bin/env python3
"""GPU time per phase of the software-pipelined FSB step, from hipEvents on the MAIN stream at
the phase boundaries of the un-profiled loop (module forward hooks / gradient hooks): the
sum of a phase's kernel durations (profiles/*_one_step.md) against this tells where the
main stream waits or is slowed down -- a kernel trace slows the host too much to show it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
batches = [synthetic.make_batch(s, 8, 40000, cfg, device=dev) for s in range(3)]
all_marks = [[]]


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    all_marks[-1].append((name, e))


def fwd_hook(name):
    def hook(mod, inp, out):
        mark(name)
    return hook


core = net
core.backbone_net.register_forward_hook(fwd_hook("backbone fwd"))
core.vgen.register_forward_hook(fwd_hook("voting fwd"))
core.pnet.register_forward_hook(fwd_hook("proposal fwd"))


def grad_mark(t, name):
    if t.requires_grad:
        t.register_hook(lambda g: mark(name))


orig_loss = train.loss_helper.get_loss


def loss_with_marks(end_points, cfg_):
    out = orig_loss(end_points, cfg_)
    mark("loss fwd")
    grad_mark(end_points['aggregated_vote_xyz'], "loss bwd + proposal bwd (to agg xyz)")
    grad_mark(end_points['vote_features'], "vote aggregation bwd")
    grad_mark(end_points['seed_features'], "voting bwd")
    return out


pipelined = "--sequential" not in sys.argv and "--no-pyramid" not in sys.argv
no_pyramid = "--no-pyramid" in sys.argv   # same batch every step, its pyramid computed once
occ = None
if "--occupant" in sys.argv:   # (with --no-pyramid) a sleeping stand-in for the FPS: wgs usec lds
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "liboccupant.so"))
    i = sys.argv.index("--occupant")
    o_wgs, o_usec, o_lds = int(sys.argv[i + 1]), int(sys.argv[i + 2]), int(sys.argv[i + 3])
    perm = torch.randperm(1024, device=dev, dtype=torch.int32)
    sink = torch.zeros(4, device=dev)
    side = torch.cuda.Stream()

    def occ():
        side.wait_stream(torch.cuda.current_stream())
        lib.occupant_launch_lds(o_wgs, 0, o_usec, ctypes.c_void_p(perm.data_ptr()), perm.numel(),
                                ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream),
                                o_lds)
ready = core.backbone_net.prefetch_sampling(batches[0]['point_clouds']) if no_pyramid else None
torch.cuda.synchronize()
steps, warm = 30, 8
sampling = core.backbone_net.prefetch_sampling(batches[0]['point_clouds']) if pipelined else None
rows = {}
order = []
free_running = "--sync" not in sys.argv   # default: no host synchronisation between the steps
all_marks = []
for it in range(warm + steps):
    marks = []
    all_marks.append(marks)
    b = batches[it % 3]
    mark("start")
    if pipelined:
        loss, end = train.train_step(net, opt, b, cfg, sampling=sampling,
                                     next_batch=batches[(it + 1) % 3], criterion=loss_with_marks)
        sampling = end['next_sampling']
    elif no_pyramid:
        if occ is not None:
            occ()
        loss, end = train.train_step(net, opt, batches[0], cfg, sampling=ready,
                                     criterion=loss_with_marks)
    else:
        loss, end = train.train_step(net, opt, b, cfg, criterion=loss_with_marks)
    mark("backbone bwd + Adam")
    if not free_running:
        torch.cuda.synchronize()
torch.cuda.synchronize()
for it, ms in enumerate(all_marks):
    if it < warm:
        continue
    prev = ms[0][1]
    for name, e in ms[1:]:
        rows.setdefault(name, []).append(prev.elapsed_time(e) * 1e3)
        if name not in order:
            order.append(name)
        prev = e
    if it + 1 < len(all_marks):   # the distance to the next step's start mark
        rows.setdefault("(to the next step's start)", []).append(
            prev.elapsed_time(all_marks[it + 1][0][1]) * 1e3)
if "(to the next step's start)" in rows:
    order.append("(to the next step's start)")
tot = 0.0
for name in order:
    v = rows[name]
    avg = sum(v) / len(v)
    tot += avg
    print("%-44s %8.1f us  (min %7.1f  max %7.1f)" % (name, avg, min(v), max(v)))
print("%-44s %8.1f us   [%s]" % ("sum", tot, "free-running loop" if free_running else
                                  "host synchronised after every step (--sync)"))
