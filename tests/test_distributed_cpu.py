"""The N>1 path on CPU: two `gloo` processes run the data-parallel VoteNet step (over the
oracle `_ext`, since there is no GPU here) and must end with identical parameters whose
gradients are the mean of the per-rank gradients -- the only collective on the path is DDP's
gradient all-reduce."""
import copy
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, mode="flat"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BTR_DP=mode)
    torch.set_num_threads(2)
    import oracle
    from backtoreality_amd.pointnet2 import pointnet2_utils
    from backtoreality_amd.votenet import config, synthetic, train
    pointnet2_utils._ext = oracle.ext_cpu

    r, w, _ = train.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    cfg = config.scannet_md40()
    net = train.build_model(cfg, dev, num_proposal=64)       # same seed on every rank
    solo = copy.deepcopy(net)
    ddp = train.wrap_ddp(net, dev)
    if mode == "ddp":
        assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    else:
        assert isinstance(ddp, train.FlatGradParallel)
        assert list(ddp.state_dict())[0].startswith("module.")
    opt = train.make_optimizer(net)
    batch = synthetic.make_batch(10 + rank, 1, 2304, cfg)     # a different scene per rank
    loss, _ = train.train_step(ddp, opt, batch, cfg)
    if mode == "flat":   # a second step from the same weights gives the same mean gradient
        g1 = [p.grad.clone() for p in net.parameters()]
        net.load_state_dict(solo.state_dict())             # back to the initial weights
        opt = train.make_optimizer(net)
        loss, _ = train.train_step(ddp, opt, batch, cfg)
        for p, g in zip(net.parameters(), g1):
            assert p.grad.data_ptr() >= ddp.flat_grad.data_ptr()
            assert torch.allclose(p.grad, g, rtol=1e-4, atol=1e-7)

    # reference: local gradient without DDP, averaged by hand
    solo_opt = train.make_optimizer(solo)
    train.train_step(solo, solo_opt, batch, cfg)
    worst = 0.0
    for (name, p), (_, ps) in zip(net.named_parameters(), solo.named_parameters()):
        g = ps.grad.clone()
        dist.all_reduce(g)
        g /= world
        denom = g.abs().max().item() + 1e-12
        worst = max(worst, (p.grad - g).abs().max().item() / denom)
    # parameters stay replicated after the step
    flat = torch.cat([p.detach().flatten() for p in net.parameters()])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    same = all(torch.equal(o, flat) for o in other)
    # BN buffers are NOT broadcast (broadcast_buffers=False, train_GF_FSB.py:250)
    q.put((rank, float(loss), worst, same))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("mode", ["flat", "ddp"])
def test_two_rank_gloo_step_matches_manual_average(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert results[0][1] != results[1][1]          # different shards -> different losses
    for _, _, worst, same in results:
        assert worst < 1e-5, worst
        assert same


def _worker_unused(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BTR_DP="flat")
    torch.set_num_threads(1)
    from backtoreality_amd.votenet import train
    train.init_distributed(backend="gloo")

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.used = torch.nn.Linear(4, 3)
            self.unused = torch.nn.Linear(4, 2)      # no loss term: never gets a gradient
            self.frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)

        def forward(self, x):
            return self.used(x) * self.frozen.sum()

    torch.manual_seed(100 + rank)                    # different initial weights per rank ...
    net = Net()
    dp = train.wrap_ddp(net, torch.device("cpu"))    # ... broadcast from rank 0 here
    w0 = net.used.weight.detach().clone()
    unused0 = net.unused.weight.detach().clone()
    gathered = [torch.empty_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    replicated = all(torch.equal(g, w0) for g in gathered)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=0.1)
    x = torch.full((5, 4), float(rank + 1))
    ok = True
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        dp(x).sum().backward()
        local = net.used.weight.grad.clone()
        train._sync_grads(dp)
        want = local.clone()
        dist.all_reduce(want)
        want /= world
        ok = ok and torch.allclose(net.used.weight.grad, want)
        ok = ok and net.unused.weight.grad is None and net.frozen.grad is None
        ok = ok and net.used.weight.grad.data_ptr() >= dp.flat_grad.data_ptr()
        opt.step()
    same_unused = torch.equal(net.unused.weight, unused0)        # skipped by the optimizer
    flat = torch.cat([p.detach().flatten() for p in net.parameters()])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    q.put((rank, replicated, bool(ok), same_unused, all(torch.equal(o, flat) for o in other)))
    dist.destroy_process_group()


def test_flat_grad_parallel_with_unused_and_frozen_parameters():
    """FlatGradParallel on a module with a parameter that never receives a gradient (the
    CenterRefine model's jitter_netD) and a frozen one: parameters are broadcast from rank 0,
    gradients averaged, the unused parameter keeps grad None on every rank (the optimizer
    skips it alike), replicas stay identical."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_unused, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, replicated, ok, same_unused, same in results:
        assert replicated and ok and same_unused and same


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started plainly (no torchrun, WORLD_SIZE unset) must start two
    ranks itself and never fall back to a silent one-rank run (gloo here; RCCL on a GPU box)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
                          "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line)["launch_check"] == 2


def _worker_small(rank, world, port, q):
    """Two data-parallel steps on a tiny scene (the whole path: rendezvous, broadcast at
    construction, one flat all-reduce per step) at the node's REAL rank count."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BTR_DP="flat")
    torch.set_num_threads(1)
    import oracle
    from backtoreality_amd.pointnet2 import pointnet2_utils
    from backtoreality_amd.votenet import config, synthetic, train
    pointnet2_utils._ext = oracle.ext_cpu
    r, w, _ = train.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = config.scannet_md40()
    torch.manual_seed(1000 + rank)     # different initial weights: rank 0's must win
    net = train.build_model(cfg, torch.device("cpu"), num_proposal=16, seed=1000 + rank)
    dp = train.wrap_ddp(net, torch.device("cpu"))
    opt = train.make_optimizer(net)
    losses = []
    for step in range(2):
        batch = synthetic.make_batch(50 * rank + step, 1, 1024, cfg)
        loss, _ = train.train_step(dp, opt, batch, cfg)
        losses.append(float(loss))
    flat = torch.cat([p.detach().flatten() for p in net.parameters()])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    ones = torch.ones(1)
    dist.all_reduce(ones)
    q.put((rank, int(ones.item()), all(torch.equal(o, flat) for o in other),
           all(l == l for l in losses)))
    dist.destroy_process_group()


def test_eight_rank_gloo_two_steps_tiny_scene():
    """BASELINE configs[2] / [4] run on 8 GPUs of one node: the rendezvous + FlatGradParallel
    path at world_size 8 (gloo, CPU oracle `_ext`, one 1 024-point scene per rank and step).
    Replicas must stay bit-identical although every rank starts from its own weights (rank 0's
    are broadcast) and sees its own shard."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_small, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    for _, ranks, same, finite in results:
        assert ranks == world and same and finite


def test_bench_launches_eight_ranks():
    """`python bench.py --gpus 8 --launch-check`: the self-launch path at the rank count of the
    driver's scaling run (torch.distributed.run, 127.0.0.1 rendezvous, one all-reduce)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8",
                          "--launch-check"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line)["launch_check"] == 8
