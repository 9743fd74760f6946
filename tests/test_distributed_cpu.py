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
