"""One rank of the RCCL data-parallel check (started by tests/test_a_rccl_ranks_gpu.py as a
fresh process per GPU; RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment).

Runs the recipe of the reference's GroupFree3D scripts (train_GF_FSB.py:172-190, :250,
:450-474: one process per GPU, NCCL process group, per-rank shards, broadcast_buffers=False)
on the VoteNet step of this package: three software-pipelined train steps under
FlatGradParallel, a different scene shard per rank.  Checks (the ones of the gloo test,
tests/test_distributed_cpu.py:55-71): the all-reduce saw every rank, the replicas stay
bit-identical, the gradients equal the hand-averaged per-rank gradients.  Prints one JSON line."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def other_steps(mode):
    """`br`: the Back-to-Reality step (two forwards per backward, train_Votenet_BR.py:267-289;
    DDP tolerates it with broadcast_buffers=False, train_GF_BR.py:330-331, 356) and `gf`: the
    GroupFree3D step (train_GF_FSB.py:287-322: clip + AdamW) under FlatGradParallel, three
    pipelined steps each; replicas must stay bit-identical, losses finite."""
    from backtoreality_amd.votenet import config, synthetic, train
    os.environ.setdefault("BTR_FORCE_DDP", "1")
    rank, world, local = train.init_distributed(backend="nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    rccl_ranks = int(ones.item())
    cfg = config.scannet_md40()
    B, N = 2, 8192
    losses = []
    if mode == "br":
        net = train.build_model(cfg, dev, num_proposal=64, domain_adaptation=True)
        dp = train.wrap_ddp(net, dev)
        opt = train.make_optimizer(net)
        bs = [synthetic.make_batch(1000 * rank + 10 * i, B, N, cfg, device=dev) for i in range(2)]
        bt = [synthetic.make_batch(100000 + 1000 * rank + 10 * i, B, N, cfg, device=dev)
              for i in range(2)]
        ss = net.backbone_net.prefetch_sampling(bs[0]['point_clouds'])
        st = None
        for i in range(3):
            out = train.train_step_br(dp, opt, bs[i % 2], bt[i % 2], cfg, sampling_S=ss,
                                      sampling_T=st, next_batch_S=bs[(i + 1) % 2],
                                      next_batch_T=bt[(i + 1) % 2])
            ss, st = out[1]['next_sampling'], out[2]['next_sampling']
            losses.append(float(out[0]))
    else:
        from backtoreality_amd.groupfree import train as gf_train
        net = gf_train.build_model(cfg, dev)
        dp = train.wrap_ddp(net, dev)
        opt = gf_train.make_optimizer(net)
        bs = [synthetic.make_batch(1000 * rank + 10 * i, B, N, cfg, device=dev, use_height=False)
              for i in range(2)]
        sampling = net.backbone_net.prefetch_sampling(bs[0]['point_clouds'])
        for i in range(3):
            loss, end = gf_train.train_step(dp, opt, bs[i % 2], cfg, sampling=sampling,
                                            next_batch=bs[(i + 1) % 2])
            sampling = end['next_sampling']
            losses.append(float(loss))
    assert isinstance(dp, train.FlatGradParallel)
    flat = torch.cat([p.detach().flatten() for p in net.parameters()])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    same = all(torch.equal(o, flat) for o in other)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"rank": rank, "mode": mode, "rccl_ranks": rccl_ranks,
                      "replicas_identical": bool(same), "losses": losses,
                      "finite": all(l == l and abs(l) < 1e30 for l in losses)}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] in ("br", "gf"):
        return other_steps(sys.argv[1])
    from backtoreality_amd.votenet import config, synthetic, train
    os.environ.setdefault("BTR_FORCE_DDP", "1")      # a process group also for WORLD_SIZE=1
    rank, world, local = train.init_distributed(backend="nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    rccl_ranks = int(ones.item())
    assert rccl_ranks == world == dist.get_world_size(), (rccl_ranks, world)
    cfg = config.scannet_md40()
    B, N = 2, 8192
    net = train.build_model(cfg, dev, num_proposal=64)           # same seed on every rank
    solo = copy.deepcopy(net)
    dp = train.wrap_ddp(net, dev)
    assert isinstance(dp, train.FlatGradParallel)
    opt = train.make_optimizer(net)
    batches = [synthetic.make_batch(1000 * rank + 10 * i, B, N, cfg, device=dev)
               for i in range(2)]                                 # a different shard per rank

    # step 1 from the common initial weights: gradient = mean over ranks of the local gradients
    sampling = net.backbone_net.prefetch_sampling(batches[0]['point_clouds'])
    loss, end = train.train_step(dp, opt, batches[0], cfg, sampling=sampling,
                                 next_batch=batches[1])
    solo_opt = train.make_optimizer(solo)
    train.train_step(solo, solo_opt, batches[0], cfg)
    worst = 0.0
    for (name, p), (_, ps) in zip(net.named_parameters(), solo.named_parameters()):
        g = ps.grad.clone()
        dist.all_reduce(g)
        g /= world
        worst = max(worst, float((p.grad - g).abs().max() / (g.abs().max() + 1e-12)))
        assert p.grad.data_ptr() >= dp.flat_grad.data_ptr()
    # two more pipelined steps (the next pyramid runs on the side stream under the backward
    # while the all-reduce is issued behind it)
    losses = [float(loss)]
    sampling = end['next_sampling']
    for i in (1, 2):
        loss, end = train.train_step(dp, opt, batches[i % 2], cfg, sampling=sampling,
                                     next_batch=batches[(i + 1) % 2])
        sampling = end['next_sampling']
        losses.append(float(loss))
    flat = torch.cat([p.detach().flatten() for p in net.parameters()])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    same = all(torch.equal(o, flat) for o in other)
    # BatchNorm buffers stay per replica (different shards -> different running statistics)
    rm = net.backbone_net.sa1.mlp_module.layer0.bn.bn.running_mean.detach().clone()
    rms = [torch.empty_like(rm) for _ in range(world)]
    dist.all_gather(rms, rm)
    buffers_differ = world == 1 or not torch.equal(rms[0], rms[-1])
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"rank": rank, "rccl_ranks": rccl_ranks, "grad_rel_err": worst,
                      "replicas_identical": bool(same), "buffers_per_replica": bool(buffers_differ),
                      "losses": losses, "finite": all(l == l and abs(l) < 1e30 for l in losses)}),
          flush=True)


if __name__ == "__main__":
    main()
