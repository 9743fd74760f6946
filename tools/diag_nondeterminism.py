import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from backtoreality_amd.votenet import config, synthetic, train
cuda = torch.device("cuda:0")
cfg = config.scannet_md40()
batches = [synthetic.make_batch(10 * i, 2, 20000, cfg, device=cuda) for i in range(3)]
def run(pipelined):
    net = train.build_model(cfg, cuda, seed=0)
    opt = train.make_optimizer(net)
    losses, sampling, vi = [], None, []
    for i, b in enumerate(batches):
        nxt = batches[i + 1] if pipelined and i + 1 < len(batches) else None
        loss, end = train.train_step(net, opt, b, cfg, sampling=sampling, next_batch=nxt)
        sampling = end.get('next_sampling')
        losses.append(float(loss)); vi.append(end['aggregated_vote_inds'].clone())
    return losses, vi
a, va = run(False); b, vb = run(False); c, vc = run(True)
print("plain  ", a); print("plain 2", b); print("piped  ", c)
print("vote inds equal plain/plain2:", [bool(torch.equal(x, y)) for x, y in zip(va, vb)], " plain/piped:", [bool(torch.equal(x, y)) for x, y in zip(va, vc)])
