import os, sys, torch
sys.path.insert(0, '/root/repo')
from backtoreality_amd.groupfree import train as gf_train
from backtoreality_amd.votenet import config, synthetic
cuda = torch.device('cuda:0')
cfg = config.scannet_md40()
batch = synthetic.make_batch(0, 2, 8192, cfg, use_height=False, device=cuda)
def run(stream, graphed=False):
    os.environ['BTR_FWD_STREAM'] = stream
    net = gf_train.build_model(cfg, cuda, dropout=0.0)
    opt = gf_train.make_optimizer(net, capturable=graphed)
    if graphed:
        step = gf_train.GraphedTrainStep(net, opt, batch, cfg, warmup=1)
        for _ in range(2): loss, _ = step(batch)
    else:
        for _ in range(3): loss, _ = gf_train.train_step(net, opt, batch, cfg)
    return net, float(loss)
def dev(a, b):
    num = sum(float((x - y).double().pow(2).sum()) for x, y in zip(a.parameters(), b.parameters()))
    den = sum(float(x.double().pow(2).sum()) for x in a.parameters())
    return (num / den) ** 0.5
e1, l1 = run('1'); e2, l2 = run('1'); e0, l0 = run('0'); g1, lg1 = run('1', True); g0, lg0 = run('0', True)
print('eager(stream) vs eager(stream):', dev(e1, e2), l1, l2)
print('eager(stream) vs eager(old)   :', dev(e1, e0), l0)
print('eager(stream) vs graph(stream):', dev(e1, g1), lg1)
print('eager(old)    vs graph(old)   :', dev(e0, g0), lg0)
print('graph(stream) vs graph(old)   :', dev(g1, g0))
