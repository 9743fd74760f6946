"""Where do the GPU step and the CPU-oracle step of smoke() part ways?"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import oracle
from backtoreality_amd.pointnet2 import _ext, pointnet2_utils
from backtoreality_amd.votenet import config, loss_helper, synthetic, votenet
cfg = config.scannet_md40()
batch = synthetic.make_batch(0, 2, 4096, cfg)
def step(device, ext):
    pointnet2_utils._ext = ext
    torch.manual_seed(0)
    net = votenet.VoteNet(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster, cfg.mean_size_arr, input_feature_dim=1, num_proposal=256).to(device)
    b = {k: v.to(device) for k, v in batch.items()}
    end = net({'point_clouds': b['point_clouds']}); end.update(b)
    loss, end = loss_helper.get_loss(end, cfg); loss.backward()
    return {k: end[k].detach().cpu() for k in ('aggregated_vote_inds', 'vote_xyz', 'fp2_features', 'loss', 'vote_loss', 'objectness_loss', 'center_loss', 'sem_cls_loss')}
g = step(torch.device('cuda:0'), _ext); c = step(torch.device('cpu'), oracle.ext_cpu)
pointnet2_utils._ext = _ext
same = (g['aggregated_vote_inds'] == c['aggregated_vote_inds']).float().mean()
print("aggregated_vote_inds equal fraction %.4f" % float(same))
for k in ('vote_xyz', 'fp2_features'):
    print(k, float((g[k] - c[k]).abs().max() / c[k].abs().max()))
for k in ('loss', 'vote_loss', 'objectness_loss', 'center_loss', 'sem_cls_loss'):
    print(k, float(g[k]), float(c[k]))
