import sys, itertools, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from backtoreality_amd import groupfree
from backtoreality_amd.groupfree import fused_stack, fused_attention
from backtoreality_amd.pointnet2 import _ext
from backtoreality_amd.votenet import config, synthetic
import test_gf_stack_gpu as T
cuda=torch.device('cuda:0')
cfg = config.scannet_md40()
batch = synthetic.make_batch(3, 2, 8192, cfg, use_height=False, device=cuda)
torch.manual_seed(0)
net = groupfree.GroupFreeDetector(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster, cfg.mean_size_arr,
          input_feature_dim=0, num_proposal=256, dropout=0.0, self_position_embedding='loc_learned',
          cross_position_embedding='xyz_learned').to(cuda)
ref=None
for it in range(6):
    for p in net.parameters(): p.grad=None
    ep = net({'point_clouds': batch['point_clouds']})
    ep.update(batch)
    loss, ep = groupfree.get_loss(ep, cfg, **T.LOSS_ARGS)
    loss.backward()
    torch.cuda.synchronize()
    cur = {'loss': loss.detach().clone(), 'center': ep['last_center'].detach().clone()}
    for k in ('0head_center','2head_center','4head_center'): cur[k]=ep[k].detach().clone()
    for k,p in net.named_parameters():
        if p.grad is not None and k.startswith(('decoder.','prediction_heads.')): cur['g:'+k]=p.grad.detach().clone()
    if ref is None: ref=cur
    else:
        bad=[(k, float((cur[k]-ref[k]).abs().max())) for k in cur if not torch.equal(cur[k],ref[k])]
        print(it, float(loss), len(bad), bad[:6])
print(_ext.graph_stats())
