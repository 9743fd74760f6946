import sys
sys.path.insert(0, '/root/repo')
import torch
from backtoreality_amd.pointnet2 import _ext
from tools.bench_ops import scenes
xyz = scenes(8, 40000)
_ext.furthest_point_sampling(xyz, 2048)
torch.cuda.synchronize()
