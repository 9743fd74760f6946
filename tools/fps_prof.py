import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.pointnet2 import _ext
from tools.bench_ops import scenes
xyz = scenes(8, 40000)
_ext.furthest_point_sampling(xyz, 2048); torch.cuda.synchronize()
os.environ["BTR_FPS_PROF"] = "1"
_ext.furthest_point_sampling(xyz, 2048); torch.cuda.synchronize()
