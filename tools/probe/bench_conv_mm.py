"""Probe: every 1x1 Conv1d / Conv2d forward as torch.matmul (one strided-batched GEMM) instead
of MIOpen's per-sample GEMMs; then the normal bench."""
import runpy
import sys

sys.path.insert(0, '/root/repo')
import torch
import torch.nn as nn

_c1, _c2 = nn.Conv1d.forward, nn.Conv2d.forward


def conv1d_forward(self, x):
    if self.kernel_size == (1,) and self.stride == (1,) and self.padding == (0,) and self.groups == 1 and x.is_cuda:
        y = torch.matmul(self.weight[:, :, 0], x)
        return y if self.bias is None else y + self.bias[:, None]
    return _c1(self, x)


def conv2d_forward(self, x):
    if (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
            and self.groups == 1 and x.is_cuda and x.shape[3] == 1):
        y = torch.matmul(self.weight[:, :, 0, 0], x[..., 0]).unsqueeze(-1)
        return y if self.bias is None else y + self.bias[None, :, None, None]
    return _c2(self, x)


nn.Conv1d.forward = conv1d_forward
nn.Conv2d.forward = conv2d_forward
sys.argv = ['bench.py', '--steps', '30', '--warmup', '5', '--no-cpu-baseline']
runpy.run_path('/root/repo/bench.py', run_name='__main__')
