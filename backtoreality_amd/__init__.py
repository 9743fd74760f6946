"""backtoreality_amd -- MI355X-native implementation of the VoteNet point-cloud hot path of
wyf-ACCEPT/BackToReality: the PointNet++ set-abstraction stack and vote aggregation.

  csrc/       hand-written HIP kernels for gfx950 + the C ABI (include/btr_pointnet2.h)
  lib/        libbtr_pointnet2.so (built in-tree by build.py / __graft_entry__.build())
  pointnet2/  drop-in mirror of the reference's pointnet2 Python interface
              (_ext, pointnet2_utils, pointnet2_modules, pytorch_utils)
  votenet/    harness: VoteNet assembly, loss, synthetic scenes and the data-parallel step
"""
__version__ = "0.1.0"
