"""GroupFree3D on the MI355X hot path: the reference's second detector
(detection/GroupFree3D/models/) -- the same PointNet++ backbone (fp2 -> 288 channels) followed
by k-closest-point sampling, a six-layer transformer decoder and per-layer prediction heads.
The point-cloud stages run on this package's HIP kernels (fused set abstraction, FPS, ball
query, three_nn / three_interpolate, gather_points for the KPS gather); the decoder uses
torch's attention / linear layers."""
from .detector import (GroupFreeDetector, GroupFreeDetector_DA,  # noqa: F401
                       GroupFreeDetector_DA_jitter)
from .loss_helper import get_loss, get_loss_DA, get_loss_DA_jitter, get_loss_weak  # noqa: F401
