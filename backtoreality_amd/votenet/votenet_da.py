"""VoteNet with domain-adaptation heads, the network of the "Back-to-Reality" recipe
(detection/Votenet/models/votenet_DA.py:31-176): VoteNet + a global domain classifier on the
seed features and a local one on the aggregated vote features, both behind a gradient-reversal
layer.  Sub-module names match the reference (`global_netD1`, `global_netD2`, `local_netD`)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from .backbone_module import Pointnet2Backbone
from .proposal_module import ProposalModule
from .voting_module import VotingModule


class GradReverse(Function):
    """Identity forward, negated gradient backward (votenet_DA.py:31-40)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output * -1.0


def grad_reverse(x):
    return GradReverse.apply(x)


def _conv_bn_relu(cin, cout):
    return [nn.Conv1d(cin, cout, 1), nn.BatchNorm1d(cout), nn.ReLU()]


class VoteNet_DA(nn.Module):
    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                 input_feature_dim=0, num_proposal=128, vote_factor=1, sampling='vote_fps'):
        super().__init__()
        assert mean_size_arr.shape[0] == num_size_cluster
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.input_feature_dim = input_feature_dim
        self.num_proposal = num_proposal
        self.vote_factor = vote_factor
        self.sampling = sampling

        self.backbone_net = Pointnet2Backbone(input_feature_dim=self.input_feature_dim)
        self.vgen = VotingModule(self.vote_factor, 256)
        self.pnet = ProposalModule(num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                                   num_proposal, sampling)
        # global domain prediction (:91-99) and local domain prediction (:113-121)
        self.global_netD1 = nn.Sequential(*(_conv_bn_relu(256, 256) + _conv_bn_relu(256, 128)))
        self.global_netD2 = nn.Linear(128, 2)
        self.local_netD = nn.Sequential(*(_conv_bn_relu(128, 128) + _conv_bn_relu(128, 128) +
                                          [nn.Conv1d(128, 1, 1)]))

    def forward(self, inputs):
        end_points = self.backbone_net(inputs['point_clouds'], {},
                                       sampling=inputs.get('sampling'))
        xyz = end_points['fp2_xyz']
        features = end_points['fp2_features']
        end_points['seed_inds'] = end_points['fp2_inds']
        end_points['seed_xyz'] = xyz
        end_points['seed_features'] = features

        xyz, features = self.vgen(xyz, features)
        features = features.div(torch.norm(features, p=2, dim=1).unsqueeze(1))
        end_points['vote_xyz'] = xyz
        end_points['vote_features'] = features
        end_points = self.pnet(xyz, features, end_points)

        g = self.global_netD1(grad_reverse(end_points['seed_features']))  # (B,128,1024)
        end_points['global_d_pred'] = self.global_netD2(torch.mean(g, dim=2))  # (B,2)
        local = self.local_netD(grad_reverse(end_points['aggregated_vote_features']))
        end_points['local_d_pred'] = torch.sigmoid(local)  # (B,1,num_proposal)
        return end_points
