"""VoteNet assembly (detection/Votenet/models/votenet.py:25-100): backbone -> voting ->
L2-normalised vote features -> proposal module.  Sub-module names (`backbone_net`, `vgen`,
`pnet`) match the reference so its checkpoints' `model_state_dict` loads."""
import torch.nn as nn

from .backbone_module import Pointnet2Backbone
from .proposal_module import ProposalModule
from .voting_module import VotingModule


class VoteNet(nn.Module):
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

    def forward(self, inputs):
        """inputs['point_clouds'] (B, N, 3 + input_feature_dim) -> end_points dict."""
        end_points = self.backbone_net(inputs['point_clouds'], {},
                                       sampling=inputs.get('sampling'))

        xyz = end_points['fp2_xyz']
        features = end_points['fp2_features']
        end_points['seed_inds'] = end_points['fp2_inds']
        end_points['seed_xyz'] = xyz
        end_points['seed_features'] = features

        # (votes + the L2 normalisation of their features, models/votenet.py:97-99, in one call)
        xyz, features = self.vgen(xyz, features, normalize=True)
        end_points['vote_xyz'] = xyz
        end_points['vote_features'] = features

        return self.pnet(xyz, features, end_points)
