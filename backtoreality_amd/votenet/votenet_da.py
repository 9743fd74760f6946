"""VoteNet with domain-adaptation heads, the network of the "Back-to-Reality" recipe
(detection/Votenet/models/votenet_DA.py:31-176): VoteNet + a global domain classifier on the
seed features and a local one on the aggregated vote features, both behind a gradient-reversal
layer.  Sub-module names match the reference (`global_netD1`, `global_netD2`, `local_netD`)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from ..pointnet2 import fused_mlp
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


def _run_head(seq, x):
    """A domain classifier / regressor head -- nn.Sequential of (Conv1d k=1, BatchNorm1d, ReLU)
    groups with an optional bare Conv1d at the end (votenet_DA.py:91-99, 113-121, 250-262) --
    as ONE library call per direction on the point-wise chain kernels (csrc/sa_layer.hip
    btr_pm_chain_*) instead of 3 stock ops per layer and MIOpen's per-sample convolution
    kernels in the backward; the parameters stay the Sequential's own (state-dict keys of the
    reference).  Falls back to the stock modules when the chain path does not apply (CPU, eval
    mode, BTR_FUSED_MLP=0)."""
    mods = list(seq)
    chain, i = [], 0
    while i < len(mods):
        if i + 2 < len(mods) and isinstance(mods[i], nn.Conv1d) and \
                isinstance(mods[i + 1], nn.BatchNorm1d) and isinstance(mods[i + 2], nn.ReLU):
            chain.append((mods[i], mods[i + 1], True))
            i += 3
        elif i == len(mods) - 1 and isinstance(mods[i], nn.Conv1d):
            chain.append((mods[i], None, False))
            i += 1
        else:
            chain = None
            break
    out = fused_mlp.run_chain(x, chain) if chain else None
    return out if out is not None else seq(x)


class VoteNet_DA(nn.Module):
    center_refine = False  # VoteNet_DA_jitter: backbone with the centre head

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

        self.backbone_net = Pointnet2Backbone(input_feature_dim=self.input_feature_dim,
                                              center_refine=self.center_refine,
                                              num_class=num_class)
        self.vgen = VotingModule(self.vote_factor, 256)
        self.pnet = ProposalModule(num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                                   num_proposal, sampling)
        # global domain prediction (:91-99) and local domain prediction (:113-121)
        self.global_netD1 = nn.Sequential(*(_conv_bn_relu(256, 256) + _conv_bn_relu(256, 128)))
        self.global_netD2 = nn.Linear(128, 2)
        self.local_netD = nn.Sequential(*(_conv_bn_relu(128, 128) + _conv_bn_relu(128, 128) +
                                          [nn.Conv1d(128, 1, 1)]))

    def forward(self, inputs, center_xyz=None, center_cls=None):
        return self.forward_head(self.forward_backbone(inputs, center_xyz, center_cls),
                                 center_xyz)

    # The forward in two stages (not in the reference): the backbone -- the large kernels -- and
    # everything behind it (voting, vote aggregation, proposal head, domain classifiers: ~120
    # launches of 5 - 20 us that leave most of the chip idle).  train.train_step_br runs the
    # source branch's second stage on a side stream beside the target branch's first.
    def forward_backbone(self, inputs, center_xyz=None, center_cls=None):
        return self.backbone_net(inputs['point_clouds'], {}, sampling=inputs.get('sampling'),
                                 center_xyz=center_xyz, center_cls=center_cls)

    def forward_head(self, end_points, center_xyz=None):
        self._center_heads(end_points, center_xyz, before_voting=True)
        xyz = end_points['fp2_xyz']
        features = end_points['fp2_features']
        end_points['seed_inds'] = end_points['fp2_inds']
        end_points['seed_xyz'] = xyz
        end_points['seed_features'] = features

        # (votes + the L2 normalisation of their features, models/votenet.py:97-99, in one call)
        xyz, features = self.vgen(xyz, features, normalize=True)
        end_points['vote_xyz'] = xyz
        end_points['vote_features'] = features
        end_points = self.pnet(xyz, features, end_points)

        g = _run_head(self.global_netD1, grad_reverse(end_points['seed_features']))  # (B,128,1024)
        end_points['global_d_pred'] = self.global_netD2(torch.mean(g, dim=2))  # (B,2)
        local = _run_head(self.local_netD, grad_reverse(end_points['aggregated_vote_features']))
        end_points['local_d_pred'] = torch.sigmoid(local)  # (B,1,num_proposal)
        self._center_heads(end_points, center_xyz, before_voting=False)
        return end_points

    def _center_heads(self, end_points, center_xyz, before_voting):
        pass


class VoteNet_DA_jitter(VoteNet_DA):
    """CenterRefine network (votenet_DA.py:179-333, train_Votenet_BR_CenterRefine.py:189):
    VoteNet_DA over the centre-head backbone plus `jitter_net` (regresses the displacement of
    each noisy GT centre from its pooled features + class one-hot) and `jitter_netD` (domain
    classifier on the same features behind the gradient-reversal layer).  Sub-module names and
    creation order match the reference."""
    center_refine = True

    def __init__(self, num_class, *args, **kwargs):
        super().__init__(num_class, *args, **kwargs)
        cf = 128 + num_class
        self.jitter_netD = nn.Sequential(*(_conv_bn_relu(cf, 128) + _conv_bn_relu(128, 128) +
                                           [nn.Conv1d(128, 1, 1)]))
        self.jitter_net = nn.Sequential(*(_conv_bn_relu(cf, 64) + [nn.Conv1d(64, 3, 1)]))

    def _center_heads(self, end_points, center_xyz, before_voting):
        if center_xyz is None:
            return
        if before_voting:  # votenet_DA.py:291-292
            end_points['jitter_pred'] = _run_head(self.jitter_net,
                                                  end_points['center_features'])  # B,3,64
        else:              # :324-327
            d = _run_head(self.jitter_netD, grad_reverse(end_points['center_features']))
            end_points['jitter_d_pred'] = torch.sigmoid(d)  # B,1,64
