"""BoxNet, the `--model boxnet` ablation of the reference (detection/Votenet/models/boxnet.py,
loss_helper_boxnet.py): VoteNet without the voting stage -- the proposal module samples and
groups the SEED points directly -- and without the vote loss; objectness labels come from
whether the sampled seed lies on an object, with no ignore zone."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import loss_helper
from .backbone_module import Pointnet2Backbone
from .proposal_module import ProposalModule

OBJECTNESS_CLS_WEIGHTS = [0.2, 0.8]  # loss_helper_boxnet.py:18


class BoxNet(nn.Module):
    """boxnet.py:20-85; same constructor and sub-module names (`backbone_net`, `pnet`)."""

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
        self.pnet = ProposalModule(num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                                   num_proposal, sampling)

    def forward(self, inputs):
        end_points = self.backbone_net(inputs['point_clouds'], {},
                                       sampling=inputs.get('sampling'))
        xyz = end_points['fp2_xyz']
        features = end_points['fp2_features']
        end_points['seed_inds'] = end_points['fp2_inds']
        end_points['seed_xyz'] = xyz
        end_points['seed_features'] = features
        return self.pnet(xyz, features, end_points)


def compute_objectness_loss(end_points):
    """loss_helper_boxnet.py:20-60: label = the vote mask of the seed a proposal was sampled
    at; every proposal counts (no ignore zone); assignment = nearest ground-truth centre."""
    agg = end_points['aggregated_vote_xyz']
    gt_center = end_points['center_label'][:, :, 0:3]
    d = torch.sum((agg.unsqueeze(2) - gt_center.unsqueeze(1)) ** 2, dim=-1)
    assignment = torch.argmin(d, dim=2)
    seed_inds = end_points['seed_inds'].long()
    end_points['seed_labels'] = torch.gather(end_points['vote_label_mask'], 1, seed_inds)
    label = torch.gather(end_points['seed_labels'], 1, end_points['aggregated_vote_inds'].long())
    mask = torch.ones(label.shape, dtype=torch.float32, device=agg.device)
    w = torch.tensor(OBJECTNESS_CLS_WEIGHTS, dtype=torch.float32, device=agg.device)
    loss = F.cross_entropy(end_points['objectness_scores'].transpose(2, 1), label, weight=w,
                           reduction='none')
    loss = torch.sum(loss * mask) / (torch.sum(mask) + 1e-6)
    return loss, label, mask, assignment


def get_loss(end_points, config):
    """loss_helper_boxnet.py:62-121: 0.5 objectness + box + 0.1 semantic, times 10."""
    loss, label, mask, assignment = compute_objectness_loss(end_points)
    end_points['objectness_loss'] = loss
    end_points['objectness_label'] = label
    end_points['objectness_mask'] = mask
    end_points['object_assignment'] = assignment
    total = float(label.shape[0] * label.shape[1])
    end_points['pos_ratio'] = torch.sum(label.float()) / total
    end_points['neg_ratio'] = torch.sum(mask) / total - end_points['pos_ratio']
    (center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss,
     sem_cls_loss) = loss_helper.compute_box_and_sem_cls_loss(end_points, config)
    end_points['center_loss'] = center_loss
    end_points['heading_cls_loss'] = heading_cls_loss
    end_points['heading_reg_loss'] = heading_reg_loss
    end_points['size_cls_loss'] = size_cls_loss
    end_points['size_reg_loss'] = size_reg_loss
    end_points['sem_cls_loss'] = sem_cls_loss
    box_loss = center_loss + 0.1 * heading_cls_loss + heading_reg_loss + 0.1 * size_cls_loss + \
        size_reg_loss
    end_points['box_loss'] = box_loss
    total_loss = (0.5 * loss + box_loss + 0.1 * sem_cls_loss) * 10
    end_points['loss'] = total_loss
    obj_pred = torch.argmax(end_points['objectness_scores'], 2)
    end_points['obj_acc'] = torch.sum((obj_pred == label.long()).float() * mask) / \
        (torch.sum(mask) + 1e-6)
    return total_loss, end_points
