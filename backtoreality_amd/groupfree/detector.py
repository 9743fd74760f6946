"""GroupFreeDetector (detection/GroupFree3D/models/detector.py:15-232), same constructor,
attribute names (state-dict keys) and end_points keys."""
import torch
import torch.nn as nn

from ..pointnet2 import fused_mlp
from ..votenet.backbone_module import Pointnet2Backbone
from ..votenet.votenet_da import _run_head, grad_reverse
from .modules import (FPSModule, GeneralSamplingModule, PointsObjClsModule,
                      PositionEmbeddingLearned, PredictHead)
from . import fused_stack
from .transformer import TransformerDecoderLayer


def _project(conv, x):
    """A bare 1x1 Conv1d; on the GPU through the point-wise chain kernels, whose result carries
    the channel-last rows the decoder layers read."""
    out = fused_mlp.run_chain(x, [(conv, None, False)])
    return out if out is not None else conv(x)


class GroupFreeDetector(nn.Module):
    """Backbone -> seed objectness -> top-k (KPS) or FPS query points -> proposal head ->
    `num_decoder_layers` x (decoder layer + prediction head), each head refining the previous
    head's (detached) box as the next layer's query position."""

    center_refine = False   # GroupFreeDetector_DA_jitter: backbone with the centre head

    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                 input_feature_dim=0, width=1, bn_momentum=0.1, sync_bn=False, num_proposal=128,
                 sampling='kps', dropout=0.1, activation="relu", nhead=8, num_decoder_layers=6,
                 dim_feedforward=2048, self_position_embedding='xyz_learned',
                 cross_position_embedding='xyz_learned'):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        assert mean_size_arr.shape[0] == self.num_size_cluster
        self.input_feature_dim = input_feature_dim
        self.num_proposal = num_proposal
        self.bn_momentum = bn_momentum
        self.sync_bn = sync_bn
        self.width = width
        self.nhead = nhead
        self.sampling = sampling
        self.num_decoder_layers = num_decoder_layers
        self.dim_feedforward = dim_feedforward
        self.self_position_embedding = self_position_embedding
        self.cross_position_embedding = cross_position_embedding

        # (the reference's detector passes `width` on and leaves the backbone's depth at its
        # default 2, detector.py:57; both are accepted by the backbone here)
        self.backbone_net = Pointnet2Backbone(input_feature_dim=self.input_feature_dim,
                                              fp2_out=288, center_refine=self.center_refine,
                                              num_class=num_class, width=width)
        if self.sampling == 'fps':
            self.fps_module = FPSModule(num_proposal)
        elif self.sampling == 'kps':
            self.points_obj_cls = PointsObjClsModule(288)
            self.gsample_module = GeneralSamplingModule()
        else:
            raise NotImplementedError
        head = lambda: PredictHead(num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                                   num_proposal, 288)
        self.proposal_head = head()
        if self.num_decoder_layers <= 0:
            return

        self.decoder_key_proj = nn.Conv1d(288, 288, kernel_size=1)
        self.decoder_query_proj = nn.Conv1d(288, 288, kernel_size=1)

        n = self.num_decoder_layers
        self_dim = {'none': None, 'xyz_learned': 3, 'loc_learned': 6}
        cross_dim = {'none': None, 'xyz_learned': 3}
        if self_position_embedding not in self_dim:
            raise NotImplementedError(
                "self_position_embedding not supported %s" % self_position_embedding)
        if cross_position_embedding not in cross_dim:
            raise NotImplementedError(
                "cross_position_embedding not supported %s" % cross_position_embedding)
        if self_dim[self_position_embedding] is None:
            self.decoder_self_posembeds = [None] * n
        else:
            self.decoder_self_posembeds = nn.ModuleList(
                PositionEmbeddingLearned(self_dim[self_position_embedding], 288) for _ in range(n))
        if cross_dim[cross_position_embedding] is None:
            self.decoder_cross_posembeds = [None] * n
        else:
            self.decoder_cross_posembeds = nn.ModuleList(
                PositionEmbeddingLearned(3, 288) for _ in range(n))

        self.decoder = nn.ModuleList(
            TransformerDecoderLayer(288, nhead, dim_feedforward, dropout, activation,
                                    self_posembed=self.decoder_self_posembeds[i],
                                    cross_posembed=self.decoder_cross_posembeds[i])
            for i in range(n))
        self.prediction_heads = nn.ModuleList(head() for _ in range(n))

        self.init_weights()
        self.init_bn_momentum()
        if self.sync_bn:
            nn.SyncBatchNorm.convert_sync_batchnorm(self)

    def forward(self, inputs, center_xyz=None, center_cls=None):
        """inputs {'point_clouds': (B, N, 3 + input_feature_dim)} -> end_points."""
        end_points = self._backbone(inputs, center_xyz, center_cls)
        points_xyz = end_points['fp2_xyz']
        points_features = end_points['fp2_features']
        end_points['seed_inds'] = end_points['fp2_inds']
        end_points['seed_xyz'] = points_xyz
        end_points['seed_features'] = points_features
        if self.sampling == 'fps':
            xyz, features, sample_inds = self.fps_module(points_xyz, points_features)
        else:
            logits = self.points_obj_cls(points_features)            # (B, 1, num_seed)
            end_points['seeds_obj_cls_logits'] = logits
            scores = torch.sigmoid(logits).squeeze(1)
            sample_inds = torch.topk(scores, self.num_proposal)[1].int()
            xyz, features, sample_inds = self.gsample_module(points_xyz, points_features,
                                                             sample_inds)
        cluster_feature, cluster_xyz = features, xyz
        end_points['query_points_xyz'] = xyz
        end_points['query_points_feature'] = features
        end_points['query_points_sample_inds'] = sample_inds

        center, size = self.proposal_head(cluster_feature, base_xyz=cluster_xyz,
                                          end_points=end_points, prefix='proposal_')
        query_pos = self._query_pos(center, size)
        if self.num_decoder_layers <= 0:
            return self._finish(end_points)

        query = _project(self.decoder_query_proj, cluster_feature)
        key = _project(self.decoder_key_proj, points_features)
        key_pos = None if self.cross_position_embedding == 'none' else points_xyz
        # the whole loop as one autograd node (csrc/gf_stack.hip) when it covers the configuration
        if fused_stack.run(self, query, key, query_pos, key_pos, cluster_xyz, end_points):
            return self._finish(end_points)
        for i in range(self.num_decoder_layers):
            prefix = 'last_' if i == self.num_decoder_layers - 1 else '%dhead_' % i
            query = self.decoder[i](query, key, query_pos, key_pos)
            self._after_decoder_layer(prefix, query, end_points)
            center, size = self.prediction_heads[i](query, base_xyz=cluster_xyz,
                                                    end_points=end_points, prefix=prefix)
            query_pos = self._query_pos(center, size)
        return self._finish(end_points)

    def _query_pos(self, center, size):
        """The next decoder layer's query position from a head's (center, pred_size): detached
        copies, concatenated for the 'loc_learned' embedding (detector.py:204-230)."""
        if self.self_position_embedding == 'none':
            return None
        if self.self_position_embedding == 'xyz_learned':
            return center.detach().clone()
        fused = getattr(center, '_btr_query_pos', None)   # written by the head decode kernel
        if fused is not None and fused[1] == center._version and fused[2] is size and \
                fused[3] == size._version:   # (neither tensor edited in place since)
            return fused[0]
        return torch.cat([center.detach().clone(), size.detach().clone()], -1)

    def _backbone(self, inputs, center_xyz, center_cls):
        # inputs['sampling']: optional handle of backbone_net.prefetch_sampling (the pyramid of
        # this cloud computed ahead, e.g. under the previous step's backward)
        return self.backbone_net(inputs['point_clouds'], {}, sampling=inputs.get('sampling'))

    # the hooks below only act on the last layer's output (fused_stack.run calls them for it alone)
    _hook_last_only = True

    def _after_decoder_layer(self, prefix, query, end_points):
        """Hook for the domain-adaptation variant."""

    def _finish(self, end_points):
        return end_points

    def init_weights(self):
        # xavier on every decoder matrix, position-embedding convolutions included
        for p in self.decoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def init_bn_momentum(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = self.bn_momentum


class GroupFreeDetector_DA(GroupFreeDetector):
    """GroupFreeDetector with the two domain classifiers of Back-to-Reality behind gradient
    reversal (detection/GroupFree3D/models/detector_DA.py:56-302): a global one on the seed
    features (`global_d_pred` (B,2)) and a local one on the last decoder layer's query
    features (`last_local_d_pred` (B,1,num_proposal), after a sigmoid).  The classifiers are
    created after the weight / BN-momentum initialisation, like the reference."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.global_netD1 = nn.Sequential(
            nn.Conv1d(288, 256, 1), nn.BatchNorm1d(256), nn.ReLU(),
            nn.Conv1d(256, 128, 1), nn.BatchNorm1d(128), nn.ReLU())
        self.global_netD2 = nn.Linear(128, 2)
        self.decoder_netD = nn.Sequential(
            nn.Conv1d(288, 128, 1), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Conv1d(128, 128, 1), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Conv1d(128, 1, 1))

    _hook_last_only = True   # (declared beside the hook it describes: fused_stack._hook_last_only)

    def _after_decoder_layer(self, prefix, query, end_points):
        if prefix == 'last_':
            end_points[prefix + 'local_d_pred'] = torch.sigmoid(
                _run_head(self.decoder_netD, grad_reverse(query)))

    def _finish(self, end_points):
        g = _run_head(self.global_netD1, grad_reverse(end_points['seed_features']))  # (B,128,num_seed)
        end_points['global_d_pred'] = self.global_netD2(torch.mean(g, dim=2))
        return end_points


class GroupFreeDetector_DA_jitter(GroupFreeDetector_DA):
    """CenterRefine variant (detector_DA.py:317-575): the backbone additionally pools seed
    features around the 64 (noisy) ground-truth centres (`ctjt_head`), and `jitter_net`
    regresses each centre's displacement from them: forward(inputs, center_xyz (B,64,3),
    center_cls (B,64)) adds 'center_features' (B,128+num_class,64) and 'jitter_pred'
    (B,3,64)."""
    center_refine = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.jitter_net = nn.Sequential(
            nn.Conv1d(128 + self.num_class, 64, 1), nn.BatchNorm1d(64), nn.ReLU(),
            nn.Conv1d(64, 3, 1))

    def _backbone(self, inputs, center_xyz, center_cls):
        end_points = self.backbone_net(inputs['point_clouds'], {}, center_xyz=center_xyz,
                                       center_cls=center_cls)
        if center_xyz is not None:
            end_points['jitter_pred'] = _run_head(self.jitter_net, end_points['center_features'])
        return end_points
