"""PointNet++ backbone of VoteNet: four set-abstraction layers and two feature-propagation
layers with the hyper-parameters hard-coded in the reference
(detection/Votenet/models/backbone_module.py:35-72) and the same attribute names
(`sa1..sa4`, `fp1`, `fp2`), so the state-dict keys are `backbone_net.sa1.mlp_module.layer0...`.
"""
import os
import weakref

import torch
import torch.nn as nn

from ..pointnet2 import _ext, fused_backbone, pointnet2_utils
from ..pointnet2.pointnet2_modules import (PointnetFPModule, PointnetSAModuleCenters,
                                           PointnetSAModuleVotes)

# backbone module -> {input shape: fused_backbone.Entry}; weak, so that modules stay picklable /
# deep-copyable (an Entry holds ctypes structures with pointers)
_NATIVE_ENTRIES = weakref.WeakKeyDictionary()
_SIDE_STREAMS = weakref.WeakKeyDictionary()   # backbone module -> {(name, device): stream}

# (npoint, radius, nsample, mlp-after-input) per SA layer -- backbone_module.py:35-69
SA_SPECS = (
    (2048, 0.2, 64, (64, 64, 128)),
    (1024, 0.4, 32, (128, 128, 256)),
    (512, 0.8, 16, (128, 128, 256)),
    (256, 1.2, 16, (128, 128, 256)),
)


class Pointnet2Backbone(nn.Module):
    """center_refine=True is the reference's `Pointnet2Backbone_jitter`
    (backbone_module.py:136-262): an extra set-abstraction head `ctjt_head` pooled around GIVEN
    centres (the noisy GT box centres) whose output, concatenated with the one-hot class of
    each centre, feeds the centre-jitter regressor of the CenterRefine recipe."""

    def __init__(self, input_feature_dim=0, fp2_out=256, center_refine=False, num_class=22,
                 width=1, depth=2):
        """width / depth: the channel multiplier and the number of hidden layers per
        set-abstraction MLP of GroupFree3D's backbone (detection/GroupFree3D/models/
        backbone_module.py:33-75: mlp = [in] + [h * width] * depth + [out * width], FP modules
        [512 w, 256 w, 256 w] and [512 w, 256 w, fp2_out]); VoteNet's is width = 1, depth = 2."""
        super().__init__()
        self.width, self.depth = int(width), int(depth)
        assert self.width >= 1 and self.depth >= 1
        w = self.width
        cin = input_feature_dim
        for i, (npoint, radius, nsample, widths) in enumerate(SA_SPECS, start=1):
            hidden, out = widths[0] * w, widths[-1] * w
            setattr(self, "sa%d" % i, PointnetSAModuleVotes(
                npoint=npoint, radius=radius, nsample=nsample,
                mlp=[cin] + [hidden] * self.depth + [out], use_xyz=True, normalize_xyz=True))
            cin = out
        self.fp1 = PointnetFPModule(mlp=[256 * w + 256 * w, 256 * w, 256 * w])
        self.fp2 = PointnetFPModule(mlp=[256 * w + 256 * w, 256 * w, fp2_out])
        self.num_class = num_class
        if center_refine:  # backbone_module.py:188-195
            self.ctjt_head = PointnetSAModuleCenters(npoint=64, radius=0.8, nsample=16,
                                                     mlp=[fp2_out, 128], use_xyz=True,
                                                     normalize_xyz=False)

    @staticmethod
    def _break_up_pc(pc):
        # the coordinate tensor is kept with the point cloud it was cut from: a prefetched
        # sampling pyramid and the forward that consumes it then see the SAME xyz object, and
        # with it the spatial sort the FPS attached to it (pointnet2/_ext.py _fps)
        # (never while a HIP graph is captured: the version test would be evaluated once, at
        # capture time, and every replay would read the coordinates of the capture batch)
        cached = getattr(pc, "_btr_xyz", None)
        capturing = pc.is_cuda and torch.cuda.is_current_stream_capturing()
        if cached is not None and cached[1] == pc._version and not capturing:
            xyz = cached[0]
        else:
            xyz = pc[..., 0:3].contiguous()
            if not capturing:
                pc._btr_xyz = (xyz, pc._version)
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    # ------------------------------------------------------------ whole-backbone library calls
    def _sa_fp(self):
        return ([self.sa1, self.sa2, self.sa3, self.sa4], [self.fp1, self.fp2])

    def _native_entry(self, pointcloud):
        """Description + plan of the whole-backbone calls (pointnet2/fused_backbone.py) for this
        input shape, or None when the configuration is not covered (then: layer by layer)."""
        sa, fp = self._sa_fp()
        if not fused_backbone.supported(sa, fp, pointcloud):
            return None
        cache = _NATIVE_ENTRIES.get(self)
        if cache is None:
            cache = _NATIVE_ENTRIES[self] = {}
        key = (tuple(pointcloud.shape), pointcloud.device, fused_backbone.fused_sa._sa_options(),
               fused_backbone._ext.fmad())
        ent = cache.get(key)
        if ent is None:
            B, N, W = pointcloud.shape
            ent = cache[key] = fused_backbone.Entry(sa, fp, B, N, W - 3)
        return ent

    def arm_fork_event(self, pointcloud):
        """Software pipelining: returns an event that the NEXT native forward of this backbone
        records behind SA level 2 (btr_backbone_fork_event), or None when the native whole-
        backbone path does not apply.  `prefetch_sampling(next_cloud, after=event)` then starts
        the next pyramid's ~2 000-step FPS chain beside SA3 .. loss .. backward -- strings of
        short latency-bound kernels -- instead of beside SA1 / SA2, whose forward is HBM-bound
        and loses most to a co-runner (votenet/train.py train_step)."""
        if not pointcloud.is_cuda or os.environ.get("BTR_OVERLAP_FPS", "1") == "0":
            return None
        entry = self._native_entry(pointcloud)
        if entry is None or torch.cuda.is_current_stream_capturing():
            return None
        ev = torch.cuda.Event()
        ev.record()      # (creates the handle; the library records it again in place)
        entry.lib.btr_backbone_fork_event(ev.cuda_event, int(os.environ.get("BTR_FORK_LEVEL", "2")))
        return ev

    def disarm_fork_event(self, pointcloud):
        """Drop an armed fork event (arm_fork_event) that no native forward consumed."""
        entry = self._native_entry(pointcloud) if pointcloud.is_cuda else None
        if entry is not None:
            entry.lib.btr_backbone_fork_event(None, 0)

    def prefetch_sampling(self, pointcloud, after=None, slot=0):
        """Start the sampling pyramid of `pointcloud` on the side stream NOW and return a
        handle to pass to forward(..., sampling=handle).  Sampling depends on coordinates
        only, so a caller that runs several forwards per step (the Back-to-Reality step runs a
        source and a target branch, train_Votenet_BR.py:277-278) can overlap the second
        branch's FPS with the first branch's forward.  Same indices as computing them inline.
        `after`: an event of the current stream the side stream waits for instead of the
        stream's whole queue (arm_fork_event).  `slot`: which of the module's prefetch streams
        (a loop that keeps TWO pyramids in flight -- batches i + 1 and i + 2 -- alternates 0 / 1:
        a pyramid is a chain of dependent steps on one CU per scene, so two of them run beside
        each other at the speed of one)."""
        if not pointcloud.is_cuda or os.environ.get("BTR_OVERLAP_FPS", "1") == "0":
            return None
        entry = self._native_entry(pointcloud)
        if entry is not None:   # one library call for the whole pyramid
            main = torch.cuda.current_stream(pointcloud.device)
            side = self._get_side_stream(pointcloud.device,
                                         "_prefetch_stream" if not slot else "_prefetch_stream%d" % slot)
            if after is not None:
                side.wait_event(after)
            else:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                handle = fused_backbone.sample(entry, pointcloud)
                # the seeds' indices (fp2_inds = the first sa2.npoint columns of sa1_inds) as a
                # dense tensor, here instead of as a copy the loss makes on the main stream
                handle.fp2_inds = handle.inds[0][:, :self.sa2.npoint].contiguous()
                handle.event = torch.cuda.Event()
                handle.event.record(side)
            handle.fp2_inds.record_stream(main)
            handle.geom.record_stream(main)
            pointcloud.record_stream(side)
            return handle
        xyz, _ = self._break_up_pc(pointcloud)
        main = torch.cuda.current_stream(xyz.device)
        # its own stream: on the stream forward() uses for levels 2-4 the prefetched pyramid
        # would queue in front of them and stall the forward it is supposed to hide under
        side = self._get_side_stream(xyz.device, "_prefetch_stream")
        side.wait_stream(main)
        npoints = [getattr(self, "sa%d" % i).npoint for i in (1, 2, 3, 4)]
        out = []
        # (the geometry extras ride on tensor attributes; a captured step only carries the index
        # tensors through its static buffers, so under capture they would be computed twice)
        geometry = (os.environ.get("BTR_PREFETCH_GEOMETRY", "1") != "0" and
                    not torch.cuda.is_current_stream_capturing())
        with torch.cuda.stream(side):
            cur = xyz
            centres = []
            for li, npoint in enumerate(npoints):
                inds = pointnet2_utils.furthest_point_sample(cur, npoint)
                inds.record_stream(main)
                # the sampled coordinates are needed here anyway (next level's input); the SA
                # layer that consumes `inds` takes them from the handle instead of gathering
                # them again on the main stream (pointnet2_modules._sample_centres)
                new_xyz = pointnet2_utils.gather_rows(cur, inds)
                new_xyz.record_stream(main)
                _ext.attach_derived(inds, "_btr_new_xyz", (new_xyz, cur), cur)
                if geometry:
                    # everything else that depends on coordinates only goes the same way: the
                    # layer's ball query ...
                    g = getattr(self, "sa%d" % (li + 1)).grouper
                    idx = pointnet2_utils.ball_query(g.radius, g.nsample, cur, new_xyz)
                    idx.record_stream(main)
                    _ext.attach_derived(new_xyz, "_btr_ball_query",
                                        (idx, cur, g.radius, g.nsample), cur)
                centres.append(new_xyz)
                ev = torch.cuda.Event()
                ev.record(side)
                out.append((inds, ev))
                cur = new_xyz
            if geometry:
                # ... and the 3-NN blend weights of the two feature-propagation modules
                # (fp1: sa3 <- sa4, fp2: sa2 <- sa3; forward() below)
                for unknown, known in ((centres[2], centres[3]), (centres[1], centres[2])):
                    idx, weight = pointnet2_utils.three_nn_weights(unknown, known)
                    idx.record_stream(main)
                    weight.record_stream(main)
                    _ext.attach_derived(unknown, "_btr_three_nn", (known, idx, weight),
                                        known)
                ev = torch.cuda.Event()
                ev.record(side)
                out[-1] = (out[-1][0], ev)   # (the last level's event also covers these)
        xyz.record_stream(side)
        return out

    def _get_side_stream(self, device, name="_side_stream"):
        # (kept outside the module's attributes: a HIP stream can be neither pickled nor
        # deep-copied, a module that has run once must still be)
        streams = _SIDE_STREAMS.get(self)
        if streams is None:
            streams = _SIDE_STREAMS[self] = {}
        key = (name, device)
        if key not in streams:
            streams[key] = _ext.new_stream(device)
        return streams[key]

    def _fps_pyramid(self, xyz):
        """Sampling indices of all four SA levels.  They depend on coordinates only, so levels
        2-4 run on a side HIP stream while the main stream is busy with SA1's grouped MLP (FPS
        is a chain of dependent steps occupying one CU per scene: nothing else can hide it).
        Returns [(inds, ready_event or None)] per level; same indices as the sequential order
        of the reference (pointnet2_modules.py:233-240 inside each SA layer)."""
        npoints = [getattr(self, "sa%d" % i).npoint for i in (1, 2, 3, 4)]
        inds1 = pointnet2_utils.furthest_point_sample(xyz, npoints[0])
        out = [(inds1, None)]
        if not xyz.is_cuda or os.environ.get("BTR_OVERLAP_FPS", "1") == "0":
            return out + [(None, None)] * 3
        main = torch.cuda.current_stream(xyz.device)
        side = self._get_side_stream(xyz.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            cur, inds = xyz, inds1
            for npoint in npoints[1:]:
                cur = pointnet2_utils.gather_rows(cur, inds)
                inds = pointnet2_utils.furthest_point_sample(cur, npoint)
                inds.record_stream(main)
                cur.record_stream(side)
                ev = torch.cuda.Event()
                ev.record(side)
                out.append((inds, ev))
        return out

    def forward(self, pointcloud: torch.Tensor, end_points=None, sampling=None, center_xyz=None,
                center_cls=None):
        """pointcloud (B, N, 3 + input_feature_dim) -> end_points with sa{1..4}_{xyz,features},
        sa1_inds, sa2_inds, fp2_{xyz,features,inds} (backbone_module.py:83-133).
        `sampling`: optional handle from prefetch_sampling(pointcloud).
        center_xyz (B,64,3) / center_cls (B,64) i64 (center_refine backbones only): adds
        'center_features' (B, 128 + num_class, 64) (backbone_module.py:257-260)."""
        end_points = end_points if end_points else {}
        if sampling is None or isinstance(sampling, fused_backbone.Sampling):
            entry = self._native_entry(pointcloud)
            if entry is not None:
                return self._forward_native(entry, pointcloud, end_points, sampling, center_xyz,
                                            center_cls)
        xyz, features = self._break_up_pc(pointcloud)
        # the coordinate slice is shared between a prefetch and the ONE forward that consumes it,
        # never across steps: a later step on the same resident tensor cuts it again
        if hasattr(pointcloud, "_btr_xyz"):
            del pointcloud._btr_xyz
        pyramid = sampling if sampling is not None else self._fps_pyramid(xyz)
        for i in (1, 2, 3, 4):
            inds, ready = pyramid[i - 1]
            if ready is not None:
                torch.cuda.current_stream(xyz.device).wait_event(ready)
            xyz, features, fps_inds = getattr(self, "sa%d" % i)(xyz, features, inds)
            if i <= 2:
                end_points["sa%d_inds" % i] = fps_inds
            end_points["sa%d_xyz" % i] = xyz
            end_points["sa%d_features" % i] = features

        features = self.fp1(end_points["sa3_xyz"], end_points["sa4_xyz"],
                            end_points["sa3_features"], end_points["sa4_features"])
        features = self.fp2(end_points["sa2_xyz"], end_points["sa3_xyz"],
                            end_points["sa2_features"], features)
        end_points["fp2_features"] = features
        end_points["fp2_xyz"] = end_points["sa2_xyz"]
        num_seed = end_points["fp2_xyz"].shape[1]
        # FPS over an FPS-ordered prefix returns 0..k-1, so the seeds' indices into the input
        # cloud are the first num_seed entries of sa1_inds (backbone_module.py:113-132)
        end_points["fp2_inds"] = end_points["sa1_inds"][:, 0:num_seed]
        return self._center_head(end_points, features, center_xyz, center_cls)

    def _forward_native(self, entry, pointcloud, end_points, sampling, center_xyz, center_cls):
        """forward() through btr_backbone_sampling / _forward (one autograd node)."""
        dev = pointcloud.device
        if hasattr(pointcloud, "_btr_xyz"):
            del pointcloud._btr_xyz
        if sampling is not None and not sampling.valid_for(entry, pointcloud):
            # computed for another tensor / shape, or the cloud was written to since: the indices
            # would describe other coordinates
            raise RuntimeError("sampling handle does not belong to this point cloud (another "
                               "tensor, another shape, or modified in place since "
                               "prefetch_sampling)")
        if sampling is None:
            # levels 2.. and the 3-NN weights on the side stream, under SA1's MLP
            side = None
            if os.environ.get("BTR_OVERLAP_FPS", "1") != "0":
                side = self._get_side_stream(dev)
            sampling = fused_backbone.sample(entry, pointcloud, side=side)
        elif sampling.event is not None:
            torch.cuda.current_stream(dev).wait_event(sampling.event)
        outs = fused_backbone.FusedBackboneFn.apply(pointcloud, sampling, entry,
                                                    entry.sink(pointcloud.device), *entry.params)
        twins, entry.last_twins = entry.last_twins, None
        for o, t in zip(outs, twins):
            _ext.attach_twin(o, t)
        L = len(sampling.inds)
        for i in range(L):
            if i < 2:
                end_points["sa%d_inds" % (i + 1)] = sampling.inds[i]
            end_points["sa%d_xyz" % (i + 1)] = sampling.xyz[i]
            end_points["sa%d_features" % (i + 1)] = outs[i]
        features = outs[-1]
        end_points["fp2_features"] = features
        end_points["fp2_xyz"] = end_points["sa2_xyz"]
        num_seed = end_points["fp2_xyz"].shape[1]
        dense = getattr(sampling, "fp2_inds", None)   # (prefetch_sampling)
        end_points["fp2_inds"] = dense if dense is not None and dense.shape[1] == num_seed \
            else end_points["sa1_inds"][:, 0:num_seed]
        return self._center_head(end_points, features, center_xyz, center_cls)

    def _center_head(self, end_points, features, center_xyz, center_cls):
        if center_xyz is not None:
            center_features = self.ctjt_head(end_points["sa2_xyz"], features,
                                             center_xyz.contiguous())
            onehot = torch.nn.functional.one_hot(center_cls, self.num_class)
            end_points["center_features"] = torch.cat(
                [center_features, onehot.transpose(1, 2).to(center_features.dtype)], dim=1)
        return end_points
