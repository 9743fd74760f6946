"""Average-precision evaluation with the reference's interface (models/ap_helper.py):
`parse_predictions`, `parse_groundtruths`, `APCalculator` -- SURVEY 8(f) #4.

Same inputs, same outputs (lists of `(class, corners (8,3) float64, score)` per scene,
`end_points['pred_mask']`, the metrics dict), but the per-box host loops of the reference
(B x K `.cpu()` round trips to decode a box, a Delaunay triangulation per box for
`remove_empty_box`, numpy NMS per scene, polygon clipping per (prediction, ground truth) pair
in a process pool) run as batched float64 tensor code plus three HIP kernels
(csrc/eval_boxes.hip: NMS, points-in-box count, oriented-box IoU).  Like the rest of the
package there is no CPU path: the inputs must live on the GPU.
"""
import numpy as np
import torch

from ..pointnet2 import _ext


def flip_axis_to_camera(pc):
    """depth (x, y, z) -> upright camera (x, -z, y); tensor or array (ap_helper.py:32-40)."""
    if isinstance(pc, torch.Tensor):
        return torch.stack((pc[..., 0], -pc[..., 2], pc[..., 1]), dim=-1)
    pc = np.asarray(pc)
    return np.stack((pc[..., 0], -pc[..., 2], pc[..., 1]), axis=-1)


def flip_axis_to_depth(pc):
    """upright camera (x, y, z) -> depth (x, z, -y) (ap_helper.py:50-54)."""
    if isinstance(pc, torch.Tensor):
        return torch.stack((pc[..., 0], pc[..., 2], -pc[..., 1]), dim=-1)
    pc = np.asarray(pc)
    return np.stack((pc[..., 0], pc[..., 2], -pc[..., 1]), axis=-1)


def get_3d_box_batch(box_size, heading_angle, center):
    """(…,3) sizes (l, w, h), (…) angles, (…,3) centres -> (…,8,3) corners, corner order and
    rotation of get_3d_box (utils/box_util.py:183-227); float64 tensors."""
    l, w, h = box_size[..., 0:1] / 2, box_size[..., 1:2] / 2, box_size[..., 2:3] / 2
    x = torch.cat((l, l, -l, -l, l, l, -l, -l), -1)
    y = torch.cat((h, h, h, h, -h, -h, -h, -h), -1)
    z = torch.cat((w, -w, -w, w, w, -w, -w, w), -1)
    c, s = torch.cos(heading_angle).unsqueeze(-1), torch.sin(heading_angle).unsqueeze(-1)
    # R = roty(angle) applied to (x, y, z): the products of the matrix row with the column,
    # zeros included in the reference's dot product do not change the sums
    cx = c * x + s * z
    cz = -s * x + c * z
    corners = torch.stack((cx, y, cz), dim=-1)
    return corners + center.unsqueeze(-2)


class SceneDetections(list):
    """One scene's detections in the reference's format -- a list of (class, corners (8,3)
    float64, score) -- that also keeps the arrays it was built from: `boxes` (n,8,3), and per
    list entry, in list order, `cls`, `score`, `box` (index into boxes).  eval_det reads the
    arrays instead of walking the tuples (45 000 per batch of 8 scenes with
    per_class_proposal: 55 of the 60 ms an evaluation batch took, tools/host_profile_eval.py).
    Any in-place change of the list drops the arrays; the tuples are then the truth again."""
    __slots__ = ('compact',)

    def __init__(self, items=(), compact=None):
        super().__init__(items)
        self.compact = compact if compact is not None and len(compact[1]) == len(self) else None


def _drops_compact(name):
    base = getattr(list, name)

    def method(self, *args, **kwargs):
        self.compact = None
        return base(self, *args, **kwargs)
    method.__name__ = name
    return method


for _name in ('__setitem__', '__delitem__', '__iadd__', '__imul__', 'append', 'extend', 'insert',
              'pop', 'remove', 'clear', 'sort', 'reverse'):
    setattr(SceneDetections, _name, _drops_compact(_name))


def _decode(end_points, dc, prefix=""):
    """Argmax class + gathered residual of the heading / size heads -> angle (B,K), size
    (B,K,3), centre in camera coordinates (B,K,3), all float64 (ap_helper.py:80-110)."""
    center = end_points[prefix + 'center']
    hcls = torch.argmax(end_points[prefix + 'heading_scores'], -1)
    hres = torch.gather(end_points[prefix + 'heading_residuals'], 2,
                        hcls.unsqueeze(-1)).squeeze(2)
    scls = torch.argmax(end_points[prefix + 'size_scores'], -1)
    sres = torch.gather(end_points[prefix + 'size_residuals'], 2,
                        scls.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, 1, 3)).squeeze(2)
    angle = dc.class2angle_batch(hcls, hres)
    size = dc.class2size_batch(scls, sres)
    return angle.contiguous(), size.contiguous(), flip_axis_to_camera(center.double()).contiguous()


def parse_predictions(end_points, config_dict, prefix=""):
    """Decode the head outputs to oriented boxes and suppress overlapping ones.

    `prefix`: which prediction head of a GroupFree3D model ('last_', 'proposal_', '0head_',
    ...; detection/GroupFree3D/models/ap_helper.py:69); its heads emit ONE objectness logit
    (sigmoid) where VoteNet's emits two (softmax) -- told apart by the last dimension.

    end_points: {center, heading_scores, heading_residuals, size_scores, size_residuals,
    sem_cls_scores, objectness_scores[, point_clouds]} on the GPU; config_dict:
    {dataset_config, remove_empty_box, use_3d_nms, nms_iou, use_old_type_nms, cls_nms,
    conf_thresh, per_class_proposal} (train_Votenet_FSB.py:204-207).

    Returns batch_pred_map_cls: per scene a list of (class, corners (8,3) float64 array in
    upright-camera coordinates, score); also stored in end_points together with 'pred_mask'
    (B,K) -- ap_helper.py:63-199."""
    dc = config_dict['dataset_config']
    center = end_points[prefix + 'center']
    if not center.is_cuda:
        raise RuntimeError("CPU not supported")
    with torch.no_grad():
        angle, size, center_cam = _decode(end_points, dc, prefix)
        corners = get_3d_box_batch(size, angle, center_cam)           # (B,K,8,3) f64
        sem = end_points[prefix + 'sem_cls_scores'].detach().float()
        pred_sem_cls = torch.argmax(sem, -1)
        sem_probs = torch.softmax(sem, -1)
        obj = end_points[prefix + 'objectness_scores'].detach().float()
        obj_prob = torch.sigmoid(obj)[..., 0] if obj.shape[-1] == 1 else \
            torch.softmax(obj, -1)[..., 1]
        B, K = obj_prob.shape

        valid = None
        if config_dict['remove_empty_box']:
            pts = end_points['point_clouds'].detach().float().contiguous()
            count = _ext.points_in_boxes(pts, center_cam, size, angle, 5)
            valid = (count >= 5).to(torch.uint8).contiguous()

        lo, hi = corners.amin(dim=2), corners.amax(dim=2)
        if not config_dict['use_3d_nms']:
            boxes = torch.stack((lo[..., 0], lo[..., 2], hi[..., 0], hi[..., 2]), -1)
            cls = None
        else:
            boxes = torch.cat((lo, hi), -1)
            cls = pred_sem_cls.int().contiguous() if config_dict.get('cls_nms', False) else None
        pick = _ext.nms_boxes(boxes.contiguous(), obj_prob.double().contiguous(),
                              config_dict['nms_iou'], config_dict['use_old_type_nms'],
                              cls=cls, valid=valid)
        keep = pick.bool() & (obj_prob > config_dict['conf_thresh'])

        pred_mask = pick.cpu().numpy().astype(np.float64)
        keep_np = keep.cpu().numpy()
        corners_np = corners.cpu().numpy()
        obj_np = obj_prob.cpu().numpy()
        if config_dict['per_class_proposal']:
            score_np = (sem_probs * obj_prob.unsqueeze(-1)).cpu().numpy()
        else:
            cls_np = pred_sem_cls.cpu().numpy()

    end_points['pred_mask'] = pred_mask
    batch_pred_map_cls = []
    for i in range(B):
        js = np.nonzero(keep_np[i])[0]
        boxes = corners_np[i][js]                    # (n,8,3), one copy; the tuples hold views
        views = list(boxes)
        n = len(js)
        if config_dict['per_class_proposal']:
            sc = np.ascontiguousarray(score_np[i][js].T)         # (classes, n)
            cur = [(ii, b, s) for ii in range(dc.num_class) for b, s in zip(views, sc[ii])]
            compact = (boxes, np.repeat(np.arange(dc.num_class, dtype=np.int64), n),
                       sc.reshape(-1).astype(np.float64),
                       np.tile(np.arange(n, dtype=np.int64), dc.num_class))
        else:
            cl, sc = cls_np[i][js], obj_np[i][js]
            cur = [(c, b, s) for c, b, s in zip(cl.tolist(), views, sc)]
            compact = (boxes, cl.astype(np.int64), sc.astype(np.float64),
                       np.arange(n, dtype=np.int64))
        batch_pred_map_cls.append(SceneDetections(cur, compact))
    end_points['batch_pred_map_cls'] = batch_pred_map_cls
    return batch_pred_map_cls


def parse_groundtruths(end_points, config_dict):
    """Ground-truth labels -> per scene a list of (class, corners (8,3) float64)
    (ap_helper.py:202-246)."""
    dc = config_dict['dataset_config']
    with torch.no_grad():
        center_cam = flip_axis_to_camera(end_points['center_label'][:, :, 0:3].double())
        angle = dc.class2angle_batch(end_points['heading_class_label'],
                                     end_points['heading_residual_label'])
        size = dc.class2size_batch(end_points['size_class_label'],
                                   end_points['size_residual_label'])
        corners = get_3d_box_batch(size, angle, center_cam).cpu().numpy()
        mask = end_points['box_label_mask'].cpu().numpy()
        sem = end_points['sem_cls_label'].cpu().numpy()
    batch_gt_map_cls = []
    for i in range(corners.shape[0]):
        batch_gt_map_cls.append([(int(sem[i, j]), corners[i, j])
                                 for j in range(corners.shape[1]) if mask[i, j] == 1])
    end_points['batch_gt_map_cls'] = batch_gt_map_cls
    return batch_gt_map_cls


# --------------------------------------------------------------------------------------- AP
def voc_ap(rec, prec):
    """Area under the monotone precision envelope (utils/eval_det.py:19-52, VOC 2010+)."""
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def _box_key(box):
    a = np.asarray(box)
    return (a.__array_interface__['data'][0], a.shape, a.strides)


def _scene_arrays(dets, code_of):
    """One scene's detection list -> (boxes (n,8,3), class code, score, box index per entry).
    A SceneDetections that still carries its arrays is read without touching the tuples;
    any other list is walked (the same box listed once per class is stored once)."""
    compact = getattr(dets, 'compact', None) if type(dets) is SceneDetections else None
    if compact is not None and len(compact[1]) == len(dets):
        boxes, cls, score, box = compact
        vals, first, inv = np.unique(cls, return_index=True, return_inverse=True)
        codes = np.empty(vals.size, dtype=np.int64)
        for k in np.argsort(first, kind="stable"):         # first appearance, like the walk
            codes[k] = code_of.setdefault(int(vals[k]), len(code_of))
        return boxes, codes[inv.reshape(-1)], score, box
    uniq, boxes, cls, score, box = {}, [], [], [], []
    for c, b, sc in dets:
        k = _box_key(b)
        at = uniq.get(k)
        if at is None:
            at = uniq[k] = len(boxes)
            boxes.append(np.asarray(b, dtype=np.float64))
        cls.append(code_of.setdefault(c, len(code_of)))
        score.append(sc)
        box.append(at)
    return (np.stack(boxes) if boxes else np.zeros((0, 8, 3)), np.array(cls, dtype=np.int64),
            np.array(score, dtype=np.float64), np.array(box, dtype=np.int64))


def eval_det(pred_all, gt_all, ovthresh=0.25, device=None):
    """utils/eval_det.py:211-256 (eval_det_multiprocessing with get_iou_obb): pred_all
    {scene: [(class, corners, score)]}, gt_all {scene: [(class, corners)]} -> rec, prec, ap
    keyed by class.  One IoU launch for all (prediction, ground truth) pairs of all scenes; the
    greedy matching (a ground-truth box is credited to the highest-scoring detection that
    picks it) is evaluated without a per-detection loop: a detection's best ground-truth box
    does not depend on the matching state, so the first detection per (scene, box) in score
    order is the true positive.  Classes are any hashable; inside they are integer codes in
    order of first appearance (predictions, then ground truths).
    `ovthresh` may be a sequence of thresholds: the IoUs and every detection's best ground-truth
    box do not depend on it, so they are computed once and the result is {threshold: (rec, prec,
    ap)} (GroupFree3D evaluates every head at 0.25 and 0.5, train_GF_FSB.py:384-445)."""
    many = isinstance(ovthresh, (tuple, list))
    thresholds = list(ovthresh) if many else [ovthresh]
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    scenes = sorted(set(pred_all.keys()) | set(gt_all.keys()), key=lambda s: (str(type(s)), s))
    S = len(scenes)
    code_of = {}
    per_scene = [_scene_arrays(pred_all.get(s, ()), code_of) for s in scenes]
    pred_codes = len(code_of)
    gb, gcode = [], []
    for s in scenes:
        g = gt_all.get(s, [])
        gb.append([np.asarray(b, dtype=np.float64) for _, b in g])
        gcode.append([code_of.setdefault(c, len(code_of)) for c, _ in g])
    classes = list(code_of)                      # dicts keep insertion order
    P = max([len(a[0]) for a in per_scene] + [1])
    G = max([len(b) for b in gb] + [1])
    c1 = np.zeros((S, P, 8, 3))
    c2 = np.zeros((S, G, 8, 3))
    gmat = np.full((S, G), -1, dtype=np.int64)   # class code of every ground-truth box
    for si in range(S):
        if len(per_scene[si][0]):
            c1[si, :len(per_scene[si][0])] = per_scene[si][0]
        if gb[si]:
            c2[si, :len(gb[si])] = np.stack(gb[si])
            gmat[si, :len(gb[si])] = gcode[si]
    with np.errstate(all="ignore"):
        iou = _ext.box3d_iou(torch.from_numpy(c1).to(device),
                             torch.from_numpy(c2).to(device)).cpu().numpy() if S else None

    cat = lambda k, dt: np.concatenate([a[k] for a in per_scene]).astype(dt, copy=False) \
        if S else np.zeros(0, dtype=dt)
    pcode, pscore, pbox = cat(1, np.int64), cat(2, np.float64), cat(3, np.int64)
    pscene = np.repeat(np.arange(S, dtype=np.int64), [len(a[1]) for a in per_scene])
    by_code = np.argsort(pcode, kind="stable")   # entries of one class, in list order
    start = np.searchsorted(pcode[by_code], np.arange(len(classes) + 1))
    out = {thr: ({}, {}, {}) for thr in thresholds}
    for code, c in enumerate(classes):
        sel = by_code[start[code]:start[code + 1]]
        if sel.size == 0:
            if code >= pred_codes:               # only the ground truths have it
                for rec, prec, ap in out.values():
                    rec[c], prec[c], ap[c] = 0, 0, 0
            continue
        order = sel[np.argsort(-pscore[sel], kind="stable")]
        nd = order.size
        sc = pscene[order]
        ovmax = np.empty(nd)
        jmax = np.empty(nd, dtype=np.int64)
        for lo in range(0, nd, 1 << 18):         # (bounded temporaries: rows x G)
            rows = slice(lo, min(nd, lo + (1 << 18)))
            sub = np.where(gmat[sc[rows]] == code, iou[sc[rows], pbox[order[rows]]], -np.inf)
            best = np.argmax(sub, axis=1)
            ovmax[rows] = sub[np.arange(best.size), best]
            jmax[rows] = best
        npos = int(np.sum(gmat == code))
        for thr, (rec, prec, ap) in out.items():
            hit = ovmax > thr
            tp = np.zeros(nd)
            keys = sc[hit] * (G + 1) + jmax[hit]
            _, first = np.unique(keys, return_index=True)
            tp[np.nonzero(hit)[0][first]] = 1.0
            fp = 1.0 - tp
            fp = np.cumsum(fp)
            tp = np.cumsum(tp)
            with np.errstate(all="ignore"):
                rec[c] = tp / float(npos)
                prec[c] = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
                ap[c] = voc_ap(rec[c], prec[c])
    return out if many else out[ovthresh]


class APCalculator(object):
    """Accumulates predictions / ground truths over batches and computes per-class AP, mAP,
    recall and AR (ap_helper.py:249-301)."""

    def __init__(self, ap_iou_thresh=0.25, class2type_map=None):
        self.ap_iou_thresh = ap_iou_thresh
        self.class2type_map = class2type_map
        self.reset()

    def step(self, batch_pred_map_cls, batch_gt_map_cls):
        bsize = len(batch_pred_map_cls)
        assert bsize == len(batch_gt_map_cls)
        for i in range(bsize):
            self.gt_map_cls[self.scan_cnt] = batch_gt_map_cls[i]
            self.pred_map_cls[self.scan_cnt] = batch_pred_map_cls[i]
            self.scan_cnt += 1

    def compute_metrics(self, ap_iou_thresh=None):
        """The reference's metrics dict at this calculator's threshold; with a sequence of
        thresholds, {threshold: metrics dict} from one pass over the boxes (see eval_det)."""
        if isinstance(ap_iou_thresh, (tuple, list)):
            res = eval_det(self.pred_map_cls, self.gt_map_cls, ovthresh=list(ap_iou_thresh))
            return {thr: self._metrics(*res[thr]) for thr in ap_iou_thresh}
        thr = self.ap_iou_thresh if ap_iou_thresh is None else ap_iou_thresh
        return self._metrics(*eval_det(self.pred_map_cls, self.gt_map_cls, ovthresh=thr))

    def _metrics(self, rec, prec, ap):
        ret_dict = {}
        for key in sorted(ap.keys()):
            clsname = self.class2type_map[key] if self.class2type_map else str(key)
            ret_dict['%s Average Precision' % (clsname)] = ap[key]
        ret_dict['mAP'] = np.mean(list(ap.values()))
        rec_list = []
        for key in sorted(ap.keys()):
            clsname = self.class2type_map[key] if self.class2type_map else str(key)
            try:
                ret_dict['%s Recall' % (clsname)] = rec[key][-1]
                rec_list.append(rec[key][-1])
            except (TypeError, IndexError):
                ret_dict['%s Recall' % (clsname)] = 0
                rec_list.append(0)
        ret_dict['AR'] = np.mean(rec_list)
        return ret_dict

    def reset(self):
        self.gt_map_cls = {}
        self.pred_map_cls = {}
        self.scan_cnt = 0
