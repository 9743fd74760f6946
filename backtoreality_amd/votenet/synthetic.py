"""Deterministic synthetic scenes with the batch-dict schema of the reference's datasets
(detection/Votenet/scannet/scannet_detection_dataset.py:197-219).

There is no ScanNet / Matterport data offline, so benchmarks and parity tests run on generated
rooms: scene `i` is drawn from numpy.random.default_rng(1000 + i).
  * surface-room (default): a box-shaped room U(4,8) x U(4,8) x U(2.4,3) m with its origin at
    a corner; 35 % of the points on the floor, 35 % on the four walls, 30 % on the faces of
    8-16 random boxes (0.3-1.5 m) standing on the floor; Gaussian jitter sigma = 5 mm;
    shuffled.  This mimics the surface density of an indoor scan (the ball-query hit counts
    depend on it).
  * uniform-volume: points uniform in the same extents (sparse / stress variant).
The height feature is z - percentile(z, 0.99) as in the loader (:122-125).
"""
import numpy as np
import torch

from .config import DatasetConfig


def _points_on_box_faces(rng, n, center, size):
    """n points uniform (by area) on the 5 visible faces of an axis-aligned box."""
    sx, sy, sz = size
    areas = np.array([sx * sy, sx * sz, sx * sz, sy * sz, sy * sz])  # top, +-y, +-x
    face = rng.choice(5, size=n, p=areas / areas.sum())
    u = rng.uniform(-0.5, 0.5, size=(n, 3)) * size
    u[face == 0, 2] = 0.5 * sz
    u[face == 1, 1] = 0.5 * sy
    u[face == 2, 1] = -0.5 * sy
    u[face == 3, 0] = 0.5 * sx
    u[face == 4, 0] = -0.5 * sx
    return u + center


def make_scene(index, num_points=40000, config=None, use_height=True, kind="surface",
               extent_scale=1.0, center_jitter=0.0):
    """One scene as a dict of numpy arrays (same keys/dtypes as the reference loader).
    center_jitter > 0 (CenterRefine recipe, scannet_detection_dataset.py:78-86,190-200): the
    GT centres are displaced by size * U(-jitter/2, jitter/2) and the displacement is returned
    as 'center_jitter' (drawn from its own generator: every other field is unchanged)."""
    config = config or DatasetConfig(22, 1, 22)
    rng = np.random.default_rng(1000 + int(index))
    ext = np.array([rng.uniform(4, 8), rng.uniform(4, 8), rng.uniform(2.4, 3.0)])
    ext[:2] *= extent_scale
    nbox = int(rng.integers(8, 17))
    sizes = rng.uniform(0.3, 1.5, size=(nbox, 3))
    centers = np.empty((nbox, 3))
    centers[:, 0] = rng.uniform(0.5 * sizes[:, 0], ext[0] - 0.5 * sizes[:, 0])
    centers[:, 1] = rng.uniform(0.5 * sizes[:, 1], ext[1] - 0.5 * sizes[:, 1])
    centers[:, 2] = 0.5 * sizes[:, 2]

    N = int(num_points)
    votes = np.zeros((N, 3))
    vote_mask = np.zeros((N,), np.int64)
    instance = np.full((N,), -1, np.int64)   # which box a point belongs to (GroupFree3D labels)
    if kind == "surface":
        n_floor = int(0.35 * N)
        n_wall = int(0.35 * N)
        n_obj = N - n_floor - n_wall
        floor = np.stack([rng.uniform(0, ext[0], n_floor), rng.uniform(0, ext[1], n_floor),
                          np.zeros(n_floor)], 1)
        w = rng.integers(0, 4, n_wall)
        t = rng.uniform(0, 1, n_wall)
        wall = np.empty((n_wall, 3))
        wall[:, 2] = rng.uniform(0, ext[2], n_wall)
        wall[:, 0] = np.where(w == 0, 0.0, np.where(w == 1, ext[0], t * ext[0]))
        wall[:, 1] = np.where(w == 2, 0.0, np.where(w == 3, ext[1], t * ext[1]))
        per = rng.multinomial(n_obj, np.full(nbox, 1.0 / nbox))
        obj, obj_vote = [], []
        for b in range(nbox):
            p = _points_on_box_faces(rng, per[b], centers[b], sizes[b])
            obj.append(p)
            obj_vote.append(centers[b] - p)
        obj = np.concatenate(obj, 0)
        pts = np.concatenate([floor, wall, obj], 0)
        votes[n_floor + n_wall:] = np.concatenate(obj_vote, 0)
        vote_mask[n_floor + n_wall:] = 1
        instance[n_floor + n_wall:] = np.repeat(np.arange(nbox), per)
        pts = pts + rng.normal(0.0, 0.005, size=pts.shape)
    elif kind == "uniform":
        pts = rng.uniform(0, 1, size=(N, 3)) * ext
        inside = np.zeros((N,), bool)
        for b in range(nbox):
            m = np.all(np.abs(pts - centers[b]) <= 0.5 * sizes[b], axis=1) & ~inside
            votes[m] = centers[b] - pts[m]
            instance[m] = b
            inside |= m
        vote_mask[inside] = 1
    else:
        raise ValueError(kind)
    perm = rng.permutation(N)
    pts, votes, vote_mask, instance = pts[perm], votes[perm], vote_mask[perm], instance[perm]

    pc = pts.astype(np.float32)
    if use_height:
        floor_h = np.percentile(pc[:, 2], 0.99)
        pc = np.concatenate([pc, (pc[:, 2] - floor_h)[:, None]], 1).astype(np.float32)

    K = config.max_num_obj
    size_cls = rng.integers(0, config.num_size_cluster, nbox)
    ret = {
        'point_clouds': pc,
        'center_label': np.zeros((K, 3), np.float32),
        'heading_class_label': np.zeros((K,), np.int64),
        'heading_residual_label': np.zeros((K,), np.float32),
        'size_class_label': np.zeros((K,), np.int64),
        'size_residual_label': np.zeros((K, 3), np.float32),
        'sem_cls_label': np.zeros((K,), np.int64),
        'box_label_mask': np.zeros((K,), np.float32),
        'vote_label': np.tile(votes, (1, 3)).astype(np.float32),  # 3 identical GT votes
        'vote_label_mask': vote_mask,
        # GroupFree3D's extra labels (GroupFree3D/scannet/scannet_detection_dataset.py:181,
        # 220-258)
        'size_gts': np.zeros((K, 3), np.float32),
        'point_obj_mask': (instance >= 0).astype(np.int64),
        'point_instance_label': instance,
    }
    ret['size_gts'][:nbox] = sizes
    ret['center_label'][:nbox] = centers
    ret['size_class_label'][:nbox] = size_cls
    ret['size_residual_label'][:nbox] = sizes - config.mean_size_arr[size_cls]
    ret['sem_cls_label'][:nbox] = size_cls % config.num_class
    ret['box_label_mask'][:nbox] = 1.0
    if config.num_heading_bin > 1:
        ret['heading_class_label'][:nbox] = rng.integers(0, config.num_heading_bin, nbox)
        ret['heading_residual_label'][:nbox] = rng.uniform(
            -0.5, 0.5, nbox) * (2 * np.pi / config.num_heading_bin)
    if center_jitter:
        delta = (np.random.default_rng(5000 + int(index)).random((K, 3)) - 0.5) * center_jitter
        size_gts = np.zeros((K, 3))
        size_gts[:nbox] = sizes
        ret['center_jitter'] = (size_gts * delta).astype(np.float32)
        ret['center_label'] = (ret['center_label'] + ret['center_jitter']).astype(np.float32)
    return ret


def make_batch(first_index, batch_size, num_points=40000, config=None, use_height=True,
               kind="surface", extent_scale=1.0, device=None, center_jitter=0.0):
    """Stack `batch_size` consecutive scenes into a dict of torch tensors on `device`."""
    scenes = [make_scene(first_index + i, num_points, config, use_height, kind, extent_scale,
                         center_jitter) for i in range(batch_size)]
    batch = {k: torch.from_numpy(np.stack([s[k] for s in scenes], 0)) for k in scenes[0]}
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch


def make_eval_case(seed, batch_size, num_points, config, num_proposal=256, device=None):
    """A labelled batch plus synthetic HEAD OUTPUTS for the evaluation path (ap_helper): about
    70 % of the proposals sit near a ground-truth box (centre / size / heading / class perturbed),
    the rest are random boxes in the room with low objectness, so that NMS, the confidence
    threshold and both true and false positives all occur.  Every class of the config owns at
    least one ground-truth box when the batch holds enough boxes.  Returns a dict with the
    batch labels and 'center', 'heading_scores', 'heading_residuals', 'size_scores',
    'size_residuals', 'sem_cls_scores', 'objectness_scores' (CPU tensors unless `device`)."""
    batch = make_batch(seed, batch_size, num_points, config)
    rng = np.random.default_rng(9000 + int(seed))
    B, K = batch_size, num_proposal
    NH, NS, NC = config.num_heading_bin, config.num_size_cluster, config.num_class
    mask = batch['box_label_mask'].numpy()
    sem = batch['sem_cls_label'].numpy().copy()
    t = 0
    for i in range(B):
        for j in range(mask.shape[1]):
            if mask[i, j] == 1:
                sem[i, j] = t % NC
                t += 1
    batch['sem_cls_label'] = torch.from_numpy(sem)
    gc = batch['center_label'].numpy()
    ghc, ghr = batch['heading_class_label'].numpy(), batch['heading_residual_label'].numpy()
    gsc, gsr = batch['size_class_label'].numpy(), batch['size_residual_label'].numpy()
    pc = batch['point_clouds'].numpy()

    center = np.zeros((B, K, 3), np.float32)
    hs = rng.normal(0, 1, (B, K, NH)).astype(np.float32)
    hr = rng.normal(0, 0.05, (B, K, NH)).astype(np.float32)
    ss = rng.normal(0, 1, (B, K, NS)).astype(np.float32)
    sr = rng.normal(0, 0.05, (B, K, NS, 3)).astype(np.float32)
    sc = rng.normal(0, 1, (B, K, NC)).astype(np.float32)
    ob = np.zeros((B, K, 2), np.float32)
    for i in range(B):
        valid = np.nonzero(mask[i] == 1)[0]
        lo, hi = pc[i, :, :3].min(0), pc[i, :, :3].max(0)
        for k in range(K):
            if valid.size and rng.random() < 0.7:
                g = valid[rng.integers(0, valid.size)]
                center[i, k] = gc[i, g] + rng.normal(0, 0.08, 3)
                hs[i, k, ghc[i, g]] += 4.0
                hr[i, k, ghc[i, g]] += ghr[i, g]
                ss[i, k, gsc[i, g]] += 4.0
                sr[i, k, gsc[i, g]] += gsr[i, g]
                sc[i, k, sem[i, g] if rng.random() < 0.85 else rng.integers(0, NC)] += 3.0
                ob[i, k] = (-1.0, 1.0) + rng.normal(0, 0.8, 2)
            else:
                center[i, k] = rng.uniform(lo, hi)
                ob[i, k] = (1.5, -1.5) + rng.normal(0, 1.0, 2)
    out = dict(batch)
    out.update({
        'center': torch.from_numpy(center), 'heading_scores': torch.from_numpy(hs),
        'heading_residuals': torch.from_numpy(hr), 'size_scores': torch.from_numpy(ss),
        'size_residuals': torch.from_numpy(sr), 'sem_cls_scores': torch.from_numpy(sc),
        'objectness_scores': torch.from_numpy(ob.astype(np.float32)),
    })
    if device is not None:
        out = {k: v.to(device) for k, v in out.items()}
    return out
