"""GPU parity: every op of the C ABI (through `pointnet2._ext`) against the CPU oracle.
Indices must be bit-exact; float outputs of pure copies bit-exact; scatter-adds (atomics,
unordered) within 1e-4 relative."""
import numpy as np
import pytest
import torch

import oracle
from backtoreality_amd.votenet import synthetic

# every test of this module runs once per rounding mode of the squared distance (conftest.py)
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("distance_mode")]


def _ext():
    from backtoreality_amd.pointnet2 import _ext
    return _ext


def _scene_xyz(B, N, first=0, kind="surface"):
    return np.stack([synthetic.make_scene(first + i, N, use_height=False, kind=kind)['point_clouds']
                     for i in range(B)], 0)


def _t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


# ----------------------------------------------- inputs on which the rounding modes disagree
@pytest.mark.parametrize("N,M", [(4096, 256), (20000, 128)])   # register / bucketed FPS kernel
def test_fps_mode_sensitive_cloud(cuda, N, M):
    from mode_cases import sphere_cloud
    xyz = sphere_cloud(3, N)
    ref = oracle.furthest_point_sampling(xyz, M)
    got = _ext().furthest_point_sampling(_t(xyz, cuda), M).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_ball_query_and_three_nn_mode_sensitive(cuda, distance_mode):
    from mode_cases import shell_ball_case
    assert _ext()._idx.btr_distance_mode() == distance_mode == oracle.fmad()
    for n in (4096, 12000):        # wave-per-centre kernel / grid kernel
        centres, pts, radius = shell_ball_case(4, n=n)
        ref = oracle.ball_query(centres, pts, radius, 64)
        got = _ext().ball_query(_t(centres, cuda), _t(pts, cuda), float(radius), 64).cpu().numpy()
        np.testing.assert_array_equal(got, ref)
    d_ref, i_ref = oracle.three_nn(centres, pts[:, :500])
    d_got, i_got = _ext().three_nn(_t(centres, cuda), _t(pts[:, :500].copy(), cuda))
    np.testing.assert_array_equal(i_got.cpu().numpy(), i_ref)
    np.testing.assert_array_equal(d_got.cpu().numpy(), d_ref)


# ----------------------------------------------------------------------------------- FPS
@pytest.mark.parametrize("B,N,M", [(2, 64, 16), (3, 200, 50), (2, 1000, 256), (2, 2048, 1024),
                                   (2, 4096, 2048), (1, 7000, 300), (2, 20000, 512)])
def test_fps_matches_oracle(cuda, B, N, M):
    xyz = _scene_xyz(B, N)
    ref = oracle.furthest_point_sampling(xyz, M)
    got = _ext().furthest_point_sampling(_t(xyz, cuda), M).cpu().numpy()
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("bs", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512])
@pytest.mark.parametrize("N", [1500, 9000])
def test_fps_tie_break_every_block_size(cuda, bs, N):
    """Duplicated points force exact ties; the winner must follow the reference's tree for
    every template instantiation of its kernel (sampling_gpu.cu:121-174)."""
    rng = np.random.default_rng(5)
    base = rng.uniform(0.2, 3.0, size=(2, N // 3, 3)).astype(np.float32)
    xyz = np.concatenate([base, base, base], 1)
    ref = oracle.furthest_point_sampling(xyz, 200, block_size=bs)
    got = _ext().furthest_point_sampling_bs(_t(xyz, cuda), 200, bs).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.fixture
def fps_lds(cuda):
    """Sets the dynamic LDS of the large-scene FPS launch for one test (btr_fps_set_lds_kb) and
    restores the default rules afterwards."""
    ext = _ext()
    yield ext.set_fps_lds_kb
    ext.set_fps_lds_kb(-1)


def test_fps_full_size_and_both_large_kernels(cuda, monkeypatch, fps_lds):
    """BASELINE configs[1] size (B=8, N=40000, M=2048): the bucketed kernel and the streaming
    kernel (BTR_FPS_IMPL=stream) both reproduce the oracle bit for bit -- the bucketed kernel
    with its running min-dists wholly in LDS (the default: 157 KB for 40 000 points), split
    between LDS and the global workspace (128 KB: 32 768 points' worth; 20 KB) and wholly in
    global memory (0): where a min-dist lives can never change an index."""
    xyz = _scene_xyz(8, 40000)
    ref = oracle.furthest_point_sampling(xyz, 2048)
    x = _t(xyz, cuda)
    got = _ext().furthest_point_sampling(x, 2048).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    assert _ext()._idx.btr_fps_lds_kb(40000) * 1024 >= 40000 * 4
    for kb in (128, 20, 0):
        fps_lds(kb)
        got = _ext().furthest_point_sampling(x, 2048).cpu().numpy()
        np.testing.assert_array_equal(got, ref)
    fps_lds(-1)
    monkeypatch.setenv("BTR_FPS_IMPL", "stream")
    got = _ext().furthest_point_sampling(x[:2], 600).cpu().numpy()
    np.testing.assert_array_equal(got, ref[:2, :600])


@pytest.mark.parametrize("lds_kb", [-1, 24, 0])
def test_fps_bucketed_ties_and_skips_wherever_the_min_dists_live(cuda, fps_lds, lds_kb):
    """Skipped points, duplicated points (exact ties) at several block sizes and a cloud of one
    repeated point, with the min-dists in LDS (default), partly there (24 KB: 6 144 points) and
    in global memory."""
    fps_lds(lds_kb)
    rng = np.random.default_rng(12)
    xyz = rng.uniform(-3, 3, size=(2, 12000, 3)).astype(np.float32)
    xyz[:, 500:900] *= 0.004
    xyz[1, 3000:6000] = xyz[1, 2999]
    ref = oracle.furthest_point_sampling(xyz, 900)
    got = _ext().furthest_point_sampling(_t(xyz, cuda), 900).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    for bs in (1, 8, 512):
        base = rng.uniform(0.2, 3.0, size=(1, 3000, 3)).astype(np.float32)
        dup = np.concatenate([base, base, base], 1)
        ref = oracle.furthest_point_sampling(dup, 300, block_size=bs)
        got = _ext().furthest_point_sampling_bs(_t(dup, cuda), 300, bs).cpu().numpy()
        np.testing.assert_array_equal(got, ref)
    same = np.tile(np.array([[1.5, -2.0, 0.25]], np.float32), (1, 9000, 1))
    got = _ext().furthest_point_sampling(_t(same, cuda), 50).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.furthest_point_sampling(same, 50))


@pytest.mark.parametrize("kind,N", [("uniform", 30000), ("surface", 65536), ("surface", 4097),
                                    ("surface", 80000)])
def test_fps_bucketed_other_distributions(cuda, kind, N):
    xyz = _scene_xyz(2, N, first=20, kind=kind)
    ref = oracle.furthest_point_sampling(xyz, 700)
    got = _ext().furthest_point_sampling(_t(xyz, cuda), 700).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_fps_bucketed_skip_rule_and_all_duplicates(cuda):
    rng = np.random.default_rng(12)
    xyz = rng.uniform(-3, 3, size=(2, 12000, 3)).astype(np.float32)
    xyz[:, 500:900] *= 0.004          # a cluster inside the skipped ball around the origin
    xyz[1, 3000:6000] = xyz[1, 2999]  # 3001 identical points: long exact ties
    ref = oracle.furthest_point_sampling(xyz, 900)
    got = _ext().furthest_point_sampling(_t(xyz, cuda), 900).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    same = np.tile(np.array([[1.5, -2.0, 0.25]], np.float32), (1, 9000, 1))
    ref = oracle.furthest_point_sampling(same, 50)
    got = _ext().furthest_point_sampling(_t(same, cuda), 50).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_fps_origin_skip_and_degenerate(cuda):
    rng = np.random.default_rng(11)
    xyz = rng.uniform(-2, 2, size=(2, 3000, 3)).astype(np.float32)
    xyz[:, 100:400] *= 0.01          # inside the x^2+y^2+z^2 <= 1e-3 ball: never selected
    xyz[1, 0] = 0.0                  # index 0 is always the first sample, even if skipped
    ref = oracle.furthest_point_sampling(xyz, 500)
    got = _ext().furthest_point_sampling(_t(xyz, cuda), 500).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    allzero = np.zeros((1, 300, 3), np.float32)  # nothing competes -> every index is 0
    got = _ext().furthest_point_sampling(_t(allzero, cuda), 10).cpu().numpy()
    np.testing.assert_array_equal(got, np.zeros((1, 10), np.int32))
    assert _ext().furthest_point_sampling(_t(xyz, cuda), 0).shape == (2, 0)


def test_fps_prefix_invariant(cuda):
    """FPS over an FPS-ordered prefix returns arange (backbone_module.py:113-132 relies on it)."""
    xyz = _scene_xyz(2, 8192)
    x = _t(xyz, cuda)
    inds = _ext().furthest_point_sampling(x, 2048)
    sub = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    again = _ext().furthest_point_sampling(sub, 1024).cpu().numpy()
    np.testing.assert_array_equal(again, np.tile(np.arange(1024, dtype=np.int32), (2, 1)))


# ---- FPS of a cloud the caller EXPECTS to be FPS-ordered (btr_furthest_point_sampling_ordered):
# the hint selects a parallel check of "the answer is 0..m-1" with the serial kernel behind it;
# it must never change an index, whatever the cloud really is.
def _ordered(x, m, bs=None):
    e = _ext()
    e.mark_fps_ordered(x)
    return e.furthest_point_sampling(x, m) if bs is None else e.furthest_point_sampling_bs(x, m, bs)


@pytest.mark.parametrize("N1,M1,M2", [(40000, 2048, 1024), (9000, 1024, 512), (5000, 512, 256),
                                       (20000, 4096, 2048), (6000, 700, 333), (3000, 100, 100)])
def test_fps_ordered_on_a_true_prefix(cuda, monkeypatch, N1, M1, M2):
    """A real pyramid level: the hypothesis holds, the answer is the identity and it is the
    oracle's answer; the serial kernel (BTR_FPS_PREFIX=0) agrees."""
    xyz = _scene_xyz(3, N1, first=40)
    x = _t(xyz, cuda)
    inds = _ext().furthest_point_sampling(x, M1)
    sub = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    ref = oracle.furthest_point_sampling(sub.cpu().numpy(), M2)
    got = _ordered(sub, M2).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    if (N1, M1) != (20000, 4096):
        # (4096 -> 2048 of this cloud: the oracle itself leaves the identity at two positions --
        # an exact tie that the keys of n = 4096 order differently than those of n = 20000 did;
        # the check notices and the serial kernel answers for that scene)
        np.testing.assert_array_equal(got, np.tile(np.arange(M2, dtype=np.int32), (3, 1)))
    monkeypatch.setenv("BTR_FPS_PREFIX", "0")
    np.testing.assert_array_equal(_ordered(sub, M2).cpu().numpy(), ref)


@pytest.mark.parametrize("B,N,M", [(2, 2048, 1024), (3, 1000, 400), (2, 4096, 2048), (1, 300, 300),
                                   (2, 64, 2), (2, 257, 100)])
def test_fps_ordered_hint_on_an_unordered_cloud(cuda, B, N, M):
    """The hypothesis is false from the first steps on: the fallback answers, bit-exact."""
    xyz = _scene_xyz(B, N, first=7)
    ref = oracle.furthest_point_sampling(xyz, M)
    assert not np.array_equal(ref[0], np.arange(M))
    np.testing.assert_array_equal(_ordered(_t(xyz, cuda), M).cpu().numpy(), ref)


@pytest.mark.parametrize("bs", [1, 16, 256, 512])
def test_fps_ordered_when_ties_break_the_identity(cuda, bs):
    """Duplicated points: the first level picks every distinct point, then duplicates under ITS
    tie-break (n = 1200).  On the 1024-point prefix the ties resolve by the keys of n = 1024, the
    sequence is no longer 0, 1, 2, ... and the check must notice -- in scene 0 only; scene 1 is
    an ordinary pyramid level whose identity holds, so one launch answers both ways."""
    rng = np.random.default_rng(31)
    base = rng.uniform(0.3, 3.0, size=(1, 600, 3)).astype(np.float32)
    dup = np.concatenate([base, base], 1)                        # every point twice
    plain = _scene_xyz(1, 1200, first=3)
    lvl1 = oracle.furthest_point_sampling(np.concatenate([dup, plain], 0), 1024, block_size=bs)
    both = np.concatenate([dup, plain], 0)
    sub = np.take_along_axis(both, lvl1[:, :, None].astype(np.int64), 1)
    ref = oracle.furthest_point_sampling(sub, 800, block_size=bs)
    assert not np.array_equal(ref[0], np.arange(800)), "the crafted ties do not break the identity"
    assert np.array_equal(ref[1], np.arange(800))
    got = _ordered(_t(sub, cuda), 800, bs).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_fps_ordered_skip_rule_and_exact_tie_with_an_earlier_point(cuda):
    """(a) a sample inside the skipped ball around the origin among 1..m-1 refutes the hypothesis
    at once; (b) point j duplicating an earlier sample (R_j = 0) is decided by the tie-break
    keys; (c) a later point exactly as far as point j (T_j[k] == R_j, k > j)."""
    xyz = _scene_xyz(2, 6000, first=11)
    x = _t(xyz, cuda)
    inds = _ext().furthest_point_sampling(x, 512)
    sub = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    a = sub.clone(); a[0, 37] = torch.tensor([0.01, 0.0, 0.01], device=cuda)
    b = sub.clone(); b[1, 200] = b[1, 13]
    c = sub.clone(); c[0, 400] = c[0, 90]; c[1, 511] = c[1, 255]
    for t in (a, b, c):
        ref = oracle.furthest_point_sampling(t.cpu().numpy(), 256)
        np.testing.assert_array_equal(_ordered(t, 256).cpu().numpy(), ref)
    # mirror pair: two points at exactly the same distance from everything sampled before
    m = np.zeros((1, 64, 3), np.float32)
    m[0, :, 0] = np.linspace(1.0, 3.0, 64)
    m[0, 1] = [5.0, 0.0, 0.0]
    m[0, 2] = [3.0, 2.0, 0.0]; m[0, 3] = [3.0, -2.0, 0.0]      # equidistant from points 0 and 1
    ref = oracle.furthest_point_sampling(m, 32)
    np.testing.assert_array_equal(_ordered(_t(m, cuda), 32).cpu().numpy(), ref)


# ---------------------------------------------------------------------------- ball query
@pytest.mark.parametrize("B,N,M,r,S", [(2, 4096, 512, 0.2, 64), (2, 2048, 1024, 0.4, 32),
                                       (2, 1024, 512, 0.8, 16), (3, 512, 256, 1.2, 16),
                                       (1, 300, 77, 0.3, 5), (2, 20000, 2048, 0.2, 64)])
def test_ball_query_matches_oracle(cuda, B, N, M, r, S):
    xyz = _scene_xyz(B, N)
    inds = oracle.furthest_point_sampling(xyz, M)
    new_xyz = np.take_along_axis(xyz, inds[:, :, None].astype(np.int64), 1)
    ref = oracle.ball_query(new_xyz, xyz, r, S)
    got = _ext().ball_query(_t(new_xyz, cuda), _t(xyz, cuda), r, S).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("kind,B,N,M,r,S", [("surface", 8, 40000, 2048, 0.2, 64),
                                             ("uniform", 2, 30000, 1024, 0.3, 32),
                                             ("surface", 2, 80000, 2048, 0.2, 64),
                                             ("surface", 2, 9000, 300, 0.05, 8),
                                             ("surface", 2, 20000, 64, 0.8, 16)])
def test_ball_query_grid_matches_oracle_and_brute_force(cuda, monkeypatch, kind, B, N, M, r, S):
    """Large scenes take the grid-culled kernel; it must reproduce the oracle and the
    brute-force HIP kernel (BTR_BQ_IMPL=brute) bit for bit (BASELINE size included)."""
    xyz = _scene_xyz(B, N, kind=kind)
    inds = oracle.furthest_point_sampling(xyz[:, : min(N, 20000)], M)
    new_xyz = np.take_along_axis(xyz, inds[:, :, None].astype(np.int64), 1)
    new_xyz[:, -1] += 50.0   # a centre far outside the scene -> empty row
    new_xyz[:, -2, 2] -= 0.15  # and one below the floor
    ref = oracle.ball_query(new_xyz, xyz, r, S)
    got = _ext().ball_query(_t(new_xyz, cuda), _t(xyz, cuda), r, S).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    monkeypatch.setenv("BTR_BQ_IMPL", "brute")
    got = _ext().ball_query(_t(new_xyz, cuda), _t(xyz, cuda), r, S).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("kind,B,N,M,r,S,scale", [("surface", 8, 40000, 2048, 0.2, 64, 1.0),
                                                  ("surface", 2, 80000, 2048, 0.2, 64, 1.7),
                                                  ("uniform", 2, 30000, 1024, 0.3, 32, 1.0),
                                                  ("surface", 3, 5000, 700, 0.25, 16, 1.0),
                                                  ("surface", 2, 131000, 512, 0.1, 64, 2.0)])
def test_ball_query_over_the_fps_buckets(cuda, monkeypatch, kind, B, N, M, r, S, scale):
    """A ball query of the tensor the large-scene FPS has just sorted searches that sort
    (csrc/ball_query_bucket.hip) instead of building a grid: oracle-exact, identical to the
    grid path, including centres that are NOT points of the cloud."""
    xyz = _scene_xyz(B, N, first=20, kind=kind) * np.array([scale, scale, 1.0], np.float32)
    x = _t(xyz, cuda)
    inds = _ext().furthest_point_sampling(x, M)
    assert getattr(x, "_btr_fps_ws", None) is not None
    new = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    new[:, ::7] += 0.03                      # off-cloud centres
    new[:, 5] = 100.0                        # one centre far outside: empty ball -> zeros
    ref = oracle.ball_query(new.cpu().numpy(), xyz, r, S)
    before = dict(_ext().BQ_CALLS)
    got = _ext().ball_query(new, x, r, S).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    assert _ext().BQ_CALLS["buckets"] == before["buckets"] + 1, "the bucket path was not taken"
    monkeypatch.setenv("BTR_BQ_BUCKETS", "0")
    np.testing.assert_array_equal(_ext().ball_query(new, x, r, S).cpu().numpy(), ref)


@pytest.mark.parametrize("damage", ["zeroed", "one_box", "scenes_swapped", "other_launch"])
def test_ball_query_over_the_fps_buckets_checks_the_box_stamps(cuda, damage):
    """The FPS kernel leaves its bucket boxes in the dead counting-sort area of its workspace
    (include/btr_pointnet2.h, btr_ball_query_buckets CONTRACT).  The query does not assume that
    area survived: every box carries the FPS launch's epoch and its position, and a workgroup
    that finds a wrong stamp bounds the buckets from the sorted points instead.  Damage the area
    in four ways between the two calls -- the neighbour lists stay the oracle's."""
    B, N, M, r, S = 3, 20000, 512, 0.25, 32
    xyz = _scene_xyz(B, N, first=60, kind="surface")
    x = _t(xyz, cuda)
    inds = _ext().furthest_point_sampling(x, M)
    ws = x._btr_fps_ws[0]
    np_ = (N + 63) // 64 * 64
    nb = np_ // 64
    boxes = ws[20 * B * np_: 20 * B * np_ + 32 * B * nb].view(torch.float32).view(B, nb, 8)
    stamp = boxes[:, :, 3].clone().view(torch.int32)
    assert int(stamp.min()) == int(stamp.max()) != 0, "the FPS kernel stamps every box"
    if damage == "zeroed":
        boxes.zero_()
    elif damage == "one_box":      # one box of scene 1 replaced by a tiny far-away box
        boxes[1, nb // 2] = torch.tensor([9., 9., 9., 0., 9.1, 9.1, 9.1, 0.], device=cuda)
    elif damage == "scenes_swapped":   # right epoch, wrong place
        tmp = boxes[0].clone()
        boxes[0] = boxes[2]
        boxes[2] = tmp
    else:     # the whole workspace restored from ANOTHER launch's (another cloud's) workspace:
        # sorted points and boxes agree with each other, but not with the launch this thread
        # noted for the address -- the stamps say so and the query bounds the buckets itself
        x2 = _t(_scene_xyz(B, N, first=70, kind="surface"), cuda)
        _ext().furthest_point_sampling(x2, M)
        ws2 = x2._btr_fps_ws[0]
        assert ws2.data_ptr() != ws.data_ptr()
        assert int(ws2[20 * B * np_ + 12: 20 * B * np_ + 16].view(torch.int32)) != int(stamp[0, 0])
        ws.copy_(ws2)
        xyz = x2.cpu().numpy()
        x.copy_(x2)                # (bumps x._version: re-attach the workspace by hand)
        x._btr_fps_ws = (ws, x._version) + tuple(x._btr_fps_ws[2:])
    new = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    ref = oracle.ball_query(new.cpu().numpy(), xyz, r, S)
    before = dict(_ext().BQ_CALLS)
    got = _ext().ball_query(new, x, r, S).cpu().numpy()
    assert _ext().BQ_CALLS["buckets"] == before["buckets"] + 1, "the bucket path was not taken"
    np.testing.assert_array_equal(got, ref)


def test_ball_query_over_the_fps_buckets_with_its_own_box_pass():
    """The bucket boxes normally come from the FPS kernel (csrc/internal.hpp fps_boxes_lookup);
    `BTR_BQ_FPS_BOXES=0` (read once per process: a child process) makes the query compute them
    itself, as it does behind any other FPS kernel: same oracle-exact indices."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BTR_BQ_FPS_BOXES="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x",
                        "-k", "test_ball_query_over_the_fps_buckets and surface",
                        "-p", "no:cacheprovider"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=900)
    tail = r.stdout.decode(errors="replace")[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


def test_ball_query_grid_duplicates_and_dense(cuda):
    rng = np.random.default_rng(21)
    xyz = rng.uniform(0, 2, size=(1, 12000, 3)).astype(np.float32)
    xyz[0, 4000:9000] = xyz[0, 3999]          # 5001 identical points in one cell
    centres = np.concatenate([xyz[:, 3999:4000], xyz[:, :50]], 1)
    ref = oracle.ball_query(centres, xyz, 0.1, 32)
    got = _ext().ball_query(_t(centres, cuda), _t(xyz, cuda), 0.1, 32).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    flat = np.zeros((1, 10000, 3), np.float32)  # zero-extent scene
    flat[0, :, 0] = 1.0
    ref = oracle.ball_query(flat[:, :5], flat, 0.2, 16)
    got = _ext().ball_query(_t(flat[:, :5].copy(), cuda), _t(flat, cuda), 0.2, 16).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_ball_query_edge_cases(cuda):
    xyz = np.zeros((1, 70, 3), np.float32)
    xyz[0, :, 0] = np.arange(70) * 0.5
    centres = np.array([[[100.0, 0, 0], [0.0, 0, 0], [0.5, 0, 0], [10.0, 0, 0]]], np.float32)
    # empty ball -> zeros; short ball -> padded with first hit; d2 == r2 is excluded (strict <)
    ref = oracle.ball_query(centres, xyz, 0.5, 4)
    got = _ext().ball_query(_t(centres, cuda), _t(xyz, cuda), 0.5, 4).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    np.testing.assert_array_equal(got[0, 0], [0, 0, 0, 0])
    np.testing.assert_array_equal(got[0, 2], [1, 1, 1, 1])
    ref = oracle.ball_query(centres, xyz, 1.25, 4)   # full ball -> first 4 in index order
    got = _ext().ball_query(_t(centres, cuda), _t(xyz, cuda), 1.25, 4).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


# -------------------------------------------------------------------- gather / group (+grad)
@pytest.mark.parametrize("B,C,N,M", [(2, 3, 4096, 2048), (3, 7, 100, 33), (2, 288, 1024, 256)])
def test_gather_points(cuda, B, C, N, M):
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, M)).astype(np.int32)
    got = _ext().gather_points(_t(pts, cuda), _t(idx, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.gather_points(pts, idx))
    go = rng.standard_normal((B, C, M)).astype(np.float32)
    idx[:, : M // 2] = idx[:, :1]  # heavy collisions
    got = _ext().gather_points_grad(_t(go, cuda), _t(idx, cuda), N).cpu().numpy()
    np.testing.assert_allclose(got, oracle.gather_points_grad(go, idx, N), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,C,N,M,S", [(2, 4, 4096, 512, 64), (2, 131, 2048, 256, 32),
                                       (1, 259, 512, 64, 16), (3, 5, 50, 7, 3)])
def test_group_points(cuda, B, C, N, M, S):
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, S)).astype(np.int32)
    idx[:, :, S // 2:] = idx[:, :, :1]  # padded rows repeat their first neighbour
    got = _ext().group_points(_t(pts, cuda), _t(idx, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.group_points(pts, idx))
    go = rng.standard_normal((B, C, M, S)).astype(np.float32)
    got = _ext().group_points_grad(_t(go, cuda), _t(idx, cuda), N).cpu().numpy()
    np.testing.assert_allclose(got, oracle.group_points_grad(go, idx, N), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------------ three_nn / interp
@pytest.mark.parametrize("B,n,m", [(2, 512, 256), (2, 1024, 512), (1, 100, 2), (2, 33, 3)])
def test_three_nn(cuda, B, n, m):
    rng = np.random.default_rng(2)
    unknown = rng.uniform(0, 4, (B, n, 3)).astype(np.float32)
    known = rng.uniform(0, 4, (B, m, 3)).astype(np.float32)
    if m >= 3:
        known[:, 2] = known[:, 0]  # exact tie: earliest index must win
    d_ref, i_ref = oracle.three_nn(unknown, known)
    d, i = _ext().three_nn(_t(unknown, cuda), _t(known, cuda))
    np.testing.assert_array_equal(i.cpu().numpy(), i_ref)
    np.testing.assert_array_equal(d.cpu().numpy(), d_ref)  # inf where m < 3


@pytest.mark.parametrize("B,C,m,n", [(2, 256, 256, 512), (2, 256, 512, 1024), (1, 3, 10, 7),
                                     (2, 5, 9000, 3000)])  # m > 8192: global-atomic inversion
def test_three_interpolate(cuda, B, C, m, n):
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((B, C, m)).astype(np.float32)
    idx = rng.integers(0, m, (B, n, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (B, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    got = _ext().three_interpolate(_t(pts, cuda), _t(idx, cuda), _t(w, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.three_interpolate(pts, idx, w))
    go = rng.standard_normal((B, C, n)).astype(np.float32)
    got = _ext().three_interpolate_grad(_t(go, cuda), _t(idx, cuda), _t(w, cuda), m).cpu().numpy()
    np.testing.assert_allclose(got, oracle.three_interpolate_grad(go, idx, w, m), rtol=1e-4,
                               atol=1e-5)


def test_reference_gradcheck_three_interpolate(cuda):
    """The reference's only test (pointnet2_test.py:18-30): gradcheck of three_interpolate on
    a (1,2,4) tensor with fixed idx/weight, atol=rtol=1e-1."""
    from backtoreality_amd.pointnet2 import pointnet2_utils
    feats = torch.tensor([[[1., 2., 3., 4.], [5., 6., 7., 8.]]], device=cuda, requires_grad=True)
    idx = torch.tensor([[[0, 1, 2], [1, 2, 3]]], dtype=torch.int32, device=cuda)
    weight = torch.tensor([[[1., 1., 1.], [2., 2., 2.]]], device=cuda)
    assert torch.autograd.gradcheck(
        lambda f: pointnet2_utils.three_interpolate(f, idx, weight), (feats,), atol=1e-1,
        rtol=1e-1, eps=1e-2, nondet_tol=1e-3)


def test_errors_match_reference(cuda):
    e = _ext()
    x = torch.zeros(1, 10, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        e.furthest_point_sampling(x, 2)
    xc = torch.zeros(1, 10, 3, device=cuda)
    with pytest.raises(RuntimeError, match="contiguous"):
        e.furthest_point_sampling(xc.transpose(1, 2).transpose(1, 2)[:, ::2], 2)
    with pytest.raises(RuntimeError, match="int tensor"):
        e.gather_points(torch.zeros(1, 3, 10, device=cuda), torch.zeros(1, 4, device=cuda))
    with pytest.raises(RuntimeError, match="float tensor"):
        e.three_nn(xc.double(), xc)


def test_gather_rows_matches_the_transpose_gather_route(cuda):
    """pointnet2_utils.gather_rows (one launch) against the reference's transpose +
    gather_operation + transpose for new_xyz (pointnet2_modules.py:238-240), values and
    gradient, with repeated indices."""
    from backtoreality_amd.pointnet2 import pointnet2_utils as U
    g = torch.Generator().manual_seed(0)
    for (B, N, M, C) in ((2, 1000, 300, 3), (3, 64, 64, 5), (1, 7, 20, 1), (4, 1024, 256, 288),
                         (2, 20000, 2048, 3)):
        src = torch.randn(B, N, C, generator=g).to(cuda).requires_grad_(True)
        idx = torch.randint(0, N, (B, M), generator=g).int().to(cuda)
        w = torch.randn(B, M, C, generator=g).to(cuda)
        out = U.gather_rows(src, idx)
        (out * w).sum().backward()
        g1, src.grad = src.grad.clone(), None
        ref = U.gather_operation(src.transpose(1, 2).contiguous(), idx).transpose(1, 2)
        (ref * w).sum().backward()
        assert torch.equal(out, ref)
        assert torch.allclose(g1, src.grad, rtol=1e-5, atol=1e-6)
