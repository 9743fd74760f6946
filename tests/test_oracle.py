"""Known-answer tests for the CPU oracle, written straight from the semantics of the
reference's CUDA kernels (no golden vectors exist in the reference for these ops):
each case cites the .cu lines it pins."""
import numpy as np
import pytest

import oracle


def test_opt_n_threads_matches_reference_rule():
    # include/cuda_utils.h:20-24: 2^floor(log2 n) clamped to [1, 512]
    for n, want in [(1, 1), (2, 2), (3, 2), (64, 64), (255, 128), (256, 256), (511, 256),
                    (512, 512), (1024, 512), (2048, 512), (40000, 512), (80000, 512)]:
        assert oracle.opt_n_threads(n) == want


def test_fps_first_index_zero_and_farthest_next():
    # sampling_gpu.cu:90-92 (idxs[0] = 0), :108-114 (arg-max of min distance)
    xyz = np.array([[[1, 1, 1], [2, 1, 1], [9, 1, 1], [5, 1, 1], [1.5, 1, 1]]], np.float32)
    out = oracle.furthest_point_sampling(xyz, 4)
    np.testing.assert_array_equal(out, [[0, 2, 3, 1]])


def test_fps_skips_points_near_origin():
    # sampling_gpu.cu:105-106: x^2+y^2+z^2 <= 1e-3 -> `continue`: never updated, never chosen,
    # even when it would be the farthest point
    xyz = np.array([[[5, 5, 5], [0.01, 0.01, 0.01], [4, 5, 5], [5, 4.5, 5]]], np.float32)
    out = oracle.furthest_point_sampling(xyz, 3)
    assert 1 not in out[0].tolist()
    np.testing.assert_array_equal(out, [[0, 2, 3]])
    # magnitude just above the threshold competes
    xyz[0, 1] = [0.02, 0.02, 0.02]
    assert oracle.furthest_point_sampling(xyz, 2)[0, 1] == 1


def test_fps_nothing_competes_gives_zero():
    # best=-1, besti=0 survive the whole reduction (:95-96) -> index 0 every time
    xyz = np.zeros((2, 50, 3), np.float32)
    np.testing.assert_array_equal(oracle.furthest_point_sampling(xyz, 5), np.zeros((2, 5)))


def test_fps_m_le_zero_is_noop():
    xyz = np.ones((1, 10, 3), np.float32)
    assert oracle.furthest_point_sampling(xyz, 0).shape == (1, 0)  # :78


def test_fps_tie_break_is_bit_reversed_slot_order():
    # Two points equally far: with block size 4 the tree (:64-70, :121-174) prefers slot
    # bitrev2(k mod 4): slots 0,2,1,3.  Points 5 (slot 1) and 6 (slot 2) tie -> 6 wins;
    # with block size 1 (no tree) the first strict max in scan order wins -> 5.
    xyz = np.ones((1, 8, 3), np.float32)
    xyz[0, 5] = [4, 1, 1]
    xyz[0, 6] = [4, 1, 1]
    assert oracle.furthest_point_sampling(xyz, 2, block_size=4)[0, 1] == 6
    assert oracle.furthest_point_sampling(xyz, 2, block_size=1)[0, 1] == 5
    assert oracle.furthest_point_sampling(xyz, 2, block_size=8)[0, 1] == 6  # slots 5->5b,6->3b
    # same thread (same slot): the smaller index wins (strict `>` in the scan, :113-114)
    xyz = np.ones((1, 12, 3), np.float32)
    xyz[0, 2] = [4, 1, 1]
    xyz[0, 6] = [4, 1, 1]
    xyz[0, 10] = [4, 1, 1]
    assert oracle.furthest_point_sampling(xyz, 2, block_size=4)[0, 1] == 2


def test_fps_closed_form_equals_block_emulation():
    rng = np.random.default_rng(3)
    base = rng.uniform(0.1, 2.0, (2, 333, 3)).astype(np.float32)
    xyz = np.concatenate([base, base, base[:, :100]], 1)  # many exact ties
    for bs in (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        a = oracle.furthest_point_sampling(xyz, 200, block_size=bs)
        b = oracle.furthest_point_sampling(xyz, 200, block_size=bs, closed_form=True)
        np.testing.assert_array_equal(a, b)


def test_fps_prefix_invariant():
    # backbone_module.py:113-132 relies on FPS(FPS-ordered prefix) == arange
    rng = np.random.default_rng(4)
    xyz = rng.uniform(0.5, 5, (1, 3000, 3)).astype(np.float32)
    inds = oracle.furthest_point_sampling(xyz, 512)
    sub = xyz[:, inds[0]]
    np.testing.assert_array_equal(oracle.furthest_point_sampling(sub, 256)[0], np.arange(256))


def test_ball_query_rules():
    xyz = np.zeros((1, 70, 3), np.float32)
    xyz[0, :, 0] = np.arange(70) * 0.5
    c = np.array([[[100.0, 0, 0], [0.0, 0, 0], [0.5, 0, 0], [10.0, 0, 0]]], np.float32)
    out = oracle.ball_query(c, xyz, 0.5, 4)
    np.testing.assert_array_equal(out[0, 0], [0, 0, 0, 0])      # empty: stays zero
    np.testing.assert_array_equal(out[0, 1], [0, 0, 0, 0])      # only point 0 (d2=.25 == r2 excluded)
    np.testing.assert_array_equal(out[0, 2], [1, 1, 1, 1])      # strict `<`: neighbours at 0.5 out
    np.testing.assert_array_equal(out[0, 3], [20, 20, 20, 20])
    out = oracle.ball_query(c, xyz, 1.25, 4)
    np.testing.assert_array_equal(out[0, 1], [0, 1, 2, 0])      # short: padded with FIRST hit
    np.testing.assert_array_equal(out[0, 3], [18, 19, 20, 21])  # full: first nsample in order
    # radius2 is computed in f32 (ball_query_gpu.cu:27)
    r = np.float32(0.1)
    xyz2 = np.array([[[0, 0, 0], [np.sqrt(np.float64(r * r)), 0, 0]]], np.float32)
    out = oracle.ball_query(np.zeros((1, 1, 3), np.float32), xyz2, float(r), 2)
    d2 = np.float32(xyz2[0, 1, 0]) * np.float32(xyz2[0, 1, 0])
    assert (out[0, 0, 1] == 1) == bool(d2 < r * r)


def test_three_nn_rules():
    known = np.array([[[0, 0, 0], [1, 0, 0], [1, 0, 0], [5, 0, 0]]], np.float32)
    unknown = np.array([[[0.9, 0, 0]]], np.float32)
    d, i = oracle.three_nn(unknown, known)
    np.testing.assert_array_equal(i[0, 0], [1, 2, 0])  # tie 1 vs 2 -> earliest (strict <)
    np.testing.assert_allclose(d[0, 0], [0.01, 0.01, 0.81], rtol=1e-5)
    d, i = oracle.three_nn(unknown, known[:, :2])      # m < 3: 1e40 -> inf, index 0
    assert np.isinf(d[0, 0, 2]) and i[0, 0, 2] == 0


def test_scatter_adds_accumulate_repeated_indices():
    go = np.ones((1, 2, 4), np.float32)
    idx = np.array([[1, 1, 1, 3]], np.int32)
    out = oracle.gather_points_grad(go, idx, 5)
    np.testing.assert_array_equal(out[0, 0], [0, 3, 0, 1, 0])
    go = np.arange(8, dtype=np.float32).reshape(1, 1, 2, 4)
    idx = np.array([[[0, 0, 2, 2], [2, 2, 2, 1]]], np.int32)
    out = oracle.group_points_grad(go, idx, 3)
    np.testing.assert_array_equal(out[0, 0], [1, 7, 2 + 3 + 4 + 5 + 6])
    w = np.array([[[0.5, 0.25, 0.25]]], np.float32)
    out = oracle.three_interpolate_grad(np.full((1, 1, 1), 4, np.float32),
                                        np.array([[[2, 2, 0]]], np.int32), w, 3)
    np.testing.assert_array_equal(out[0, 0], [1, 0, 3])


def test_reference_interpolate_vector():
    # values of the reference's own test (pointnet2_test.py:18-30)
    feats = np.array([[[1, 2, 3, 4], [5, 6, 7, 8]]], np.float32)
    idx = np.array([[[0, 1, 2], [1, 2, 3]]], np.int32)
    w = np.array([[[1, 1, 1], [2, 2, 2]]], np.float32)
    out = oracle.three_interpolate(feats, idx, w)
    np.testing.assert_array_equal(out, [[[6, 18], [18, 42]]])


# ------------------------------------------------- the three roundings of the squared distance
@pytest.fixture
def restore_mode():
    yield
    oracle.set_fmad(1)


def test_distance_modes_follow_their_formula(restore_mode):
    """three_nn returns the squared distance itself: each oracle mode must equal its formula
    evaluated in exact rational arithmetic with one rounding per machine operation."""
    from mode_cases import sq3_exact
    rng = np.random.default_rng(11)
    u = rng.uniform(-3, 3, (1, 300, 3)).astype(np.float32)
    k = rng.uniform(-3, 3, (1, 1, 3)).astype(np.float32)
    got = {}
    for mode in (0, 1, 2):
        oracle.set_fmad(mode)
        assert oracle.lib().btr_oracle_fmad_mode() == mode
        d2, idx = oracle.three_nn(u, k)
        got[mode] = d2[0, :, 0].copy()
        dx, dy, dz = (u[0, :, a] - k[0, 0, a] for a in range(3))   # float32 subtractions
        want = np.array([sq3_exact(dx[i], dy[i], dz[i], mode) for i in range(300)], np.float32)
        np.testing.assert_array_equal(got[mode], want)
    assert (got[0] != got[1]).any() and (got[1] != got[2]).any() and (got[0] != got[2]).any()


def test_distance_modes_change_indices_on_crafted_inputs(restore_mode):
    """On ordinary scenes the modes agree; on these inputs they must not (otherwise the GPU
    twins of this test could not tell the libraries apart)."""
    from mode_cases import shell_ball_case, sphere_cloud
    xyz = sphere_cloud(3, 4096)
    centres, pts, radius = shell_ball_case(4)
    fps, bq = {}, {}
    for mode in (0, 1, 2):
        oracle.set_fmad(mode)
        fps[mode] = oracle.furthest_point_sampling(xyz, 256)
        bq[mode] = oracle.ball_query(centres, pts, radius, 64)
    for a, b in ((0, 1), (1, 2), (0, 2)):
        assert (fps[a] != fps[b]).any(), ("fps", a, b)
        assert (bq[a] != bq[b]).any(), ("ball_query", a, b)
