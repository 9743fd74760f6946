"""The evaluation-path oracle (oracle/eval_oracle.py, numpy) against outputs of the REFERENCE's
ap_helper / nms / box_util / eval_det (tests/golden/eval_ap.npz, tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import eval_oracle as eo
from tests import eval_common as ec


@pytest.fixture(scope="module")
def gold():
    return np.load(ec.GOLDEN, allow_pickle=False)


def test_box3d_iou_matches_reference(gold):
    got = np.array([eo.box3d_iou(a, b) for a, b in zip(gold["iou_c1"], gold["iou_c2"])])
    assert np.allclose(got, gold["iou_expected"], rtol=1e-9, atol=1e-12)
    assert got.max() > 0.8 and (got == 0).sum() >= 50  # near-identical and disjoint pairs


def test_nms_matches_reference(gold):
    for dim in (3, 2):
        boxes = gold["nms%d_boxes" % dim]
        for old in (False, True):
            pick = eo.nms_boxes(boxes[:, :2 * dim], boxes[:, 2 * dim], 0.25, old)
            assert pick == gold["nms%d_pick_%s" % (dim, "old" if old else "iou")].tolist()
    boxes = gold["nms3_boxes"]
    pick = eo.nms_boxes(boxes[:, :6], boxes[:, 6], 0.25, False, cls=gold["nms3_cls"])
    assert pick == gold["nms3_pick_samecls"].tolist()


@pytest.mark.parametrize("tag", ["scannet", "matterport"])
@pytest.mark.parametrize("cname", ["train", "empty_old", "bev"])
def test_parse_and_ap_match_reference(gold, tag, cname):
    cfg, case = ec.make_case(tag)
    cd = dict(ec.EVAL_CONFIGS[cname], dataset_config=cfg)
    case = {k: v.numpy() for k, v in case.items()}
    pred, pred_mask, corners = eo.parse_predictions(case, cd)
    key = "%s_%s_" % (tag, cname)
    assert np.array_equal(pred_mask.astype(np.uint8), gold[key + "pred_mask"])
    ec.check_lists(gold, key, pred, 1e-6)
    gt = eo.parse_groundtruths(case, cd)
    if cname == "train":
        assert np.allclose(corners, gold[key + "corners"], rtol=0, atol=1e-12)
        assert [len(g) for g in gt] == gold[key + "gt_n"].tolist()
        assert np.allclose(np.stack([b for g in gt for _, b in g]), gold[key + "gt_corners"],
                           rtol=0, atol=1e-12)
    for thr in (0.25, 0.5):
        ec.check_metrics(gold, key, thr, eo.metrics(pred, gt, thr), 1e-9)
