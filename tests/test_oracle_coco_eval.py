"""CPU: known-answer cases for oracle/coco_eval_ref.py, the restatement of pycocotools' COCOeval (bbox) that the device path of
oneshotdet_amd.evaluation.evaluate_predictions_on_coco is checked against.  pycocotools itself is absent (PARITY UNPINNED: see the
oracle's header); these hand-computed configurations are what pins the restatement:
precision is sampled at 101 recall thresholds, so a detector that finds 1 of 2 boxes before its first false positive scores
51 / 101 at that IoU threshold (recall thresholds 0 .. 0.5 see precision 1, the rest 0)."""
import numpy as np

from oracle import coco_eval_ref as oc


def ann(img, cat, box, crowd=0, area=None):
    return dict(image_id=img, category_id=cat, bbox=list(map(float, box)), area=float(box[2] * box[3] if area is None else area), iscrowd=crowd)


def det(img, cat, box, score):
    return dict(image_id=img, category_id=cat, bbox=list(map(float, box)), score=float(score))


def test_perfect_detection():
    r = oc.evaluate([ann(1, 1, [10, 10, 100, 100])], [det(1, 1, [10, 10, 100, 100], 0.9)])
    s = r["stats"]
    # (precision = tp / (tp + fp + eps): 1 - 2e-16 where it is "1")
    np.testing.assert_allclose(s[[0, 1, 2, 5]], 1.0, rtol=0, atol=1e-12)        # AP, AP50, AP75, AP large (area 10,000 > 96^2)
    assert s[3] == -1.0 and s[4] == -1.0                                        # no small / medium ground truth
    assert s[6] == 1.0 and s[7] == 1.0 and s[8] == 1.0


def test_one_of_two_boxes_before_the_false_positive():
    gts = [ann(1, 1, [0, 0, 100, 100]), ann(1, 1, [200, 200, 100, 100])]
    dts = [det(1, 1, [0, 0, 100, 100], 0.9), det(1, 1, [200, 200, 100, 62], 0.8)]       # second: IoU 0.62 with its box
    assert abs(oc.bb_iou(dts[1]["bbox"], gts[1]["bbox"], False) - 0.62) < 1e-12
    r = oc.evaluate(gts, dts)
    s = r["stats"]
    # thresholds .5 .55 .6: both detections true -> AP 1; .65 ...95 (7 of them): tp = [1, 1], fp = [0, 1] -> 51 / 101
    np.testing.assert_allclose(s[0], (3 * 1.0 + 7 * 51 / 101) / 10, rtol=0, atol=1e-12)
    np.testing.assert_allclose(s[1], 1.0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(s[2], 51 / 101, rtol=0, atol=1e-12)
    np.testing.assert_allclose(s[8], (3 * 1.0 + 7 * 0.5) / 10, rtol=0, atol=1e-12)      # AR at 100 detections
    assert s[6] == 0.5                                                                  # AR at 1 detection per image: the 0.9 one only


def test_crowd_region_absorbs_detections_and_is_not_a_positive():
    # one real box found; three detections inside a crowd region: ignored (neither true nor false positives); the crowd box is
    # not counted among the positives either -> AP 1.  Against a crowd box the union is the detection's own area.
    gts = [ann(1, 1, [0, 0, 50, 50]), ann(1, 1, [300, 300, 200, 200], crowd=1)]
    dts = [det(1, 1, [0, 0, 50, 50], 0.5)] + [det(1, 1, [310 + 10 * i, 310, 40, 40], 0.9 - 0.1 * i) for i in range(3)]
    assert oc.bb_iou(dts[1]["bbox"], gts[1]["bbox"], True) == 1.0
    r = oc.evaluate(gts, dts)
    np.testing.assert_allclose(r["stats"][0], 1.0, rtol=0, atol=1e-12)
    assert r["stats"][8] == 1.0
    # without the crowd flag the same three detections are false positives ahead of the true one
    gts[1]["iscrowd"] = 0
    assert oc.evaluate(gts, dts)["stats"][0] < 0.6


def test_area_ranges_and_empty_categories():
    gts = [ann(1, 1, [0, 0, 20, 20]), ann(2, 1, [0, 0, 50, 50]), ann(2, 2, [100, 100, 200, 200])]
    dts = [det(1, 1, [0, 0, 20, 20], 0.9), det(2, 1, [0, 0, 50, 50], 0.8)]              # category 2 is never detected
    r = oc.evaluate(gts, dts)
    s = r["stats"]
    np.testing.assert_allclose(s[[3, 4]], 1.0, rtol=0, atol=1e-12)      # small (400) and medium (2,500) boxes of category 1 found
    assert s[5] == 0.0                                    # the large box (category 2) missed: precision 0 everywhere
    np.testing.assert_allclose(s[0], 0.5, rtol=0, atol=1e-12)       # mean over the two categories
    assert r["precision"].shape == (10, 101, 2, 4, 3) and r["recall"].shape == (10, 2, 4, 3)
    # an image with detections but no ground truth of the category: false positives only
    r2 = oc.evaluate(gts, dts + [det(3, 1, [0, 0, 30, 30], 0.95)], img_ids=[1, 2, 3])
    assert r2["stats"][0] < s[0]


def test_more_than_100_detections_are_cut_and_maxdets_slices():
    gts = [ann(1, 1, [0, 0, 10, 10])]
    dts = [det(1, 1, [500, 500, 10, 10], 1.0 - 1e-3 * i) for i in range(150)] + [det(1, 1, [0, 0, 10, 10], 0.5)]
    r = oc.evaluate(gts, dts)                             # the true detection is rank 151: beyond the 100 kept
    assert r["stats"][0] == 0.0 and r["stats"][8] == 0.0
