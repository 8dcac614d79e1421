"""GPU (-m gpu): PASCAL VOC evaluation — the device-side matching, curves and AP of oneshotdet_amd/evaluation.py
(osd_voc_match / osd_voc_curves / osd_voc_ap) against tests/golden/voc_eval.npz, recorded through the REAL reference's eval_detection_voc / calc_detection_voc_prec_rec
(data/datasets/evaluation/voc/voc_eval.py:48-216) on synthetic detections: AP of both metrics and the whole precision / recall
curves to 1e-6 (they are ratios of integers: in fact exactly), the match flags against the oracle's per-image restatement, and
edge cases (no detections, no ground truth of a class, difficult boxes, duplicates of one box, an empty dataset entry)."""
import numpy as np
import pytest
import torch

import golden_utils as gu
from oracle import voc_eval_ref as ov

pytestmark = pytest.mark.gpu


def boxlists(preds, gts, size=(800, 600)):
    from oneshotdet_amd.modules import BoxList
    pbl, gbl = [], []
    for (pb, pl, ps), (gb, gl, gd) in zip(preds, gts):
        a = BoxList(torch.from_numpy(np.asarray(pb, np.float32).reshape(-1, 4)), size)
        a.add_field("labels", torch.from_numpy(np.asarray(pl, np.int64)))
        a.add_field("scores", torch.from_numpy(np.asarray(ps, np.float32)))
        b = BoxList(torch.from_numpy(np.asarray(gb, np.float32).reshape(-1, 4)), size)
        b.add_field("labels", torch.from_numpy(np.asarray(gl, np.int64)))
        b.add_field("difficult", torch.from_numpy(np.asarray(gd, np.uint8)))
        pbl.append(a)
        gbl.append(b)
    return pbl, gbl


def test_voc_ap_equals_the_reference_fixture():
    from oneshotdet_amd import evaluation as ev
    f = gu.load("voc_eval.npz")
    preds, gts = gu.voc_eval_inputs()
    pbl, gbl = boxlists(preds, gts)
    for tag, use07 in (("ap07", True), ("ap_area", False)):
        r = ev.eval_detection_voc(pbl, gbl, iou_thresh=0.5, use_07_metric=use07)
        np.testing.assert_allclose(np.nan_to_num(r["ap"], nan=-1.0), np.nan_to_num(f[tag], nan=-1.0), rtol=0, atol=1e-6)
        assert abs(r["map"] - float(f[tag + "_map"])) <= 1e-6
    prec, rec = ev.calc_detection_voc_prec_rec(gbl, pbl, 0.5)
    assert len(prec) == int(f["n_classes"])
    for l in range(len(prec)):
        if "prec.%d" % l in f.files:
            np.testing.assert_array_equal(np.nan_to_num(prec[l]), np.nan_to_num(f["prec.%d" % l]))
        else:
            assert prec[l] is None
        if "rec.%d" % l in f.files:
            np.testing.assert_array_equal(rec[l], f["rec.%d" % l])


def test_voc_ap_from_curve_lists_long_classes_and_missing_curves():
    """calc_detection_voc_ap on the reference's argument format (lists of per-class arrays, None for absent classes) against the
    oracle's restatement of voc_eval.py:161-216: classes longer than one 256-element chunk (the carried suffix maximum and prefix
    counts), NaN precision heads (only ignored detections so far), a class without a recall curve, an empty class."""
    from oneshotdet_amd import evaluation as ev
    rng = np.random.RandomState(5)
    prec, rec = [], []
    for n, npos in ((1000, 300), (257, 5), (0, 3), (40, 0), (256, 1000)):
        flags = rng.choice([-1, 0, 1], size=n, p=[0.1, 0.5, 0.4]).astype(np.int8)
        if n:
            flags[:3] = -1
        tp, fp = np.cumsum(flags == 1), np.cumsum(flags == 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            prec.append(tp / (fp + tp))
        rec.append(tp / npos if npos > 0 else None)
    prec.insert(2, None)
    rec.insert(2, None)
    for use07 in (True, False):
        got = ev.calc_detection_voc_ap(prec, rec, use_07_metric=use07)
        want = ov.calc_detection_voc_ap(prec, rec, use_07_metric=use07)
        np.testing.assert_allclose(np.nan_to_num(got, nan=-1.0), np.nan_to_num(want, nan=-1.0), rtol=0, atol=1e-12)
    # 11-point metric: the reference's own order of additions, so the same bits
    np.testing.assert_array_equal(np.nan_to_num(ev.calc_detection_voc_ap(prec, rec, True), nan=-1.0),
                                  np.nan_to_num(ov.calc_detection_voc_ap(prec, rec, True), nan=-1.0))


def test_voc_match_flags_equal_the_oracle_per_image():
    """Device flags, reordered the way the reference orders a class's detections, equal the restatement's; thresholds 0.3 / 0.5 / 0.75."""
    from oneshotdet_amd import evaluation as ev
    preds, gts = gu.voc_eval_inputs(seed=3, n_images=9, max_det=200, max_gt=30, n_classes=4)
    pbl, gbl = boxlists(preds, gts)
    for thr in (0.3, 0.5, 0.75):
        flags = ev.voc_match(pbl, gbl, thr)
        for (mt, mg), (pb, pl, ps), (gb, gl, gd) in zip(flags, preds, gts):
            want = ov.match_image(pb, pl, ps, gb, gl, gd, thr)
            for l, (ws, wm, _) in want.items():
                sel = np.nonzero(pl == l)[0]
                order = sel[np.argsort(ps[sel], kind="stable")[::-1]]
                np.testing.assert_array_equal(mt[order], wm)
        r = ev.eval_detection_voc(pbl, gbl, thr, False)
        o = ov.eval_detection_voc(preds, gts, thr, False)
        np.testing.assert_allclose(np.nan_to_num(r["ap"], nan=-1.0), np.nan_to_num(o["ap"], nan=-1.0), rtol=0, atol=1e-9)


def test_voc_edge_cases():
    from oneshotdet_amd import evaluation as ev
    box = np.array([[10.0, 10.0, 50.0, 50.0]], np.float32)
    e4, e0 = np.zeros((0, 4), np.float32), np.zeros((0,), np.int64)
    # three duplicates of one box: only the highest-scoring one is a true positive; a difficult box is ignored; an image
    # without detections still counts its positives; an image without ground truth makes its detections false positives
    preds = [(np.repeat(box, 3, 0), np.array([1, 1, 1]), np.array([0.5, 0.9, 0.7], np.float32)),
             (box, np.array([1]), np.array([0.8], np.float32)),
             (e4, e0, np.zeros((0,), np.float32)),
             (box + 200, np.array([1]), np.array([0.95], np.float32))]
    gts = [(box, np.array([1]), np.array([0], np.uint8)),
           (box, np.array([1]), np.array([1], np.uint8)),
           (box, np.array([1]), np.array([0], np.uint8)),
           (e4, e0, np.zeros((0,), np.uint8))]
    pbl, gbl = boxlists(preds, gts)
    flags = ev.voc_match(pbl, gbl, 0.5)
    np.testing.assert_array_equal(flags[0][0], [0, 1, 0])
    np.testing.assert_array_equal(flags[1][0], [-1])
    assert len(flags[2][0]) == 0
    np.testing.assert_array_equal(flags[3][0], [0])
    for use07 in (True, False):
        r = ev.eval_detection_voc(pbl, gbl, 0.5, use07)
        o = ov.eval_detection_voc(preds, gts, 0.5, use07)
        np.testing.assert_allclose(np.nan_to_num(r["ap"], nan=-1.0), np.nan_to_num(o["ap"], nan=-1.0), rtol=0, atol=1e-12)
    from oneshotdet_amd import _lib
    big = (np.zeros((600, 4), np.float32), np.ones(600, np.int64), np.zeros(600, np.uint8))
    with pytest.raises(_lib.OsdError):
        ev.voc_match(*boxlists([preds[0]], [big]), 0.5)
