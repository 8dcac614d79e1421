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


# ---- COCO-style evaluation (oneshotdet_amd.evaluation.evaluate_predictions_on_coco) against oracle/coco_eval_ref.py.  PARITY
# UNPINNED: pycocotools is absent; the oracle restates its published algorithm and is pinned by hand-computed cases
# (tests/test_oracle_coco_eval.py).  What is checked here is that the device matching + host integration reproduce the oracle
# exactly: every entry of the precision / recall tables and the 12 summary numbers.
def _coco_dataset(seed, n_images=12, n_cats=3, max_gt=9, max_det=130):
    rng = np.random.RandomState(seed)
    gts, dts = [], []
    for img in range(1, n_images + 1):
        for cat in range(1, n_cats + 1):
            ng = rng.randint(0, max_gt + 1) if rng.rand() > 0.2 else 0
            boxes = []
            for _ in range(ng):
                w, h = rng.choice([12.0, 40.0, 150.0]) * (0.5 + rng.rand()), rng.choice([12.0, 40.0, 150.0]) * (0.5 + rng.rand())
                x, y = rng.rand() * 600, rng.rand() * 400
                boxes.append([x, y, w, h])
                crowd = int(rng.rand() < 0.15)
                # the annotation's area is not always w * h (segment areas): keep some near the range borders
                area = w * h * (1.0 if rng.rand() < 0.7 else 0.6)
                gts.append(dict(image_id=img, category_id=cat, bbox=[x, y, w, h], area=area, iscrowd=crowd))
            nd = rng.randint(0, max_det + 1) if rng.rand() > 0.15 else 0
            for _ in range(nd):
                if boxes and rng.rand() < 0.6:          # a jittered copy of a ground-truth box
                    b = np.array(boxes[rng.randint(len(boxes))]) + rng.randn(4) * np.array([4, 4, 6, 6])
                    b[2:] = np.maximum(b[2:], 1.0)
                else:
                    b = np.array([rng.rand() * 600, rng.rand() * 400, 5 + rng.rand() * 200, 5 + rng.rand() * 200])
                score = float(np.round(rng.rand(), 2)) if rng.rand() < 0.3 else float(rng.rand())      # some tied scores
                dts.append(dict(image_id=img, category_id=cat, bbox=b.tolist(), score=score))
    return gts, dts


@pytest.mark.parametrize("seed", [0, 1])
def test_coco_eval_equals_the_oracle(seed):
    from oneshotdet_amd import evaluation as ev
    from oracle import coco_eval_ref as oc
    gts, dts = _coco_dataset(seed)
    img_ids, cat_ids = list(range(1, 13)), [1, 2, 3, 4]                  # category 4: no ground truth, no detections
    want = oc.evaluate(gts, dts, img_ids=img_ids, cat_ids=cat_ids)
    got = ev.evaluate_predictions_on_coco({"annotations": gts, "images": [{"id": i} for i in img_ids],
                                           "categories": [{"id": c} for c in cat_ids]}, dts)
    np.testing.assert_array_equal(got.eval["precision"], want["precision"])
    np.testing.assert_array_equal(got.eval["recall"], want["recall"])
    np.testing.assert_array_equal(got.stats, want["stats"])
    assert (want["precision"][:, :, 3] == -1).all() and 0 < want["stats"][0] < 1
    assert set(got.results()["bbox"]) == {"AP", "AP50", "AP75", "APs", "APm", "APl"}


def test_coco_match_flags_equal_the_oracle_per_pair_and_edge_cases():
    from oneshotdet_amd import _lib, evaluation as ev
    from oracle import coco_eval_ref as oc
    gts, dts = _coco_dataset(7, n_images=4, n_cats=2)
    by_g, by_d = {}, {}
    for g in gts:
        by_g.setdefault((g["image_id"], g["category_id"]), []).append(g)
    for d in dts:
        by_d.setdefault((d["image_id"], d["category_id"]), []).append(d)
    keys = sorted(set(by_g) | set(by_d))
    pairs = []
    for k in keys:
        g, d = by_g.get(k, []), by_d.get(k, [])
        order = np.argsort([-x["score"] for x in d], kind="mergesort")[:100]
        pairs.append((np.array([d[i]["bbox"] for i in order], np.float64).reshape(-1, 4), np.array([x["bbox"] for x in g], np.float64).reshape(-1, 4),
                      np.array([x["area"] for x in g], np.float64), np.array([x["iscrowd"] for x in g], np.uint8)))
    out = ev.coco_match(pairs)
    for k, (dm, di, gi) in zip(keys, out):
        for a, rng_ in enumerate(oc.AREA_RNG):
            e = oc.evaluate_img(by_g.get(k, []), by_d.get(k, []), rng_, 100)
            np.testing.assert_array_equal(dm[a] > 0, e["dt_matches"] > 0)
            np.testing.assert_array_equal(di[a], e["dt_ignore"])
            np.testing.assert_array_equal(gi[a], e["gt_ignore"])
    # no detections at all; no ground truth at all; nothing at all
    g1 = [dict(image_id=1, category_id=1, bbox=[0.0, 0.0, 50.0, 50.0], area=2500.0, iscrowd=0)]
    for g, d in ((g1, []), ([], [dict(image_id=1, category_id=1, bbox=[0.0, 0.0, 50.0, 50.0], score=0.5)]), ([], [])):
        want = oc.evaluate(g, d, img_ids=[1], cat_ids=[1])
        got = ev.evaluate_predictions_on_coco({"annotations": g, "images": [{"id": 1}], "categories": [{"id": 1}]}, d)
        np.testing.assert_array_equal(got.stats, want["stats"])
    big = (np.zeros((1, 4)), np.zeros((600, 4)), np.zeros((600,)), np.zeros((600,), np.uint8))
    with pytest.raises(_lib.OsdError):
        ev.coco_match([big])
