"""PASCAL VOC detection evaluation (the reference: data/datasets/evaluation/voc/voc_eval.py): the same three
functions with the same signatures — eval_detection_voc, calc_detection_voc_prec_rec, calc_detection_voc_ap — on BoxLists
(modules.BoxList; fields `labels`, `scores` on the predictions, `labels`, `difficult` on the ground truth).

Everything numeric runs on the device, three launches per evaluation: osd_voc_match (IoU of every detection against every box
of its class and the greedy claim in score order, all images at once), osd_voc_curves (true / false positive prefix counts ->
precision / recall of every class) and osd_voc_ap (suffix-maximum precision envelope -> both AP metrics of every class).  The
host's part is bookkeeping: the dataset's detections are ordered ONCE by (class, descending score), class ranges and the
per-class count of non-difficult ground-truth boxes are integer histograms.
Equal scores: the reference orders them by an unstable argsort; here the later detection (higher image, then higher index) goes
first."""
import numpy as np
import torch

from . import _lib, ops


def voc_match(pred_boxlists, gt_boxlists, iou_thresh=0.5, device="cuda"):
    """-> per image (match flags int8 [n_det] in the detections' own order, matched ground-truth index int32 [n_det])."""
    n = len(pred_boxlists)
    assert len(gt_boxlists) == n, "Length of gt and pred lists need to be same."
    max_det = max([len(p) for p in pred_boxlists] + [1])
    max_gt = max([len(g) for g in gt_boxlists] + [1])
    db = np.zeros((n, max_det, 4), np.float32)
    ds = np.zeros((n, max_det), np.float32)
    dl = np.zeros((n, max_det), np.int32)
    dc = np.zeros((n,), np.int32)
    gb = np.zeros((n, max_gt, 4), np.float32)
    gl = np.zeros((n, max_gt), np.int32)
    gd = np.zeros((n, max_gt), np.uint8)
    gc = np.zeros((n,), np.int32)
    for i, (p, g) in enumerate(zip(pred_boxlists, gt_boxlists)):
        k, m = len(p), len(g)
        dc[i], gc[i] = k, m
        if k:
            db[i, :k] = torch.as_tensor(p.bbox).float().cpu().numpy()
            ds[i, :k] = torch.as_tensor(p.get_field("scores")).float().cpu().numpy()
            dl[i, :k] = torch.as_tensor(p.get_field("labels")).cpu().numpy()
        if m:
            gb[i, :m] = torch.as_tensor(g.bbox).float().cpu().numpy()
            gl[i, :m] = torch.as_tensor(g.get_field("labels")).cpu().numpy()
            gd[i, :m] = torch.as_tensor(g.get_field("difficult")).cpu().numpy().astype(np.uint8)
    dev = torch.device(device)
    t = [torch.from_numpy(a).to(dev) for a in (db, ds, dl, dc, gb, gl, gd, gc)]
    match = torch.empty((n, max_det), device=dev, dtype=torch.int8)
    mg = torch.empty((n, max_det), device=dev, dtype=torch.int32)
    _lib.call("osd_voc_match", *[ops._ptr(a) for a in t], n, max_det, max_gt, float(iou_thresh), ops._ptr(match), ops._ptr(mg),
              ops._stream())
    match, mg = match.cpu().numpy(), mg.cpu().numpy()
    return [(match[i, :dc[i]], mg[i, :dc[i]]) for i in range(n)]


def _as_np(t, dtype):
    return torch.as_tensor(t).cpu().numpy().astype(dtype, copy=False)


class _Curves(object):
    """Per-class curves of one evaluation: flat float64 arrays + class ranges, as osd_voc_curves writes them."""
    __slots__ = ("prec", "rec", "begin", "has_prec", "has_rec")


def _curves(gt_boxlists, pred_boxlists, iou_thresh, device="cuda"):
    flags = voc_match(pred_boxlists, gt_boxlists, iou_thresh, device)
    empty_i, empty_f = np.zeros((0,), np.int64), np.zeros((0,), np.float32)
    det_cls = np.concatenate([_as_np(p.get_field("labels"), np.int64) if len(p) else empty_i for p in pred_boxlists] + [empty_i])
    det_score = np.concatenate([_as_np(p.get_field("scores"), np.float32) if len(p) else empty_f for p in pred_boxlists] + [empty_f])
    det_flag = np.concatenate([f for f, _ in flags] + [np.zeros((0,), np.int8)])
    gt_cls = np.concatenate([_as_np(g.get_field("labels"), np.int64) if len(g) else empty_i for g in gt_boxlists] + [empty_i])
    gt_hard = np.concatenate([_as_np(g.get_field("difficult"), bool) if len(g) else np.zeros((0,), bool) for g in gt_boxlists] +
                             [np.zeros((0,), bool)])
    n_cls = int(max(det_cls.max(initial=-1), gt_cls.max(initial=-1))) + 1
    # one ordering pass for the whole dataset: class ascending, score descending, later detections first among equal scores
    order = np.lexsort((-np.arange(det_cls.size), -det_score, det_cls))
    begin = np.searchsorted(det_cls[order], np.arange(n_cls + 1)).astype(np.int32)
    n_pos = np.bincount(gt_cls[~gt_hard], minlength=n_cls).astype(np.int32)
    cv = _Curves()
    cv.begin = begin
    cv.has_prec = ((np.diff(begin) > 0) | (np.bincount(gt_cls, minlength=n_cls) > 0)).astype(np.uint8)
    cv.has_rec = (cv.has_prec.astype(bool) & (n_pos > 0)).astype(np.uint8)
    dev = torch.device(device)
    n = int(det_cls.size)
    prec = torch.empty((max(n, 1),), device=dev, dtype=torch.float64)
    rec = torch.empty((max(n, 1),), device=dev, dtype=torch.float64)
    if n_cls:
        t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (det_flag[order], begin, n_pos)]
        _lib.call("osd_voc_curves", ops._ptr(t[0]), ops._ptr(t[1]), ops._ptr(t[2]), n_cls, ops._ptr(prec), ops._ptr(rec), ops._stream())
    cv.prec, cv.rec = prec[:n], rec[:n]
    return cv


def _ap(cv, use_07_metric):
    n_cls = len(cv.has_prec)
    ap = torch.empty((max(n_cls, 1),), device=cv.prec.device, dtype=torch.float64)
    if n_cls:
        t = [torch.from_numpy(np.ascontiguousarray(a)).to(cv.prec.device) for a in (cv.begin, cv.has_prec, cv.has_rec)]
        _lib.call("osd_voc_ap", ops._ptr(cv.prec), ops._ptr(cv.rec), ops._ptr(t[0]), ops._ptr(t[1]), ops._ptr(t[2]), n_cls,
                  int(bool(use_07_metric)), ops._ptr(ap), ops._stream())
    return ap[:n_cls].cpu().numpy()


def calc_detection_voc_prec_rec(gt_boxlists, pred_boxlists, iou_thresh=0.5):
    """voc_eval.py:70-158.  -> (prec, rec): lists indexed by class id, float64 arrays over the class's detections in descending
    score order (None for a class id that never occurs; rec None for a class without non-difficult ground truth)."""
    cv = _curves(gt_boxlists, pred_boxlists, iou_thresh)
    prec, rec = cv.prec.cpu().numpy(), cv.rec.cpu().numpy()
    cut = lambda a, c: a[cv.begin[c]:cv.begin[c + 1]]                    # noqa: E731
    n_cls = len(cv.has_prec)
    return ([cut(prec, c) if cv.has_prec[c] else None for c in range(n_cls)],
            [cut(rec, c) if cv.has_rec[c] else None for c in range(n_cls)])


def calc_detection_voc_ap(prec, rec, use_07_metric=False, device="cuda"):
    """voc_eval.py:161-216 on curves given as lists (the reference's argument format): the 11-point VOC 2007 metric or the area
    under the monotone precision envelope, every class in one launch (osd_voc_ap); NaN for a class without curves."""
    n_cls = len(prec)
    cv = _Curves()
    cv.has_prec = np.array([p is not None for p in prec], np.uint8)
    cv.has_rec = np.array([p is not None and r is not None for p, r in zip(prec, rec)], np.uint8)
    sizes = [len(p) if p is not None else 0 for p in prec]
    cv.begin = np.concatenate(([0], np.cumsum(sizes))).astype(np.int32)
    flat = lambda xs: np.concatenate([np.asarray(x if x is not None else np.full((k,), np.nan), np.float64)    # noqa: E731
                                      for x, k in zip(xs, sizes)] + [np.zeros((1,), np.float64)])
    dev = torch.device(device)
    cv.prec, cv.rec = torch.from_numpy(flat(prec)).to(dev), torch.from_numpy(flat(rec)).to(dev)
    return _ap(cv, use_07_metric) if n_cls else np.zeros((0,), np.float64)


def eval_detection_voc(pred_boxlists, gt_boxlists, iou_thresh=0.5, use_07_metric=False):
    """voc_eval.py:48-67.  -> {"ap": per-class AP (nan for absent classes), "map": their nan-mean}.  The curves stay on the device
    between the two launches; the host receives ap[]."""
    assert len(gt_boxlists) == len(pred_boxlists), "Length of gt and pred lists need to be same."
    ap = _ap(_curves(gt_boxlists, pred_boxlists, iou_thresh), use_07_metric)
    return {"ap": ap, "map": np.nanmean(ap)}


# ------------------------------------------------------------------------------------------------------------------------------
# COCO-style evaluation, bbox (data/datasets/evaluation/coco/coco_eval.py:385-408: the reference hands its detections to
# pycocotools' COCOeval and reads `stats`).  pycocotools is a third-party package that is not in this image: the algorithm here is
# its published one (cocoeval.py evaluate / accumulate / summarize), restated — PARITY UNPINNED, see oracle/coco_eval_ref.py.
# The per-(image, category) matching at 10 IoU thresholds x 4 area ranges runs on the device in ONE launch (osd_coco_match); the
# host sorts, concatenates and integrates.
COCO_IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
COCO_REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
COCO_MAX_DETS = (1, 10, 100)
COCO_AREA_RNG = ((0 ** 2, 1e5 ** 2), (0 ** 2, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e5 ** 2))
COCO_AREA_LBL = ("all", "small", "medium", "large")
COCO_BBOX_METRICS = ("AP", "AP50", "AP75", "APs", "APm", "APl")      # COCOResults.METRICS["bbox"], coco_eval.py:441-443


class CocoEval(object):
    """What the reference reads off a pycocotools COCOeval after evaluate() / accumulate() / summarize(): `stats` (12 numbers)
    and `eval` = {precision [T,R,K,A,M], recall [T,K,A,M], scores [T,R,K,A,M]}; `results()` = COCOResults' bbox metrics."""

    def __init__(self, precision, recall, scores, cat_ids, img_ids):
        import types
        self.eval = dict(precision=precision, recall=recall, scores=scores)
        self.cat_ids, self.img_ids = cat_ids, img_ids
        # what the reference's consumers read off COCOeval.params (coco_eval.py: compute_thresholds_for_classes, COCOResults)
        self.params = types.SimpleNamespace(iouType="bbox", iouThrs=COCO_IOU_THRS.copy(), recThrs=COCO_REC_THRS.copy(),
                                            maxDets=list(COCO_MAX_DETS), areaRng=[list(r) for r in COCO_AREA_RNG],
                                            areaRngLbl=list(COCO_AREA_LBL), catIds=list(cat_ids), imgIds=list(img_ids), useCats=1)
        self.stats = self._summarize()

    def _one(self, ap, iou_thr=None, area="all", max_det=100):
        a, m = COCO_AREA_LBL.index(area), COCO_MAX_DETS.index(max_det)
        s = self.eval["precision" if ap else "recall"]
        if iou_thr is not None:
            s = s[np.where(iou_thr == COCO_IOU_THRS)[0]]
        s = s[..., a, m]
        valid = s[s > -1]
        return -1.0 if valid.size == 0 else float(valid.mean())

    def _summarize(self):
        one = self._one
        return np.array([one(1), one(1, iou_thr=.5), one(1, iou_thr=.75), one(1, area="small"), one(1, area="medium"),
                         one(1, area="large"), one(0, max_det=1), one(0, max_det=10), one(0), one(0, area="small"),
                         one(0, area="medium"), one(0, area="large")])

    def results(self):
        return {"bbox": {k: float(v) for k, v in zip(COCO_BBOX_METRICS, self.stats[:6])}}


def coco_match(pairs, device="cuda"):
    """pairs: list of (det_boxes [d, 4] xywh float64 sorted by descending score, gt_boxes [g, 4] xywh, gt_area [g], gt_crowd [g]).
    -> per pair (dt_match int32 [A, T, d], dt_ignore bool [A, T, d], gt_ignore bool [A, g]) from ONE launch of osd_coco_match."""
    import ctypes as C
    n = len(pairs)
    A, T = len(COCO_AREA_RNG), len(COCO_IOU_THRS)
    if n == 0:
        return []
    max_det = max([len(p[0]) for p in pairs] + [1])
    max_gt = max([len(p[1]) for p in pairs] + [1])
    db, dc = np.zeros((n, max_det, 4), np.float64), np.zeros((n,), np.int32)
    gb, ga = np.zeros((n, max_gt, 4), np.float64), np.zeros((n, max_gt), np.float64)
    gc, gn = np.zeros((n, max_gt), np.uint8), np.zeros((n,), np.int32)
    for i, (d, g, area, crowd) in enumerate(pairs):
        dc[i], gn[i] = len(d), len(g)
        if len(d):
            db[i, :len(d)] = d
        if len(g):
            gb[i, :len(g)], ga[i, :len(g)], gc[i, :len(g)] = g, area, crowd
    dev = torch.device(device)
    t = [torch.from_numpy(x).to(dev) for x in (db, dc, gb, ga, gc, gn)]
    dm = torch.empty((n, A, T, max_det), device=dev, dtype=torch.int32)
    di = torch.empty((n, A, T, max_det), device=dev, dtype=torch.uint8)
    gi = torch.empty((n, A, max_gt), device=dev, dtype=torch.uint8)
    thr = (C.c_double * T)(*COCO_IOU_THRS.tolist())
    rng = (C.c_double * (2 * A))(*[float(v) for r in COCO_AREA_RNG for v in r])
    _lib.call("osd_coco_match", *[ops._ptr(x) for x in t], n, max_det, max_gt, thr, T, rng, A, ops._ptr(dm), ops._ptr(di), ops._ptr(gi),
              ops._stream())
    dm, di, gi = dm.cpu().numpy(), di.cpu().numpy().astype(bool), gi.cpu().numpy().astype(bool)
    return [(dm[i, :, :, :dc[i]], di[i, :, :, :dc[i]], gi[i, :, :gn[i]]) for i in range(n)]


def evaluate_predictions_on_coco(coco_gt, coco_results, json_result_file=None, iou_type="bbox", img_ids=None, cat_ids=None, device="cuda"):
    """coco_eval.py:385-408, the reference's positional order (coco_gt, coco_results, json_result_file, iou_type='bbox'): coco_gt =
    the COCO-format ground truth (a dict with "annotations" — image_id, category_id, bbox [x, y, w, h], area, iscrowd — and
    optionally "images" / "categories", or the annotation list itself), coco_results = the detection list
    prepare_for_coco_detection builds (image_id, category_id, bbox, score); json_result_file: where the reference dumps
    coco_results before handing the file to pycocotools — written here too when given (nothing reads it back).  -> CocoEval."""
    if json_result_file is not None:
        import json
        with open(json_result_file, "w") as f:
            json.dump(list(coco_results), f)
    if iou_type != "bbox":
        raise NotImplementedError("iou_type %r: only the box metric is built (the hot path has no mask / keypoint head)" % iou_type)
    anns = coco_gt["annotations"] if isinstance(coco_gt, dict) else list(coco_gt)
    if img_ids is None:
        img_ids = [im["id"] for im in coco_gt["images"]] if isinstance(coco_gt, dict) and coco_gt.get("images") else \
            list({a["image_id"] for a in anns} | {d["image_id"] for d in coco_results})
    if cat_ids is None:
        cat_ids = [c["id"] for c in coco_gt["categories"]] if isinstance(coco_gt, dict) and coco_gt.get("categories") else \
            list({a["category_id"] for a in anns})
    img_ids, cat_ids = sorted(img_ids), sorted(cat_ids)
    by_g, by_d = {}, {}
    for a in anns:
        by_g.setdefault((a["image_id"], a["category_id"]), []).append(a)
    for d in coco_results:
        by_d.setdefault((d["image_id"], d["category_id"]), []).append(d)
    # one row per (category, image) that has anything to evaluate, detections in descending score order (stable), cut at 100
    keys, pairs, scores = [], [], []
    for cat in cat_ids:
        for img in img_ids:
            g, d = by_g.get((img, cat), []), by_d.get((img, cat), [])
            if not g and not d:
                continue
            sc = np.array([x["score"] for x in d], np.float64)
            order = np.argsort(-sc, kind="mergesort")[:COCO_MAX_DETS[-1]]
            keys.append((cat, img))
            scores.append(sc[order])
            pairs.append((np.array([d[i]["bbox"] for i in order], np.float64).reshape(-1, 4),
                          np.array([x["bbox"] for x in g], np.float64).reshape(-1, 4),
                          np.array([x["area"] for x in g], np.float64), np.array([int(x.get("iscrowd", 0)) for x in g], np.uint8)))
    matched = coco_match(pairs, device)
    T, R, K, A, M = len(COCO_IOU_THRS), len(COCO_REC_THRS), len(cat_ids), len(COCO_AREA_RNG), len(COCO_MAX_DETS)
    precision, recall, pscores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
    rows_of = {}
    for i, (cat, _) in enumerate(keys):
        rows_of.setdefault(cat, []).append(i)
    eps = np.spacing(1)
    for k, cat in enumerate(cat_ids):
        rows = rows_of.get(cat, [])
        if not rows:
            continue
        for a in range(A):
            gt_ig = np.concatenate([matched[i][2][a] for i in rows])
            npig = int(np.count_nonzero(~gt_ig))
            if npig == 0:
                continue
            for m, max_det in enumerate(COCO_MAX_DETS):
                sc = np.concatenate([scores[i][:max_det] for i in rows])
                order = np.argsort(-sc, kind="mergesort")
                hit = np.concatenate([matched[i][0][a][:, :max_det] for i in rows], axis=1)[:, order] > 0
                ign = np.concatenate([matched[i][1][a][:, :max_det] for i in rows], axis=1)[:, order]
                tp = np.cumsum(hit & ~ign, axis=1).astype(np.float64)          # [T, nd]
                fp = np.cumsum(~hit & ~ign, axis=1).astype(np.float64)
                nd = tp.shape[1]
                if nd == 0:
                    recall[:, k, a, m] = 0
                    precision[:, :, k, a, m] = 0
                    pscores[:, :, k, a, m] = 0
                    continue
                rc = tp / npig
                pr = tp / (fp + tp + eps)
                recall[:, k, a, m] = rc[:, -1]
                env = np.maximum.accumulate(pr[:, ::-1], axis=1)[:, ::-1]     # precision made monotone from the right
                ssc = sc[order]
                for t in range(T):
                    pos = np.searchsorted(rc[t], COCO_REC_THRS, side="left")
                    ok = pos < nd
                    q, s_ = np.zeros((R,)), np.zeros((R,))
                    q[ok], s_[ok] = env[t][pos[ok]], ssc[pos[ok]]
                    precision[t, :, k, a, m], pscores[t, :, k, a, m] = q, s_
    return CocoEval(precision, recall, pscores, cat_ids, img_ids)
