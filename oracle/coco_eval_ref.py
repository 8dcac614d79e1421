"""ORACLE (test infrastructure: imported by tests/ only, never by the product) — COCO-style detection evaluation, bbox.

PARITY UNPINNED.  The reference evaluates its COCO-format configs through `pycocotools.cocoeval.COCOeval`
(data/datasets/evaluation/coco/coco_eval.py:385-408: COCOeval(coco_gt, coco_dt, "bbox"); evaluate(); accumulate();
summarize()), a third-party dependency that is NOT under /root/reference: cocodataset/cocoapi, PythonAPI, installed from the
repository's master by INSTALL.md:34-38 (no pinned version), and absent from this image (no network).  This file restates
that package's published algorithm (cocoeval.py: _prepare / computeIoU / evaluateImg / accumulate / summarize; maskApi.c:
bbIou) with plain Python loops; no fixture recorded through pycocotools exists, so what pins it are the known-answer cases
of tests/test_oracle_coco_eval.py (hand-computed AP / AR of small configurations) — not the package itself.

Conventions restated:
  * boxes are [x, y, w, h]; IoU = intersection / union in double precision with NO '+1'; against a crowd ground-truth box the
    union is the detection's own area;
  * a ground-truth box is ignored for an area range when it is a crowd box or its `area` field lies outside the range; ignored
    boxes are sorted behind the others (stable);
  * detections of an (image, category) are sorted by descending score (stable) and cut at maxDets[-1]; each, in that order,
    takes the not-yet-taken (or crowd) ground-truth box with the highest IoU >= the threshold, preferring non-ignored boxes
    (the scan stops at the first ignored box once a non-ignored match is held); a detection matched to an ignored box, or
    unmatched and with an area outside the range, is itself ignored;
  * precision is made monotone from the right, sampled at 101 recall thresholds by searchsorted(side='left'); AP / AR are
    means over the entries that are not -1."""
import numpy as np

IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
MAX_DETS = (1, 10, 100)
AREA_RNG = ((0 ** 2, 1e5 ** 2), (0 ** 2, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e5 ** 2))
AREA_LBL = ("all", "small", "medium", "large")


def bb_iou(d, g, crowd):
    """maskApi.c bbIou for one pair: [x, y, w, h] boxes, double precision."""
    w = min(d[2] + d[0], g[2] + g[0]) - max(d[0], g[0])
    if w <= 0:
        return 0.0
    h = min(d[3] + d[1], g[3] + g[1]) - max(d[1], g[1])
    if h <= 0:
        return 0.0
    i = w * h
    u = d[2] * d[3] if crowd else d[2] * d[3] + g[2] * g[3] - i
    return i / u


def evaluate_img(gt, dt, a_rng, max_det, iou_thrs=IOU_THRS):
    """cocoeval.py evaluateImg for one (image, category).  gt: dicts with bbox / area / iscrowd; dt: dicts with bbox / score."""
    if len(gt) == 0 and len(dt) == 0:
        return None
    ig = [bool(g.get("iscrowd", 0)) or g["area"] < a_rng[0] or g["area"] > a_rng[1] for g in gt]
    gtind = np.argsort(np.array(ig, dtype=np.int64), kind="mergesort") if gt else np.zeros((0,), np.int64)
    gts = [gt[i] for i in gtind]
    gt_ig = np.array([ig[i] for i in gtind], dtype=bool)
    dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
    dts = [dt[i] for i in dtind[:max_det]]
    T, G, D = len(iou_thrs), len(gts), len(dts)
    gtm, dtm, dt_ig = np.zeros((T, G), np.int64), np.zeros((T, D), np.int64), np.zeros((T, D), bool)
    for ti, t in enumerate(iou_thrs):
        for di, d in enumerate(dts):
            iou, m = min(t, 1 - 1e-10), -1
            for gi, g in enumerate(gts):
                crowd = bool(g.get("iscrowd", 0))
                if gtm[ti, gi] > 0 and not crowd:
                    continue
                if m > -1 and not gt_ig[m] and gt_ig[gi]:
                    break
                o = bb_iou(d["bbox"], g["bbox"], crowd)
                if o < iou:
                    continue
                iou, m = o, gi
            if m == -1:
                continue
            dt_ig[ti, di] = gt_ig[m]
            dtm[ti, di] = m + 1
            gtm[ti, m] = di + 1
    out_of_range = np.array([d["bbox"][2] * d["bbox"][3] < a_rng[0] or d["bbox"][2] * d["bbox"][3] > a_rng[1] for d in dts], dtype=bool)
    dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, out_of_range[None, :].repeat(T, 0) if D else np.zeros((T, 0), bool)))
    return dict(dt_matches=dtm, dt_scores=np.array([d["score"] for d in dts], np.float64), gt_ignore=gt_ig, dt_ignore=dt_ig)


def evaluate(gts, dts, img_ids=None, cat_ids=None):
    """evaluate() + accumulate() + summarize().  -> dict(precision [T,R,K,A,M], recall [T,K,A,M], stats [12])."""
    img_ids = sorted(set(g["image_id"] for g in gts) | set(d["image_id"] for d in dts)) if img_ids is None else sorted(img_ids)
    cat_ids = sorted(set(g["category_id"] for g in gts)) if cat_ids is None else sorted(cat_ids)
    by_g, by_d = {}, {}
    for g in gts:
        by_g.setdefault((g["image_id"], g["category_id"]), []).append(g)
    for d in dts:
        by_d.setdefault((d["image_id"], d["category_id"]), []).append(d)
    T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(cat_ids), len(AREA_RNG), len(MAX_DETS)
    precision, recall = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M))
    for k, cat in enumerate(cat_ids):
        for a, rng in enumerate(AREA_RNG):
            E = [evaluate_img(by_g.get((i, cat), []), by_d.get((i, cat), []), rng, MAX_DETS[-1]) for i in img_ids]
            E = [e for e in E if e is not None]
            if not E:
                continue
            for m, max_det in enumerate(MAX_DETS):
                scores = np.concatenate([e["dt_scores"][:max_det] for e in E])
                inds = np.argsort(-scores, kind="mergesort")
                dtm = np.concatenate([e["dt_matches"][:, :max_det] for e in E], axis=1)[:, inds]
                dt_ig = np.concatenate([e["dt_ignore"][:, :max_det] for e in E], axis=1)[:, inds]
                gt_ig = np.concatenate([e["gt_ignore"] for e in E])
                npig = np.count_nonzero(gt_ig == 0)
                if npig == 0:
                    continue
                tps = np.logical_and(dtm, np.logical_not(dt_ig))
                fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                tp_sum, fp_sum = np.cumsum(tps, axis=1).astype(float), np.cumsum(fps, axis=1).astype(float)
                for t in range(T):
                    tp, fp = tp_sum[t], fp_sum[t]
                    nd = len(tp)
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    q = np.zeros((R,))
                    recall[t, k, a, m] = rc[-1] if nd else 0
                    pr = pr.tolist()
                    for i in range(nd - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    for ri, pi in enumerate(np.searchsorted(rc, REC_THRS, side="left")):
                        if pi < nd:
                            q[ri] = pr[pi]
                    precision[t, :, k, a, m] = q
    return dict(precision=precision, recall=recall, stats=summarize(precision, recall))


def summarize(precision, recall):
    def one(ap, iou_thr=None, area="all", max_det=100):
        a, m = AREA_LBL.index(area), MAX_DETS.index(max_det)
        s = precision if ap else recall
        if iou_thr is not None:
            s = s[np.where(iou_thr == IOU_THRS)[0]]
        s = s[:, :, :, a, m] if ap else s[:, :, a, m]
        return -1.0 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))
    return np.array([one(1), one(1, iou_thr=.5), one(1, iou_thr=.75), one(1, area="small"), one(1, area="medium"), one(1, area="large"),
                     one(0, max_det=1), one(0, max_det=10), one(0), one(0, area="small"), one(0, area="medium"), one(0, area="large")])
