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
