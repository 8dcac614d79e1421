"""PASCAL VOC detection evaluation (the reference: data/datasets/evaluation/voc/voc_eval.py): the same three
functions with the same signatures — eval_detection_voc, calc_detection_voc_prec_rec, calc_detection_voc_ap — on BoxLists
(modules.BoxList; fields `labels`, `scores` on the predictions, `labels`, `difficult` on the ground truth).

The matching of detections to ground-truth boxes (IoU of every detection against every box of its class, the greedy claim in
score order) runs on the device, all images in ONE launch (osd_voc_match); the precision / recall curves and the AP integrals —
cumulative sums over a few thousand flags — are host numpy, the reference's own expressions.
Equal scores: the reference orders them by an unstable argsort; here equal scores are taken in descending index order."""
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


def calc_detection_voc_prec_rec(gt_boxlists, pred_boxlists, iou_thresh=0.5):
    """voc_eval.py:70-158.  -> (prec, rec): lists indexed by class id (None for a class that never occurs)."""
    flags = voc_match(pred_boxlists, gt_boxlists, iou_thresh)
    n_pos, score, match = {}, {}, {}
    for (mt, _), p, g in zip(flags, pred_boxlists, gt_boxlists):
        pl = torch.as_tensor(p.get_field("labels")).cpu().numpy() if len(p) else np.zeros((0,), np.int64)
        ps = torch.as_tensor(p.get_field("scores")).float().cpu().numpy() if len(p) else np.zeros((0,), np.float32)
        gl = torch.as_tensor(g.get_field("labels")).cpu().numpy() if len(g) else np.zeros((0,), np.int64)
        gdf = torch.as_tensor(g.get_field("difficult")).cpu().numpy().astype(bool) if len(g) else np.zeros((0,), bool)
        for l in np.unique(np.concatenate((pl, gl)).astype(int)):
            sel = np.nonzero(pl == l)[0]
            order = sel[np.argsort(ps[sel], kind="stable")[::-1]]          # descending score; equal scores: higher index first
            n_pos[l] = n_pos.get(l, 0) + int(np.logical_not(gdf[gl == l]).sum())
            score.setdefault(l, []).extend(ps[order].tolist())
            match.setdefault(l, []).extend(mt[order].tolist())
    n_fg_class = max(n_pos.keys()) + 1
    prec, rec = [None] * n_fg_class, [None] * n_fg_class
    for l in n_pos:
        score_l = np.array(score[l])
        match_l = np.array(match[l], dtype=np.int8)
        order = score_l.argsort()[::-1]
        match_l = match_l[order]
        tp = np.cumsum(match_l == 1)
        fp = np.cumsum(match_l == 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            prec[l] = tp / (fp + tp)                                     # nan where fp + tp == 0, like the reference
        if n_pos[l] > 0:
            rec[l] = tp / n_pos[l]
    return prec, rec


def calc_detection_voc_ap(prec, rec, use_07_metric=False):
    """voc_eval.py:161-216: the 11-point VOC 2007 metric or the area under the monotone precision envelope."""
    n_fg_class = len(prec)
    ap = np.empty(n_fg_class)
    for l in range(n_fg_class):
        if prec[l] is None or rec[l] is None:
            ap[l] = np.nan
            continue
        if use_07_metric:
            ap[l] = 0
            for t in np.arange(0.0, 1.1, 0.1):
                p = 0 if np.sum(rec[l] >= t) == 0 else np.max(np.nan_to_num(prec[l])[rec[l] >= t])
                ap[l] += p / 11
        else:
            mpre = np.concatenate(([0], np.nan_to_num(prec[l]), [0]))
            mrec = np.concatenate(([0], rec[l], [1]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            i = np.where(mrec[1:] != mrec[:-1])[0]
            ap[l] = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return ap


def eval_detection_voc(pred_boxlists, gt_boxlists, iou_thresh=0.5, use_07_metric=False):
    """voc_eval.py:48-67.  -> {"ap": per-class AP (nan for absent classes), "map": their nan-mean}."""
    assert len(gt_boxlists) == len(pred_boxlists), "Length of gt and pred lists need to be same."
    prec, rec = calc_detection_voc_prec_rec(pred_boxlists=pred_boxlists, gt_boxlists=gt_boxlists, iou_thresh=iou_thresh)
    ap = calc_detection_voc_ap(prec, rec, use_07_metric=use_07_metric)
    return {"ap": ap, "map": np.nanmean(ap)}
