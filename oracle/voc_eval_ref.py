"""ORACLE — test infrastructure only.  CPU restatement of the reference's PASCAL VOC detection evaluation
(data/datasets/evaluation/voc/voc_eval.py, paths relative to /root/reference/maskrcnn_benchmark/): numpy, float32 box
arithmetic like the reference (BoxList holds float32 tensors).  Only `tests/` may import this module.

Pinned by tests/golden/voc_eval.npz, recorded through the REAL reference's eval_detection_voc / calc_detection_voc_prec_rec
(tests/golden/make_golden.py gen_voc_eval) on synthetic detections and ground truth: precision / recall arrays per class and
both AP metrics must agree exactly (tests/test_oracle_golden.py).
"""
import numpy as np


def boxlist_iou(a, b):
    """structures/boxlist_ops.py:221-256 ('+1' pixel convention, float32)."""
    a = np.asarray(a, np.float32).reshape(-1, 4)
    b = np.asarray(b, np.float32).reshape(-1, 4)
    one = np.float32(1)
    area1 = (a[:, 2] - a[:, 0] + one) * (a[:, 3] - a[:, 1] + one)        # bounding_box.py:226-231
    area2 = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.maximum(rb - lt + one, np.float32(0))
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2[None, :] - inter)


def match_image(pred_bbox, pred_label, pred_score, gt_bbox, gt_label, gt_difficult, iou_thresh=0.5):
    """The per-image body of calc_detection_voc_prec_rec (voc_eval.py:84-137).  Returns {label: (scores sorted descending the
    way the reference sorts them, match flags 1 / 0 / -1, number of non-difficult ground-truth boxes)}."""
    out = {}
    pred_bbox = np.asarray(pred_bbox, np.float32).reshape(-1, 4)
    gt_bbox = np.asarray(gt_bbox, np.float32).reshape(-1, 4)
    pred_label, gt_label = np.asarray(pred_label), np.asarray(gt_label)
    pred_score = np.asarray(pred_score, np.float32)
    gt_difficult = np.asarray(gt_difficult).astype(bool)
    for l in np.unique(np.concatenate((pred_label, gt_label)).astype(int)):
        pm = pred_label == l
        pb, ps = pred_bbox[pm], pred_score[pm]
        order = ps.argsort()[::-1]                                      # :93-95
        pb, ps = pb[order], ps[order]
        gm = gt_label == l
        gb, gd = gt_bbox[gm], gt_difficult[gm]
        n_pos = int(np.logical_not(gd).sum())
        match = []
        if len(pb) and not len(gb):
            match = [0] * len(pb)                                       # :106-108
        elif len(pb):
            pb2, gb2 = pb.copy(), gb.copy()
            pb2[:, 2:] += 1                                             # :111-114 (then boxlist_iou adds its own + 1)
            gb2[:, 2:] += 1
            iou = boxlist_iou(pb2, gb2)
            gt_index = iou.argmax(axis=1)
            gt_index[iou.max(axis=1) < np.float32(iou_thresh)] = -1
            selec = np.zeros(len(gb), dtype=bool)
            for gi in gt_index:                                          # :125-137
                if gi >= 0:
                    if gd[gi]:
                        match.append(-1)
                    else:
                        match.append(0 if selec[gi] else 1)
                    selec[gi] = True
                else:
                    match.append(0)
        out[int(l)] = (ps, np.asarray(match, np.int8), n_pos)
    return out


def calc_detection_voc_prec_rec(per_image):
    """voc_eval.py:70-158 from the per-image results of match_image."""
    n_pos, score, match = {}, {}, {}
    for res in per_image:
        for l, (ps, mt, npos) in res.items():
            n_pos[l] = n_pos.get(l, 0) + npos
            score.setdefault(l, []).extend(ps.tolist())
            match.setdefault(l, []).extend(mt.tolist())
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
            prec[l] = tp / (fp + tp)
        if n_pos[l] > 0:
            rec[l] = tp / n_pos[l]
    return prec, rec


def calc_detection_voc_ap(prec, rec, use_07_metric=False):
    """voc_eval.py:161-216."""
    ap = np.empty(len(prec))
    for l in range(len(prec)):
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


def eval_detection_voc(preds, gts, iou_thresh=0.5, use_07_metric=False):
    """voc_eval.py:48-67.  preds: [(boxes [n,4], labels [n], scores [n])], gts: [(boxes [g,4], labels [g], difficult [g])]."""
    assert len(preds) == len(gts)
    per_image = [match_image(pb, pl, ps, gb, gl, gd, iou_thresh) for (pb, pl, ps), (gb, gl, gd) in zip(preds, gts)]
    prec, rec = calc_detection_voc_prec_rec(per_image)
    ap = calc_detection_voc_ap(prec, rec, use_07_metric)
    return {"ap": ap, "map": np.nanmean(ap), "prec": prec, "rec": rec}
